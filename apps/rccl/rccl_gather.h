// The trajectory gather of the C++ VisualOdometry app's multi-device mode (--batch --gpus N --rccl): ONE process, one host
// thread and one RCCL communicator per device (ncclCommInitAll), one all_gather of the per-device [pairs][6] fp64 state
// blocks -- straight from each engine's device buffer (phovo_engine_results_device_ptr) -- after which rank 0 holds every
// shard's states and chains the poses (apps/PhotoconsistencyVisualOdometry/PhotoconsistencyVisualOdometry.cpp:233-243).
// The alignment itself needs no collective: pairs are independent (:175,222-224).
#pragma once

#include <string>
#include <vector>

namespace phovo_rccl {

class Group {
 public:
  // One communicator per entry of `devices` (distinct HIP device indices: RCCL refuses two ranks on one device).
  // Returns false and fills `error` when RCCL or HIP fails.
  bool create(const std::vector<int> &devices, std::string *error);
  // Called by rank `rank`'s host thread, every rank once per gather.  d_states: that rank's device buffer of count x 6
  // doubles (count may differ per rank; max_count is the largest).  After every rank has returned true, gathered(r)
  // points at rank r's count x 6 doubles in host memory (valid on every rank's view: one process).
  bool all_gather_states(int rank, const void *d_states, int count, int max_count, std::string *error);
  const double *gathered(int rank) const { return host_.data() + (size_t)rank * (size_t)max_count_ * 6; }
  void destroy();
  ~Group() { destroy(); }

 private:
  struct Rank { int device = -1; void *comm = nullptr; void *stream = nullptr; void *send = nullptr; void *recv = nullptr; size_t capacity = 0; };
  std::vector<Rank> ranks_;
  std::vector<double> host_;
  int max_count_ = 0;
};

}  // namespace phovo_rccl

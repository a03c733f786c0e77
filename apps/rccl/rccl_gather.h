// The trajectory gather of the C++ VisualOdometry app's multi-device mode (--batch --gpus N --rccl): ONE process, one host
// thread and one RCCL communicator per device (ncclCommInitAll), one all_gather of the per-device [pairs][6] fp64 state
// blocks -- straight from each engine's device buffer (phovo_engine_results_device_ptr) -- after which rank 0 holds every
// shard's states and chains the poses (apps/PhotoconsistencyVisualOdometry/PhotoconsistencyVisualOdometry.cpp:233-243).
// The alignment itself needs no collective: pairs are independent (:175,222-224).
#pragma once

#include <mutex>
#include <string>
#include <vector>

namespace phovo_rccl {

class Group {
 public:
  // One communicator per entry of `devices` (distinct HIP device indices: RCCL refuses two ranks on one device).
  // Returns false and fills `error` when RCCL or HIP fails.
  bool create(const std::vector<int> &devices, std::string *error);
  // A gather is two calls per rank, each by that rank's host thread, with a vote of all ranks in between (shard_vote.h):
  //   stage()   everything that can fail on ONE rank without the others noticing -- buffers, the padded copy of that rank's
  //             count x 6 doubles (d_states: its device buffer; count may differ per rank, max_count is the largest; a rank
  //             without pairs passes count 0) -- and NO collective;
  //   gather()  the all_gather itself: only when every rank's stage() succeeded, because a rank that stays away leaves the
  //             others waiting in it.  A rank whose gather() fails aborts every communicator of the group (abort()), so
  //             that the others return with an error instead of waiting.
  // After every rank's gather() has returned true, gathered(r) points at rank r's count x 6 doubles in host memory (valid
  // on every rank's view: one process).
  bool stage(int rank, const void *d_states, int count, int max_count, std::string *error);
  bool gather(int rank, std::string *error);
  // stage() + gather() for a caller that has no other ranks to wait for (one device)
  bool all_gather_states(int rank, const void *d_states, int count, int max_count, std::string *error)
  {
    return stage(rank, d_states, count, max_count, error) && gather(rank, error);
  }
  void abort();
  const double *gathered(int rank) const { return host_.data() + (size_t)rank * (size_t)max_count_ * 6; }
  void destroy();
  ~Group() { destroy(); }

 private:
  struct Rank { int device = -1; void *comm = nullptr; void *stream = nullptr; void *send = nullptr; void *recv = nullptr; size_t capacity = 0; int staged = 0; };
  std::mutex abort_mutex_;
  std::vector<Rank> ranks_;
  std::vector<double> host_;
  int max_count_ = 0;
};

}  // namespace phovo_rccl

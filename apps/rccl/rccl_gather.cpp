#include "rccl_gather.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

namespace phovo_rccl {

namespace {
bool hip_ok(hipError_t e, const char *what, std::string *error)
{
  if (e == hipSuccess) return true;
  if (error) *error = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}
bool nccl_ok(ncclResult_t r, const char *what, std::string *error)
{
  if (r == ncclSuccess) return true;
  if (error) *error = std::string(what) + ": " + ncclGetErrorString(r);
  return false;
}
}  // namespace

bool Group::create(const std::vector<int> &devices, std::string *error)
{
  destroy();
  const int n = (int)devices.size();
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++)
      if (devices[i] == devices[j]) {
        if (error) *error = "RCCL needs one rank per device: device " + std::to_string(devices[i]) + " is named twice";
        return false;
      }
  std::vector<ncclComm_t> comms((size_t)n);
  if (!nccl_ok(ncclCommInitAll(comms.data(), n, devices.data()), "ncclCommInitAll", error)) return false;
  ranks_.resize((size_t)n);
  for (int r = 0; r < n; r++) {          // every communicator is owned by the group from here on: destroy() releases them all
    ranks_[r].device = devices[r];
    ranks_[r].comm = comms[r];
  }
  for (int r = 0; r < n; r++) {
    hipStream_t s = nullptr;
    if (!hip_ok(hipSetDevice(devices[r]), "hipSetDevice", error) ||
        !hip_ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate", error)) {
      destroy();
      return false;
    }
    ranks_[r].stream = s;
  }
  return true;
}

bool Group::stage(int rank, const void *d_states, int count, int max_count, std::string *error)
{
  if (rank < 0 || rank >= (int)ranks_.size() || count < 0 || count > max_count || max_count <= 0 || (count > 0 && !d_states)) {
    if (error) *error = "stage: bad rank or count";
    return false;
  }
  Rank &k = ranks_[(size_t)rank];
  const int n = (int)ranks_.size();
  const size_t block = (size_t)max_count * 6 * sizeof(double);
  hipStream_t stream = static_cast<hipStream_t>(k.stream);
  if (!hip_ok(hipSetDevice(k.device), "hipSetDevice", error)) return false;
  if (block > k.capacity) {               // send: one padded block; recv: one block per rank
    if (k.send) (void)hipFree(k.send);
    if (k.recv) (void)hipFree(k.recv);
    k.send = k.recv = nullptr;
    if (!hip_ok(hipMalloc(&k.send, block), "hipMalloc(send)", error) ||
        !hip_ok(hipMalloc(&k.recv, block * (size_t)n), "hipMalloc(recv)", error))
      return false;
    k.capacity = block;
  }
  k.staged = max_count;
  if (rank == 0) { max_count_ = max_count; host_.assign((size_t)n * (size_t)max_count * 6, 0.0); }
  // the shard's states, padded with zeros to the common block size (all_gather moves equal blocks)
  if (!hip_ok(hipMemsetAsync(k.send, 0, block, stream), "hipMemsetAsync", error)) return false;
  if (count > 0 && !hip_ok(hipMemcpyAsync(k.send, d_states, (size_t)count * 6 * sizeof(double), hipMemcpyDeviceToDevice, stream),
                           "hipMemcpyAsync(states)", error))
    return false;
  return hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize(stage)", error);
}

bool Group::gather(int rank, std::string *error)
{
  if (rank < 0 || rank >= (int)ranks_.size() || ranks_[(size_t)rank].staged <= 0) {
    if (error) *error = "gather: nothing staged for this rank";
    return false;
  }
  Rank &k = ranks_[(size_t)rank];
  const int n = (int)ranks_.size();
  const size_t block = (size_t)k.staged * 6 * sizeof(double);
  hipStream_t stream = static_cast<hipStream_t>(k.stream);
  // one host thread per rank calls this concurrently: the collective completes when every rank has joined
  const bool ok =
      hip_ok(hipSetDevice(k.device), "hipSetDevice", error) &&
      nccl_ok(ncclAllGather(k.send, k.recv, (size_t)k.staged * 6, ncclDouble, static_cast<ncclComm_t>(k.comm), stream),
              "ncclAllGather", error) &&
      (rank != 0 || hip_ok(hipMemcpyAsync(host_.data(), k.recv, block * (size_t)n, hipMemcpyDeviceToHost, stream),
                           "hipMemcpyAsync(gathered)", error)) &&
      hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize", error);
  k.staged = 0;
  if (!ok) abort();        // the other ranks may be inside the collective, waiting for this one
  return ok;
}

void Group::abort()
{
  std::lock_guard<std::mutex> lock(abort_mutex_);
  for (Rank &k : ranks_) {
    if (k.comm) (void)ncclCommAbort(static_cast<ncclComm_t>(k.comm));     // frees the communicator: destroy() must not touch it again
    k.comm = nullptr;
  }
}

void Group::destroy()
{
  for (Rank &k : ranks_) {
    if (k.device >= 0) (void)hipSetDevice(k.device);
    if (k.send) (void)hipFree(k.send);
    if (k.recv) (void)hipFree(k.recv);
    if (k.stream) (void)hipStreamDestroy(static_cast<hipStream_t>(k.stream));
    if (k.comm) (void)ncclCommDestroy(static_cast<ncclComm_t>(k.comm));
  }
  ranks_.clear();
}

}  // namespace phovo_rccl

// The vote N shard threads take BEFORE a collective that every one of them must join (--batch --gpus N --rccl): a thread
// that failed on its way there -- engine creation, upload, alignment -- would otherwise simply return, and the other N - 1
// would wait in ncclAllGather for a rank that never comes.  Every shard calls arrive() exactly once, whether it succeeded
// or not (also a shard with no pairs); arrive() returns when all have, with the same answer for everyone: true only if
// every shard said ok.  Plain C++17 (mutex + condition variable), no HIP, no RCCL: tests/native/shard_vote_test.cpp.
#pragma once

#include <condition_variable>
#include <mutex>

namespace phovo_rccl {

class ShardVote {
 public:
  explicit ShardVote(int shards) : shards_(shards) {}
  bool arrive(bool ok)
  {
    std::unique_lock<std::mutex> lock(m_);
    if (!ok) failed_++;
    if (++arrived_ >= shards_) cv_.notify_all();
    else cv_.wait(lock, [this] { return arrived_ >= shards_; });
    return failed_ == 0;
  }
  int failed() const { return failed_; }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  int shards_, arrived_ = 0, failed_ = 0;
};

}  // namespace phovo_rccl

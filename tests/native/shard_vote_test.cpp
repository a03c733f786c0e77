// CPU test of apps/rccl/shard_vote.h (g++ -std=c++17 -pthread; no HIP, no RCCL): N threads stand for the shard threads of
// PhotoconsistencyVisualOdometry --batch --gpus N --rccl, a blocking barrier stands for the all_gather (it returns only
// when all N have joined, as the collective does).  With one failing shard nobody may enter the "collective", everybody
// must return, and everybody must have been told the same thing.
//   usage: shard_vote_test <shards> <failing shard or -1>      prints "entered=<k> told_ok=<k>", exit 0
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "rccl/shard_vote.h"

int main(int argc, char **argv)
{
  const int n = argc > 1 ? std::atoi(argv[1]) : 4, failing = argc > 2 ? std::atoi(argv[2]) : -1;
  phovo_rccl::ShardVote vote(n), collective(n);      // (the second instance is the stand-in collective: arrive() blocks until all have)
  std::atomic<int> entered(0), told_ok(0);
  std::vector<std::thread> shards;
  for (int g = 0; g < n; g++)
    shards.emplace_back([&, g] {
      std::this_thread::sleep_for(std::chrono::milliseconds(3 * ((g * 7) % 5)));      // shards get there at different times
      const bool ok = g != failing;                                                   // (a failing shard arrives too)
      if (vote.arrive(ok)) {
        told_ok++;
        entered++;
        collective.arrive(true);
      }
    });
  for (auto &t : shards) t.join();
  std::printf("entered=%d told_ok=%d failed=%d\n", entered.load(), told_ok.load(), vote.failed());
  return 0;
}

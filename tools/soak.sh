#!/bin/bash
# Robustness session on the GPU box (through gpurun), in two calls (a call is limited to 20 minutes):
#     bash tools/soak.sh gpurun_out/<dir> fuzz [seed [scale]]   long randomised sweeps against the oracle (tests/tools/fuzz_parity.py;
#                                                    other seeds and `scale` times the case counts for a second session)
#     bash tools/soak.sh gpurun_out/<dir> soak      thousands of shuffled copies through the work queue (tools/queue_soak.py)
set -o pipefail
O=$1
mkdir -p $O
run() { name=$1; shift; timeout -k 10 1000 "$@" > $O/$name.txt 2>&1; echo "$name rc=$? $(tail -3 $O/$name.txt | tr '\n' ' ' | cut -c1-300)"; }
if [ "$2" = "fuzz" ]; then
  S=${3:-51}; X=${4:-1}
  run fuzz_plain python3 tests/tools/fuzz_parity.py $((3000 * X)) $S
  run fuzz_ext python3 tests/tools/fuzz_parity.py $((1200 * X)) $((S + 1)) ext
  run fuzz_big python3 tests/tools/fuzz_parity.py $((300 * X)) $((S + 2)) big
  run fuzz_strips python3 tests/tools/fuzz_parity.py $((1500 * X)) $((S + 3)) strips
else
  run soak_shipped python3 tools/queue_soak.py 300 shipped
  run soak_pipelined python3 tools/queue_soak.py 300 pipelined
  run soak_layered python3 tools/queue_soak.py 150 layered
  run soak_fixed python3 tools/queue_soak.py 200 fixed
  run soak_slide python3 tools/queue_soak.py 100 slide
  run single_pair python3 tests/tools/single_pair_latency.py
fi

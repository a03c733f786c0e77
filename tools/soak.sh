#!/bin/bash
# Robustness session on the GPU box (through gpurun), in two calls (a call is limited to 20 minutes):
#     bash tools/soak.sh gpurun_out/<dir> fuzz      long randomised sweeps against the oracle (tests/tools/fuzz_parity.py)
#     bash tools/soak.sh gpurun_out/<dir> soak      thousands of shuffled copies through the work queue (tools/queue_soak.py)
set -o pipefail
O=$1
mkdir -p $O
run() { name=$1; shift; timeout -k 10 500 "$@" > $O/$name.txt 2>&1; echo "$name rc=$? $(tail -3 $O/$name.txt | tr '\n' ' ' | cut -c1-300)"; }
if [ "$2" = "fuzz" ]; then
  run fuzz_plain python3 tests/tools/fuzz_parity.py 3000 51
  run fuzz_ext python3 tests/tools/fuzz_parity.py 1200 52 ext
  run fuzz_big python3 tests/tools/fuzz_parity.py 300 53 big
  run fuzz_strips python3 tests/tools/fuzz_parity.py 1500 54 strips
else
  run soak_shipped python3 tools/queue_soak.py 300 shipped
  run soak_pipelined python3 tools/queue_soak.py 300 pipelined
  run soak_layered python3 tools/queue_soak.py 150 layered
  run soak_fixed python3 tools/queue_soak.py 200 fixed
  run soak_slide python3 tools/queue_soak.py 100 slide
  run single_pair python3 tests/tools/single_pair_latency.py
fi

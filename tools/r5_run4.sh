#!/bin/bash
# round 5, GPU session 4: byte intensities on level 0 -- whole GPU suite, then cfg1 / cfg5 / headline next to the fp64 layout
set -o pipefail
O=gpurun_out/r5d
mkdir -p $O
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
echo "pytest gpu rc=$rc"; tail -5 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  timeout -k 10 300 python3 bench.py --workload cfg1 --distinct 32 --no-cpu-baseline --no-reference-termination > $O/cfg1_bytes.$rep.json 2> $O/cfg1_bytes.$rep.err
  python3 tools/benchsum.py $O/cfg1_bytes.$rep.json "cfg1 byte intensities .$rep" || tail -3 $O/cfg1_bytes.$rep.err
  timeout -k 10 300 python3 bench.py --no-level0-compaction --workload cfg1 --distinct 32 --no-cpu-baseline --no-reference-termination > $O/cfg1_fp64.$rep.json 2> $O/cfg1_fp64.$rep.err
  python3 tools/benchsum.py $O/cfg1_fp64.$rep.json "cfg1 fp64 intensities .$rep" || tail -3 $O/cfg1_fp64.$rep.err
done
timeout -k 10 300 python3 bench.py --workload cfg1 --thresholds shipped --pairs 512 --distinct 32 --no-cpu-baseline > $O/cfg1_shipped.json 2> $O/cfg1_shipped.err
python3 tools/benchsum.py $O/cfg1_shipped.json "cfg1 shipped thresholds" || tail -3 $O/cfg1_shipped.err
timeout -k 10 600 python3 tests/tools/fuzz_parity.py 150 7 big > $O/fuzz_big.txt 2>&1; echo "fuzz big rc=$?"; tail -4 $O/fuzz_big.txt

#!/bin/bash
set -o pipefail
O=gpurun_out/r5e
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "level0 or sliding or wide or full_hd or half_pixel" > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  timeout -k 10 300 python3 bench.py --workload cfg1 --distinct 32 --no-cpu-baseline --no-reference-termination > $O/cfg1_bytes.$rep.json 2> $O/cfg1_bytes.$rep.err
  python3 tools/benchsum.py $O/cfg1_bytes.$rep.json "cfg1 byte intensities .$rep" || tail -3 $O/cfg1_bytes.$rep.err
  timeout -k 10 300 python3 bench.py --no-level0-compaction --workload cfg1 --distinct 32 --no-cpu-baseline --no-reference-termination > $O/cfg1_fp64.$rep.json 2> $O/cfg1_fp64.$rep.err
  python3 tools/benchsum.py $O/cfg1_fp64.$rep.json "cfg1 fp64 intensities .$rep" || tail -3 $O/cfg1_fp64.$rep.err
done
timeout -k 10 300 python3 bench.py --workload cfg5 --pairs 2048 --distinct 32 --no-cpu-baseline --no-reference-termination > $O/cfg5.json 2> $O/cfg5.err
python3 tools/benchsum.py $O/cfg5.json "cfg5" || tail -3 $O/cfg5.err

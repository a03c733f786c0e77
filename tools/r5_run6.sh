#!/bin/bash
set -o pipefail
O=gpurun_out/r5f
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "level0 or sliding or full_hd" > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for t in 0 1 2 3 4 6; do
  PHOVO_SLIDE_TOUCH=$t timeout -k 10 300 python3 tools/bench_with.py tuning --workload cfg1 --distinct 32 --no-cpu-baseline --no-reference-termination > $O/cfg1_bytes_t$t.json 2> $O/cfg1_bytes_t$t.err
  python3 tools/benchsum.py $O/cfg1_bytes_t$t.json "cfg1 bytes touch=$t" || tail -3 $O/cfg1_bytes_t$t.err
  PHOVO_SLIDE_TOUCH=$t timeout -k 10 300 python3 tools/bench_with.py tuning --no-level0-compaction --workload cfg1 --distinct 32 --no-cpu-baseline --no-reference-termination > $O/cfg1_fp64_t$t.json 2> $O/cfg1_fp64_t$t.err
  python3 tools/benchsum.py $O/cfg1_fp64_t$t.json "cfg1 fp64  touch=$t" || tail -3 $O/cfg1_fp64_t$t.err
  PHOVO_SLIDE_TOUCH=$t timeout -k 10 300 python3 tools/bench_with.py tuning --workload cfg5 --pairs 2048 --distinct 32 --no-cpu-baseline --no-reference-termination > $O/cfg5_t$t.json 2> $O/cfg5_t$t.err
  python3 tools/benchsum.py $O/cfg5_t$t.json "cfg5       touch=$t" || tail -3 $O/cfg5_t$t.err
done

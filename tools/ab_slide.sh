#!/bin/bash
# parity of the sliding-window kernel's callers, then an A/B of the in-tree library against library B on the two workloads
# whose top level takes that kernel:   bash tools/ab_slide.sh gpurun_out/<dir> <library B>
set -o pipefail
O=$1; B=$2
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_extensions.py -m gpu -x -q -k "sliding or window or 1280 or large or full_hd or layered or variant" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
run() { name=$1 lib=$2; shift 2
  local prog="bench.py"; [ -n "$lib" ] && prog="tools/bench_with.py $(realpath "$lib")"      # library B only through the tools/ opt-in
  timeout -k 10 300 python3 $prog --no-cpu-baseline --no-reference-termination "$@" > "$O/$name.json" 2> "$O/$name.err"
  python3 tools/benchsum.py "$O/$name.json" "$name" || tail -3 "$O/$name.err"; }
for rep in 1 2; do
  run cfg1_new.$rep "" --workload cfg1
  run cfg1_old.$rep "$B" --workload cfg1
  run cfg5_new.$rep "" --workload cfg5 --pairs 2048
  run cfg5_old.$rep "$B" --workload cfg5 --pairs 2048
done

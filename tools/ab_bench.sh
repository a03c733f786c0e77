#!/bin/bash
# A/B of two builds of libphovo_hip.so on the GPU box (through gpurun, from the repository root):
#     bash tools/ab_bench.sh gpurun_out/r3/ab1 <library B>      (e.g. a build of an earlier commit, or one with -DPHOVO_AB_*)
# Runs bench.py (no CPU legs) in its three diagnostic shapes -- shipped thresholds, fixed iterations, every plane streamed
# once -- with the in-tree library and with library B, and prints one line per run.
set -o pipefail
OUT=$1
B=${2:?library B (a second build of libphovo_hip.so) is required}
mkdir -p "$OUT"
run() {   # name, library ("" = in-tree), bench arguments
  local name=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export PHOVO_HIP_LIBRARY=$(realpath "$lib"); else unset PHOVO_HIP_LIBRARY; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reference-termination "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  python3 tools/benchsum.py "$OUT/$name.json" "$name" || true
}
for rep in 1 2; do
run shipped_new.$rep "" --thresholds shipped
run shipped_old.$rep "$B" --thresholds shipped
run fixed_new.$rep "" 
run fixed_old.$rep "$B"
run once_new.$rep "" --max-iterations 0,0,1,1
run once_old.$rep "$B" --max-iterations 0,0,1,1
done
run shipped2048_new "" --thresholds shipped --pairs 2048
run shipped2048_old "$B" --thresholds shipped --pairs 2048
run cfg3_shipped_new "" --thresholds shipped --workload cfg3
run cfg3_shipped_old "$B" --thresholds shipped --workload cfg3

#!/bin/bash
# A/B of two builds of libphovo_hip.so on the GPU box (through gpurun, from the repository root):
#     bash tools/ab_bench.sh gpurun_out/r4/ab1 <library B> [shapes...]     (e.g. a build of an earlier commit)
# Runs bench.py (no CPU legs) with the in-tree library and with library B, interleaved twice, and prints one line per run.
# shapes: shipped fixed layered cfg3 shipped2048 (default: all)
set -o pipefail
OUT=$1
B=${2:?library B (a second build of libphovo_hip.so) is required}
shift 2
SHAPES=${*:-shipped fixed layered cfg3 shipped2048}
mkdir -p "$OUT"
run() {   # name, library ("" = in-tree), bench arguments
  local name=$1 lib=$2; shift 2
  local prog="bench.py"; [ -n "$lib" ] && prog="tools/bench_with.py $(realpath "$lib")"      # library B only through the tools/ opt-in
  timeout -k 10 300 python3 $prog --no-cpu-baseline --no-reference-termination "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  python3 tools/benchsum.py "$OUT/$name.json" "$name" || tail -3 "$OUT/$name.err"
}
for rep in 1 2; do
  for s in $SHAPES; do
    case $s in
      shipped)     args="--thresholds shipped" ;;
      fixed)       args="" ;;
      layered)     args="--thresholds shipped --scene layered --distinct 128" ;;
      cfg3)        args="--thresholds shipped --workload cfg3" ;;
      shipped2048) args="--thresholds shipped --pairs 2048" ;;
      *) echo "unknown shape $s"; exit 2 ;;
    esac
    run ${s}_new.$rep "" $args
    run ${s}_old.$rep "$B" $args
  done
done

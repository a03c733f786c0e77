#!/bin/bash
# A/B of the bilinear extension between the in-tree library and a second build (through gpurun, from the repository root):
#     bash tools/ab_bilinear.sh gpurun_out/<dir> <library B>
# The extension's GPU tests first (in-tree library), then bench.py --bilinear per storage with each library, interleaved twice.
set -o pipefail
OUT=$1
B=${2:?library B (a second build of libphovo_hip.so) is required}
mkdir -p "$OUT"
timeout -k 10 600 python3 -m pytest tests -q -m gpu -x -k "bilinear or extension or storage" > "$OUT/pytest_ext.log" 2>&1; rc=$?
tail -3 "$OUT/pytest_ext.log"; [ $rc -eq 0 ] || exit $rc
run() {   # name, library ("" = in-tree), bench arguments
  local name=$1 lib=$2; shift 2
  local prog="bench.py"; [ -n "$lib" ] && prog="tools/bench_with.py $(realpath "$lib")"
  timeout -k 10 300 python3 $prog --no-cpu-baseline --no-reference-termination --bilinear "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  python3 tools/benchsum.py "$OUT/$name.json" "$name" || tail -3 "$OUT/$name.err"
}
for rep in 1 2; do
  for st in f64 f32 f16; do
    run ${st}_new.$rep "" --storage $st
    run ${st}_old.$rep "$B" --storage $st
  done
done
timeout -k 10 500 python3 tests/tools/fuzz_parity.py 1500 77 ext > "$OUT/fuzz_ext.log" 2>&1; rc=$?
tail -4 "$OUT/fuzz_ext.log"; exit $rc

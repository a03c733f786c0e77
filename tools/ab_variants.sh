#!/bin/bash
# A/B of several builds of the library on the GPU box: bash tools/ab_variants.sh <out dir> <bench args...> -- lib1 lib2 ...
# ("" or "intree" = the in-tree library).  Two rounds, interleaved, one line per run.
set -o pipefail
OUT=$1; shift
ARGS=()
while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
mkdir -p "$OUT"
for rep in 1 2; do
  for lib in "$@"; do
    name=$(basename "$lib" .so)
    if [ "$lib" = "intree" ]; then unset PHOVO_HIP_LIBRARY; else export PHOVO_HIP_LIBRARY=$(realpath "$lib"); fi
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reference-termination "${ARGS[@]}" > "$OUT/$name.$rep.json" 2> "$OUT/$name.$rep.err"
    python3 tools/benchsum.py "$OUT/$name.$rep.json" "$name.$rep" || tail -3 "$OUT/$name.$rep.err"
  done
done

#!/bin/bash
# A/B of one environment switch on the GPU box: bash tools/ab_env.sh <out dir> VAR "v1 v2 ..." <bench args...>
set -o pipefail
OUT=$1; VAR=$2; VALS=$3; shift 3
mkdir -p "$OUT"
for rep in 1 2; do
  for v in $VALS; do
    export $VAR=$v
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reference-termination "$@" > "$OUT/$VAR.$v.$rep.json" 2> "$OUT/$VAR.$v.$rep.err"
    python3 tools/benchsum.py "$OUT/$VAR.$v.$rep.json" "$VAR=$v.$rep" || tail -3 "$OUT/$VAR.$v.$rep.err"
  done
done

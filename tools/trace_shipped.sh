#!/bin/bash
# Kernel trace of bench.py's shipped-threshold mode (every launch of the capped levels listed in dispatch order):
#     bash tools/trace_shipped.sh gpurun_out/r3/trace_new [library] [extra bench.py arguments]
# then  python tools/shipped_summary.py gpurun_out/r3/trace_new
set -o pipefail
OUT=$(realpath -m "$1"); shift
LIB=$1; shift
ROOT=$(pwd)
mkdir -p "$OUT"
if [ -n "$LIB" ]; then export PHOVO_HIP_LIBRARY=$(realpath "$LIB"); fi
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --thresholds shipped $*"
echo "$BENCH" > "$OUT/command.txt"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1
echo "trace rc=$?"

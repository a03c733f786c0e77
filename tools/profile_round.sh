#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repository root):
#     bash tools/profile_round.sh gpurun_out/prof_r01 [extra bench.py arguments, e.g. --max-iterations 0,0,1,1]
# then, back in the container:  python tools/parse_rocprof.py gpurun_out/prof_r01 profiles r01_final --current
# PROFILE_SKIP_SQ=1 leaves the SQ pass out, PROFILE_SKIP_CAL=1 the two calibration passes.
# Trace and counters are separate passes (never --pmc together with another trace domain); the profiled program
# itself follows `--` (python3, no env/bash hop); rocprofv3 runs from /tmp with TMPDIR=/tmp.
set -e -o pipefail
OUT=$(realpath -m "$1")
shift
ROOT=$(pwd)
mkdir -p "$OUT"
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-reference-termination --distinct 32 $*"      # (a later --distinct among the extra arguments wins)
echo "$BENCH" > "$OUT/command.txt"
sha256sum "$ROOT/photoconsistency-visual-odometry_amd/libphovo_hip.so" > "$OUT/library.sha256"      # which build the counters belong to
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import phovo_amd; from phovo_amd import native; print(native.source_sha256())" > "$OUT/source.sha256"      # ... and which sources
cd /tmp
export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
if [ -z "$PROFILE_SKIP_CAL" ]; then
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cal_fetch" -- python3 $ROOT/tools/pmc_calibrate.py > "$OUT/cal_fetch.log" 2>&1
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/cal_write" -- python3 $ROOT/tools/pmc_calibrate.py > "$OUT/cal_write.log" 2>&1
fi
# where the wave cycles go (issue vs parked), VALU share and the effective clock: one SQ pass + GRBM
[ -n "$PROFILE_SKIP_SQ" ] || timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1
# keep what travels back small: the per-dispatch traces are not needed, the stats and counter tables are
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
echo "profiles collected under $OUT"

set -e
OUT=$(realpath -m gpurun_out/r2/sq_c5_slide)
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
PHOVO_SLIDE_GEOM=512 timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 $ROOT/bench.py --workload cfg5 --pairs 2048 --steps 4 --warmup 2 --no-cpu-baseline --no-reference-termination > $OUT/log.txt 2>&1
PHOVO_SLIDE_GEOM=512 timeout -k 10 240 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/pmc_sq2" -- python3 $ROOT/bench.py --workload cfg5 --pairs 2048 --steps 4 --warmup 2 --no-cpu-baseline --no-reference-termination > $OUT/log2.txt 2>&1 || true

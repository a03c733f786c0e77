#!/bin/bash
# round 4, first GPU call: MFMA f64 probe, the fused-launch tests, A/B of the fused launch against the round-3 library
set -o pipefail
O=gpurun_out/r4a
mkdir -p $O
./tools/probes/mfma_f64_probe > $O/mfma_probe.txt 2>&1; head -6 $O/mfma_probe.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_fused.py -m gpu -x -q > $O/pytest_fused.log 2>&1; echo "pytest fused rc=$?"; tail -5 $O/pytest_fused.log
bash tools/ab_bench.sh $O/ab photoconsistency-visual-odometry_amd/libphovo_hip_r3.so shipped layered cfg3 fixed 2>&1 | tee $O/ab.txt

#!/bin/bash
# round 4, second GPU call: fused + pipelining tests, the whole GPU suite, A/B against the round-3 library
set -o pipefail
O=gpurun_out/r4b
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_fused.py -m gpu -x -q > $O/pytest_fused.log 2>&1; rc=$?; echo "pytest fused rc=$rc"; tail -5 $O/pytest_fused.log
[ $rc -ne 0 ] && exit $rc
bash tools/ab_bench.sh $O/ab photoconsistency-visual-odometry_amd/libphovo_hip_r3.so shipped layered cfg3 2>&1 | tee $O/ab.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -5 $O/pytest_gpu.log

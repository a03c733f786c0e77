#!/bin/bash
# round 4, third GPU call: instruction trims of the scatter kernel -- parity, then A/B against the build before them
set -o pipefail
O=gpurun_out/r4c
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
bash tools/ab_bench.sh $O/ab photoconsistency-visual-odometry_amd/libphovo_hip_base.so fixed shipped cfg3 2>&1 | tee $O/ab.txt

#!/bin/bash
# parity of the scatter kernels, then an A/B of the in-tree library against library B on the headline and the shipped mode
set -o pipefail
O=$1; B=$2
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
bash tools/ab_bench.sh $O/ab $B fixed shipped cfg3 2>&1 | tee $O/ab.txt

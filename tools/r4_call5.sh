#!/bin/bash
set -o pipefail
O=gpurun_out/r4e
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_extensions.py -m gpu -x -q > $O/pytest_ext.log 2>&1; rc=$?; echo "pytest ext rc=$rc"; tail -5 $O/pytest_ext.log
[ $rc -ne 0 ] && exit $rc
b() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reference-termination "$@" > $O/$name.json 2> $O/$name.err; python3 tools/benchsum.py $O/$name.json $name || tail -5 $O/$name.err; }
for rep in 1 2; do
b bil_lds.$rep --bilinear
b bil_gather.$rep --bilinear --bilinear-gather
b bil16_lds.$rep --bilinear --storage f16
b bil16_gather.$rep --bilinear --storage f16 --bilinear-gather
done
b bil32_lds --bilinear --storage f32
b cfg5_bil --workload cfg5 --pairs 2048 --storage f16 --huber 0.05 --bilinear

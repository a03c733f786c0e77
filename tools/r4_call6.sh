#!/bin/bash
set -o pipefail
O=gpurun_out/r4f
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest gpu rc=$rc"; tail -6 $O/pytest_gpu.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reference-termination "$@" > $O/$name.json 2> $O/$name.err; python3 tools/benchsum.py $O/$name.json $name || tail -5 $O/$name.err; }
b bil --bilinear
b bil16 --bilinear --storage f16
exit $rc

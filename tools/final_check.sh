#!/bin/bash
# What the driver runs at round end, on the GPU box (through gpurun): pytest -m gpu, the default bench.py line, smoke().
set -o pipefail
O=${1:-gpurun_out/final_check}
mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest gpu rc=$rc"; tail -4 $O/pytest_gpu.log
timeout -k 10 400 python3 bench.py --steps 20 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; python3 tools/benchsum.py $O/bench_default.json default || tail -5 $O/bench_default.err
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
exit $rc

#!/bin/bash
# round 5, second GPU session: whole GPU suite on the new defaults (one arithmetic per pair), default bench line, multi-rank rehearsal at the headline shape
set -o pipefail
O=gpurun_out/r5b
mkdir -p $O
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
echo "pytest gpu rc=$rc"; tail -5 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 tools/benchsum.py $O/bench_default.json default || tail -3 $O/bench_default.err
bash tools/rehearse_dist.sh $O/dist
timeout -k 10 300 python3 tools/timeline.py > $O/timeline_1024.txt 2>&1; grep -E "^==|device time|idle|busy" $O/timeline_1024.txt

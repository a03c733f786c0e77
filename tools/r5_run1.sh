#!/bin/bash
# round 5, first GPU session: the fused launch with long pairs set aside -- parity, timeline, bench A/B
set -o pipefail
O=gpurun_out/r5a
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused.py -m gpu -x -q > $O/pytest_fused.log 2>&1; rc=$?
echo "pytest fused rc=$rc"; tail -3 $O/pytest_fused.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 tools/timeline.py --probe 0 4 6 8 > $O/timeline_1024.txt 2>&1; echo "timeline rc=$?"; grep -E "^==|device time|idle" $O/timeline_1024.txt
timeout -k 10 300 python3 tools/timeline.py --distinct 32 --probe 0 6 > $O/timeline_32.txt 2>&1; echo "timeline32 rc=$?"; grep -E "^==|device time|idle" $O/timeline_32.txt
for k in 0 4 6 8 12; do
  PHOVO_PROBE_ITERATIONS=$k timeout -k 10 300 python3 tools/bench_with.py tuning --thresholds shipped --no-cpu-baseline --steps 20 --pipeline off > $O/shipped_k$k.json 2> $O/shipped_k$k.err
  python3 tools/benchsum.py $O/shipped_k$k.json "shipped one-at-a-time K=$k" || tail -3 $O/shipped_k$k.err
done
for k in 0 6; do
  PHOVO_PROBE_ITERATIONS=$k timeout -k 10 300 python3 tools/bench_with.py tuning --thresholds shipped --no-cpu-baseline --steps 20 --pipeline on > $O/shipped_pipe_k$k.json 2> $O/shipped_pipe_k$k.err
  python3 tools/benchsum.py $O/shipped_pipe_k$k.json "shipped pipelined K=$k" || tail -3 $O/shipped_pipe_k$k.err
  PHOVO_PROBE_ITERATIONS=$k timeout -k 10 300 python3 tools/bench_with.py tuning --thresholds shipped --no-cpu-baseline --steps 20 --pipeline off --scene layered --distinct 128 > $O/layered_k$k.json 2> $O/layered_k$k.err
  python3 tools/benchsum.py $O/layered_k$k.json "layered one-at-a-time K=$k" || tail -3 $O/layered_k$k.err
  PHOVO_PROBE_ITERATIONS=$k timeout -k 10 300 python3 tools/bench_with.py tuning --thresholds shipped --no-cpu-baseline --steps 20 --pipeline off --workload cfg3 > $O/cfg3_k$k.json 2> $O/cfg3_k$k.err
  python3 tools/benchsum.py $O/cfg3_k$k.json "cfg3 one-at-a-time K=$k" || tail -3 $O/cfg3_k$k.err
done
timeout -k 10 400 python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python3 tools/benchsum.py $O/bench_default.json default || tail -3 $O/bench_default.err

#!/bin/bash
# The profiling session of round 3 on the GPU box (through gpurun, from the repository root):
#     bash tools/profile_r03.sh gpurun_out/r3f
# then, back in the container:  bash tools/profile_r03.sh --parse gpurun_out/r3f     (writes profiles/r03_*)
set -o pipefail
if [ "$1" = "--parse" ]; then
  O=$2
  python3 tools/parse_rocprof.py $O/prof_full profiles r03_full --current > /dev/null
  python3 tools/parse_rocprof.py $O/prof_once profiles r03_stream_once --calibration profiles/r03_full_pmc_traffic.json > /dev/null
  python3 tools/parse_rocprof.py $O/prof_cfg5 profiles r03_cfg5 --pairs 2048 --calibration profiles/r03_full_pmc_traffic.json > /dev/null
  python3 tools/shipped_profile.py $O/prof_shipped $O/bench_shipped.json profiles r03_shipped --calibration profiles/r03_full_pmc_traffic.json
  python3 tools/shipped_profile.py $O/prof_cfg3_shipped $O/bench_cfg3_shipped.json profiles r03_cfg3_shipped --calibration profiles/r03_full_pmc_traffic.json
  python3 tools/shipped_profile.py $O/prof_layered $O/bench_layered.json profiles r03_shipped_layered --calibration profiles/r03_full_pmc_traffic.json
  exit 0
fi
O=$1
mkdir -p $O
bash tools/profile_round.sh $O/prof_full || exit 1
timeout -k 10 200 python3 bench.py --no-cpu-baseline --thresholds shipped > $O/bench_shipped.json 2> $O/bench_shipped.err || exit 1
PROFILE_SKIP_CAL=1 bash tools/profile_round.sh $O/prof_shipped --thresholds shipped || exit 1
timeout -k 10 200 python3 bench.py --no-cpu-baseline --thresholds shipped --workload cfg3 > $O/bench_cfg3_shipped.json 2> $O/bench_cfg3_shipped.err || exit 1
PROFILE_SKIP_CAL=1 bash tools/profile_round.sh $O/prof_cfg3_shipped --thresholds shipped --workload cfg3 || exit 1
# the layered desk-like scene, 128 distinct pairs: the shipped thresholds on data with depth discontinuities and invalid regions
timeout -k 10 250 python3 bench.py --no-cpu-baseline --thresholds shipped --scene layered --distinct 128 > $O/bench_layered.json 2> $O/bench_layered.err || exit 1
PROFILE_SKIP_CAL=1 PROFILE_SKIP_SQ=1 bash tools/profile_round.sh $O/prof_layered --thresholds shipped --scene layered --distinct 128 || exit 1
PROFILE_SKIP_CAL=1 PROFILE_SKIP_SQ=1 bash tools/profile_round.sh $O/prof_once --max-iterations 0,0,1,1 || exit 1
PROFILE_SKIP_CAL=1 bash tools/profile_round.sh $O/prof_cfg5 --workload cfg5 --pairs 2048 || exit 1
echo "profiling session done"

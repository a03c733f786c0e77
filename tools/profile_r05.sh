#!/bin/bash
# The profiling sessions of round 5 on the GPU box (through gpurun, from the repository root), in two calls:
#     bash tools/profile_r05.sh gpurun_out/r4p a        (headline: six passes + stream-once; shipped thresholds: plane, layered, 5-level)
#     bash tools/profile_r05.sh gpurun_out/r4p b        (cfg5, cfg1, bilinear extension)
# then, back in the container:  bash tools/profile_r05.sh --parse gpurun_out/r4p     (writes profiles/r05_*)
set -o pipefail
if [ "$1" = "--parse" ]; then
  O=$2
  P="python3 tools/parse_rocprof.py"
  $P $O/prof_full profiles r05_full --current --bench $O/bench_default.json > /dev/null
  CAL="--calibration profiles/r05_full_pmc_traffic.json"
  $P $O/prof_once profiles r05_stream_once $CAL > /dev/null
  $P $O/prof_shipped profiles r05_shipped $CAL --bench $O/bench_shipped.json > /dev/null
  $P $O/prof_layered profiles r05_shipped_layered $CAL --bench $O/bench_layered.json > /dev/null
  $P $O/prof_cfg3_shipped profiles r05_cfg3_shipped $CAL --bench $O/bench_cfg3_shipped.json > /dev/null
  [ -d $O/prof_cfg5 ] && $P $O/prof_cfg5 profiles r05_cfg5 --pairs 2048 $CAL --bench $O/bench_cfg5.json > /dev/null
  [ -d $O/prof_cfg1 ] && $P $O/prof_cfg1 profiles r05_cfg1 --pairs 512 $CAL --bench $O/bench_cfg1.json > /dev/null
  [ -d $O/prof_bilinear ] && $P $O/prof_bilinear profiles r05_bilinear $CAL --bench $O/bench_bilinear.json > /dev/null
  [ -d $O/prof_bilinear_f16 ] && $P $O/prof_bilinear_f16 profiles r05_bilinear_f16 $CAL --bench $O/bench_bilinear_f16.json > /dev/null
  [ -d $O/prof_bilinear_f32 ] && $P $O/prof_bilinear_f32 profiles r05_bilinear_f32 $CAL --bench $O/bench_bilinear_f32.json > /dev/null
  mkdir -p profiles/r05_runs
  cp $O/bench_*.json profiles/r05_runs/ 2>/dev/null
  exit 0
fi
O=$1
mkdir -p $O
bench() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -5 $O/bench_$name.err; exit 1; }; python3 tools/benchsum.py $O/bench_$name.json $name; }
if [ "$2" = "a" ]; then
  bench default --cpu-seconds 8
  bash tools/profile_round.sh $O/prof_full || exit 1
  PROFILE_SKIP_CAL=1 PROFILE_SKIP_SQ=1 bash tools/profile_round.sh $O/prof_once --max-iterations 0,0,1,1 || exit 1
  bench shipped --no-cpu-baseline --thresholds shipped
  PROFILE_SKIP_CAL=1 bash tools/profile_round.sh $O/prof_shipped --thresholds shipped --pipeline off --distinct 1024 || exit 1
  bench layered --no-cpu-baseline --thresholds shipped --scene layered --distinct 128
  PROFILE_SKIP_CAL=1 PROFILE_SKIP_SQ=1 bash tools/profile_round.sh $O/prof_layered --thresholds shipped --pipeline off --scene layered --distinct 128 || exit 1
  bench cfg3_shipped --no-cpu-baseline --thresholds shipped --workload cfg3
  PROFILE_SKIP_CAL=1 PROFILE_SKIP_SQ=1 bash tools/profile_round.sh $O/prof_cfg3_shipped --thresholds shipped --pipeline off --workload cfg3 --distinct 1024 || exit 1
else
  bench cfg5 --no-cpu-baseline --workload cfg5 --pairs 2048
  PROFILE_SKIP_CAL=1 bash tools/profile_round.sh $O/prof_cfg5 --workload cfg5 --pairs 2048 || exit 1
  bench cfg1 --cpu-seconds 6 --cpu-threads 1 --workload cfg1
  PROFILE_SKIP_CAL=1 bash tools/profile_round.sh $O/prof_cfg1 --workload cfg1 || exit 1
  bench bilinear --no-cpu-baseline --bilinear
  PROFILE_SKIP_CAL=1 bash tools/profile_round.sh $O/prof_bilinear --bilinear || exit 1
  bench bilinear_f16 --no-cpu-baseline --bilinear --storage f16
  PROFILE_SKIP_CAL=1 bash tools/profile_round.sh $O/prof_bilinear_f16 --bilinear --storage f16 || exit 1
  bench bilinear_f32 --no-cpu-baseline --bilinear --storage f32
  PROFILE_SKIP_CAL=1 PROFILE_SKIP_SQ=1 bash tools/profile_round.sh $O/prof_bilinear_f32 --bilinear --storage f32 || exit 1
  bench cfg5_f16 --no-cpu-baseline --workload cfg5 --pairs 2048 --storage f16 --huber 0.05
  bench cfg3 --no-cpu-baseline --workload cfg3
fi
echo "profiling session $2 done"

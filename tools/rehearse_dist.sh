#!/bin/bash
# Rehearsals of bench.py's multi-rank path on the one-GPU box (no scaling number comes out of these):
#   one rank under torch.distributed.run with RCCL (init, barrier, all_reduce, all_gather from the engine's device buffer) at
#   the HEADLINE shape -- default pairs (8192), 20 steps -- next to the plain one-process run of the same command, so that the
#   per-rank overhead of the distributed path (one all_gather per step) is a measured number: target <= 1 % apart;
#   five gloo ranks sharing the card (the box allows six processes on it).
set -o pipefail
O=${1:-gpurun_out/r5_dist}
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
  timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --distinct 32 --no-cpu-baseline --no-reference-termination > $O/bench_plain.$rep.json 2> $O/bench_plain.$rep.err
  echo "plain.$rep rc=$?"; python3 tools/benchsum.py $O/bench_plain.$rep.json plain.$rep || tail -5 $O/bench_plain.$rep.err
  PHOVO_BENCH_FORCE_DIST=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$rep \
    bench.py --gpus 1 --steps 20 --warmup 5 --distinct 32 --no-cpu-baseline --no-reference-termination > $O/bench_nccl1.$rep.json 2> $O/bench_nccl1.$rep.err
  echo "nccl1.$rep rc=$?"; python3 tools/benchsum.py $O/bench_nccl1.$rep.json nccl1.$rep || tail -5 $O/bench_nccl1.$rep.err
done
PHOVO_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 5 --steps 5 --warmup 1 --pairs 1024 --distinct 32 --no-cpu-baseline > $O/bench_gloo5.json 2> $O/bench_gloo5.err
echo "gloo5 rc=$?"; python3 tools/benchsum.py $O/bench_gloo5.json gloo5 || tail -5 $O/bench_gloo5.err

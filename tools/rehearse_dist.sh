#!/bin/bash
# Rehearsals of bench.py's multi-rank path on the one-GPU box (no scaling number comes out of these):
#   one rank under torch.distributed.run with RCCL (init, barrier, all_reduce, all_gather from the engine's device buffer),
#   five gloo ranks sharing the card (the box allows six processes on it).
set -o pipefail
O=${1:-gpurun_out/r4d_dist}
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
PHOVO_BENCH_FORCE_DIST=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 5 --warmup 1 --pairs 2048 --no-cpu-baseline > $O/bench_nccl1.json 2> $O/bench_nccl1.err
echo "nccl1 rc=$?"; python3 tools/benchsum.py $O/bench_nccl1.json nccl1 || tail -5 $O/bench_nccl1.err
PHOVO_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 5 --steps 5 --warmup 1 --pairs 1024 --no-cpu-baseline > $O/bench_gloo5.json 2> $O/bench_gloo5.err
echo "gloo5 rc=$?"; python3 tools/benchsum.py $O/bench_gloo5.json gloo5 || tail -5 $O/bench_gloo5.err

#!/bin/bash
# Timing of bench.py workloads with a probe build of the library next to the in-tree one (through gpurun, from the repository
# root):   bash tools/probe_ab.sh gpurun_out/<dir> <probe library> <workload> [bench arguments]
# A probe build may compute wrong results on purpose (it answers "what would it cost / save"); nothing here checks them.
set -o pipefail
O=$1; B=${2:?probe library}; shift 2
mkdir -p $O
for rep in 1 2; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-reference-termination "$@" > $O/tree.$rep.json 2> $O/tree.$rep.err
  python3 tools/benchsum.py $O/tree.$rep.json tree.$rep || tail -3 $O/tree.$rep.err
  timeout -k 10 300 python3 tools/bench_with.py $(realpath $B) --no-cpu-baseline --no-reference-termination "$@" > $O/probe.$rep.json 2> $O/probe.$rep.err
  python3 tools/benchsum.py $O/probe.$rep.json probe.$rep || tail -3 $O/probe.$rep.err
done

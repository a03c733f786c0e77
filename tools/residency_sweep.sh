#!/bin/bash
# What would the sliding-window kernel deliver if a launch's resident planes fitted the Infinity Cache?  The same kernel on
# FEWER pairs than CUs: with p <= 256 pairs every pair has a CU to itself and p x (planes of two frames) are resident, so the
# per-CU rate at p = 16 ... 64 (resident set below 256 MiB) against p >= 256 (streams from HBM) is the upper bound of what
# several CUs per pair could buy on levels above ~39 k pixels -- before any cost of cutting a pair across workgroups.
#     bash tools/residency_sweep.sh gpurun_out/<dir>
set -o pipefail
O=${1:?output directory}
mkdir -p $O
for p in 32 64 96 128 192 256 512 2048; do
  timeout -k 10 300 python3 bench.py --workload cfg5 --pairs $p --distinct 32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-termination > $O/cfg5_p$p.json 2> $O/cfg5_p$p.err
  python3 tools/benchsum.py $O/cfg5_p$p.json "cfg5 pairs=$p" || tail -3 $O/cfg5_p$p.err
done
for p in 8 16 32 64 128 256 512; do
  timeout -k 10 300 python3 bench.py --workload cfg1 --pairs $p --distinct 32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-termination > $O/cfg1_p$p.json 2> $O/cfg1_p$p.err
  python3 tools/benchsum.py $O/cfg1_p$p.json "cfg1 pairs=$p" || tail -3 $O/cfg1_p$p.err
done
python3 - "$O" <<'PY'
import glob, json, os, sys
O = sys.argv[1]
print("# workload pairs | kernel threads | launch ms | GB/s algorithmic | per-CU share of 8 TB/s used (pairs <= 256: pairs/256 of the chip) | resident MB")
for wl, level_px in (("cfg5", 76800), ("cfg1", 307200)):
    for f in sorted(glob.glob(os.path.join(O, f"{wl}_p*.json")), key=lambda s: int(s.split("_p")[-1].split(".")[0])):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        p = d["config"]["pairs_per_gpu"]
        for l in d["roofline"]["launches"]:
            if l["pixels"][0] != level_px:
                continue
            share = min(p, 256) / 256.0
            print(f"{wl} {p:5d} | {l['kernel'][:34]:34s} {l['threads']:4d} | {l['avg_launch_ms']:8.3f} | {l['achieved_GBs']:8.1f} | "
                  f"{l['achieved_GBs'] / (8000.0 * share):6.3f} | {min(p, 256) * level_px * 40 / 1e6:8.1f}")
PY

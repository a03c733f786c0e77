"""Per-launch summary of a kernel trace of the shipped-threshold mode (tools/trace_shipped.sh, tools/profile_round.sh with
`--thresholds shipped`): the Gauss-Newton launches of one bench step in dispatch order, averaged over the timed steps.

    python tools/shipped_summary.py <dir with stats/> [--steps 10] [--warmup 2] [--json out.json]

A capped level is two or three launches of (possibly) different instantiations; the stats CSV merges launches of one
instantiation across levels, so this reads the per-dispatch trace instead and attributes every launch to its place in
the step (coarse level first, then its follow-up launches, then the next level ...)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    src = sys.argv[1]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 10
    f = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)
    if not f:
        raise SystemExit("no *_kernel_trace.csv under " + src)
    rows = [r for r in csv.DictReader(open(f[0])) if "gn_level_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # launches per step = total / (warm-up + timed + the reference-termination leg's steps ...): find the period as the
    # shortest prefix of kernel names that repeats through the whole list
    names = [(r["Kernel_Name"], r["Grid_Size_X"]) for r in rows]
    period = next(p for p in range(1, len(names) + 1) if len(names) % p == 0 and names == names[:p] * (len(names) // p))
    n_steps = len(names) // period
    timed = rows[-steps * period:] if n_steps >= steps else rows
    acc = defaultdict(list)
    gaps = defaultdict(list)
    for s in range(len(timed) // period):
        for j in range(period):
            r = timed[s * period + j]
            acc[j].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            if j:
                gaps[j].append(int(r["Start_Timestamp"]) - int(timed[s * period + j - 1]["End_Timestamp"]))
    out = []
    for j in range(period):
        r = timed[j]
        name = r["Kernel_Name"]
        short = name[name.index("gn_level_kernel"):].split("(")[0]
        out.append(dict(position=j, kernel=short, grid_size=int(r["Grid_Size_X"]), workgroup_size=int(r["Workgroup_Size_X"]),
                        mean_us=sum(acc[j]) / len(acc[j]) / 1e3, min_us=min(acc[j]) / 1e3, max_us=max(acc[j]) / 1e3,
                        gap_before_us=(sum(gaps[j]) / len(gaps[j]) / 1e3) if j else None))
    total = sum(o["mean_us"] for o in out)
    res = dict(launches_per_step=period, steps_seen=n_steps, steps_averaged=len(timed) // period, launches=out,
               sum_of_launches_us=total)
    for o in out:
        print(f'{o["position"]:2d} {o["kernel"]:64s} grid {o["grid_size"]:7d} wg {o["workgroup_size"]:5d} '
              f'mean {o["mean_us"]:9.1f} us  (min {o["min_us"]:.1f}, max {o["max_us"]:.1f})'
              + (f'  gap {o["gap_before_us"]:.1f}' if o["gap_before_us"] is not None else ""))
    print(f"sum {total:.1f} us per step over {period} launches; {res['steps_averaged']} steps averaged of {n_steps} seen")
    if "--json" in sys.argv:
        json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()

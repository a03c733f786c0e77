"""Per-launch profile of an enqueue with data-dependent termination (bench.py --thresholds shipped): every launch of
the capped levels attributed to its place in the step, with its share of the algorithmic bytes.

    python tools/shipped_profile.py <dir of tools/profile_round.sh ... --thresholds shipped> <bench json of the same
           command> profiles <tag> [--calibration profiles/<tag of a session with cal passes>_pmc_traffic.json] [--steps 10]

A capped level is two or three launches, possibly of the same instantiation as another level's (the 1024-thread form
finishes the long pairs of every level), and consecutive levels overlap on two streams -- so neither the stats CSV
(merged per instantiation) nor start-time order says which launch is which.  The host enqueues the launches of a step
in a fixed order, though, and rocprofv3 numbers dispatches in that order: the launches of a step are `period`
consecutive Dispatch_Ids of gn_level_kernel*, `period` = len(bench json "launches") (bench.py restates the engine's
schedule and counts the pair-iterations of every launch from the iteration counts the pairs reported).

Inputs: stats/ (--kernel-trace --stats), pmc_fetch/, pmc_write/, pmc_sq/ (separate --pmc passes; in those passes the
profiler runs one kernel at a time, so their durations are NOT the overlapped ones -- durations come from stats/ only).
Output: <tag>_launches.json and a table on stdout; <tag>_kernel_stats.csv is the stats CSV, copied."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

HBM_PEAK = 8000.0e9


def trace_rows(d):
    f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    if not f:
        return []
    rows = [r for r in csv.DictReader(open(f[0])) if "gn_level_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def counter_rows(d):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    if not f:
        return {}
    per = defaultdict(dict)                  # dispatch id -> {counter: value}, kernel name
    names = {}
    for r in csv.DictReader(open(f[0])):
        if "gn_level_kernel" not in r["Kernel_Name"]:
            continue
        i = int(r["Dispatch_Id"])
        per[i][r["Counter_Name"]] = per[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[i] = r["Kernel_Name"]
    return per, names


def short(name):
    i = name.index("gn_level_kernel")
    return name[i:].split("(")[0]


def by_position(ids, period, steps):
    """ids: dispatch ids in order; returns {position: [ids of the last `steps` steps]}"""
    n_steps = len(ids) // period
    use = ids[(n_steps - min(steps, n_steps)) * period:n_steps * period]
    return {j: use[j::period] for j in range(period)}, n_steps


def main():
    src, bench_json, dst, tag = sys.argv[1:5]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 10
    cal_file = sys.argv[sys.argv.index("--calibration") + 1] if "--calibration" in sys.argv else None
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    launches = bench.get("launches") or (bench.get("reference_termination") or {}).get("launches")
    if not launches:
        raise SystemExit("the bench json has no 'launches' (run bench.py --thresholds shipped)")
    period = len(launches)
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))

    rows = trace_rows(os.path.join(src, "stats"))
    if len(rows) % period:
        raise SystemExit(f"{len(rows)} gn_level_kernel dispatches are not a multiple of {period} launches per step")
    by_id = {int(r["Dispatch_Id"]): r for r in rows}
    pos, n_steps = by_position([int(r["Dispatch_Id"]) for r in rows], period, steps)
    ff = wf = 1.0
    cal = None
    if cal_file and os.path.exists(cal_file):
        cal = json.load(open(cal_file)).get("calibration")
        ff = (cal.get("FETCH_SIZE") or {}).get("factor") or 1.0
        wf = (cal.get("WRITE_SIZE") or {}).get("factor") or 1.0
    pmc = {}
    for key, d in (("fetch", "pmc_fetch"), ("write", "pmc_write"), ("sq", "pmc_sq")):
        got = counter_rows(os.path.join(src, d))
        if got:
            per, names = got
            ids = sorted(per)
            if len(ids) % period == 0:
                p, _ = by_position(ids, period, steps)
                pmc[key] = (per, names, p)

    # the span of a step on the device: first launch's start to the last end (launches overlap on two streams)
    spans = []
    ids_sorted = [int(r["Dispatch_Id"]) for r in rows]
    for s in range(n_steps - min(steps, n_steps), n_steps):
        grp = [by_id[i] for i in ids_sorted[s * period:(s + 1) * period]]
        spans.append(max(int(r["End_Timestamp"]) for r in grp) - min(int(r["Start_Timestamp"]) for r in grp))
    out = dict(tag=tag, command=open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else None,
               launches_per_step=period, steps_seen=n_steps, steps_averaged=min(steps, n_steps),
               calibration=cal, step_span_us=sum(spans) / len(spans) / 1e3, launches=[])
    total_bytes = 0.0
    for j, work in enumerate(launches):
        grp = [by_id[i] for i in pos[j]]
        name = short(grp[0]["Kernel_Name"])
        if any(short(r["Kernel_Name"]) != name for r in grp):
            raise SystemExit(f"position {j}: different kernels in different steps -- the schedule is not what bench.py restated")
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp]
        mean_ns = sum(dur) / len(dur)
        e = dict(position=j, level=work["level"], launch=work["launch"], kernel=name,
                 workgroups=int(grp[0]["Grid_Size_X"]) // int(grp[0]["Workgroup_Size_X"]),
                 threads=int(grp[0]["Workgroup_Size_X"]), stream=grp[0].get("Stream_Id"),
                 pairs=work["pairs"], pair_iterations=work["pair_iterations"], algorithmic_bytes=work["algorithmic_bytes"],
                 mean_us=mean_ns / 1e3, min_us=min(dur) / 1e3, max_us=max(dur) / 1e3,
                 achieved_GBs=work["algorithmic_bytes"] / mean_ns if mean_ns else 0.0)
        e["frac_of_8TBs"] = e["achieved_GBs"] * 1e9 / HBM_PEAK
        total_bytes += work["algorithmic_bytes"]
        if "fetch" in pmc and "write" in pmc:
            fper, fnames, fpos = pmc["fetch"]
            wper, _, wpos = pmc["write"]
            if short(fnames[fpos[j][0]]) == name:
                fv = sum(fper[i].get("FETCH_SIZE", 0.0) for i in fpos[j]) / len(fpos[j])
                wv = sum(wper[i].get("WRITE_SIZE", 0.0) for i in wpos[j]) / len(wpos[j])
                e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"] = fv, wv
                e["hbm_bytes_per_launch"] = fv * 1024.0 * ff + wv * 1024.0 * wf
                if work["algorithmic_bytes"] > 0:
                    e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / work["algorithmic_bytes"]
        if "sq" in pmc:
            sper, snames, spos = pmc["sq"]
            if short(snames[spos[j][0]]) == name:
                m = defaultdict(float)
                for i in spos[j]:
                    for c, v in sper[i].items():
                        m[c] += v / len(spos[j])
                wc = m.get("SQ_WAVE_CYCLES", 0.0)
                if wc > 0:
                    e["sq"] = {"parked (SQ_WAIT_ANY)": m["SQ_WAIT_ANY"] / wc, "issue stall (SQ_WAIT_INST_ANY)": m["SQ_WAIT_INST_ANY"] / wc,
                               "issuing (SQ_ACTIVE_INST_ANY)": m["SQ_ACTIVE_INST_ANY"] / wc,
                               "issuing VALU (SQ_ACTIVE_INST_VALU)": m["SQ_ACTIVE_INST_VALU"] / wc,
                               "SQ_INSTS_VALU": m.get("SQ_INSTS_VALU", 0.0)}
        out["launches"].append(e)
    out["algorithmic_bytes_per_step"] = total_bytes
    out["whole_step_achieved_GBs"] = total_bytes / (out["step_span_us"] * 1e3)
    out["whole_step_frac_of_8TBs"] = out["whole_step_achieved_GBs"] * 1e9 / HBM_PEAK
    # per level: the launches of a level overlap with the next level's, so a level has no wall time of its own; the sum of
    # its launches' durations over-counts the overlapped part and is given as such
    lv = defaultdict(lambda: dict(bytes=0.0, sum_us=0.0))
    for e in out["launches"]:
        lv[e["level"]]["bytes"] += e["algorithmic_bytes"]
        lv[e["level"]]["sum_us"] += e["mean_us"]
    out["levels"] = [dict(level=l, algorithmic_bytes=v["bytes"], sum_of_launch_durations_us=v["sum_us"],
                          achieved_GBs_over_that_sum=v["bytes"] / (v["sum_us"] * 1e3) if v["sum_us"] else 0.0)
                     for l, v in sorted(lv.items(), reverse=True)]
    json.dump(out, open(os.path.join(dst, f"{tag}_launches.json"), "w"), indent=1)
    for e in out["launches"]:
        print(f'{e["position"]} L{e["level"]} {e["launch"]:52s} {e["kernel"]:60s} wgs {e["workgroups"]:5d} pairs {e["pairs"]:6d} '
              f'{e["mean_us"]:8.1f} us  {e["achieved_GBs"]:7.0f} GB/s = {e["frac_of_8TBs"]:.2f}'
              + (f'  traffic x{e["traffic_over_algorithmic"]:.2f}' if "traffic_over_algorithmic" in e else "")
              + (f'  VALU {4 * e["sq"]["issuing VALU (SQ_ACTIVE_INST_VALU)"]:.2f} parked {e["sq"]["parked (SQ_WAIT_ANY)"]:.2f}' if "sq" in e else ""))
    print(f'step span {out["step_span_us"]:.1f} us, {total_bytes / 1e9:.2f} GB algorithmic -> {out["whole_step_achieved_GBs"]:.0f} GB/s = '
          f'{out["whole_step_frac_of_8TBs"]:.3f} of 8 TB/s ({out["steps_averaged"]} steps averaged of {n_steps})')


if __name__ == "__main__":
    main()

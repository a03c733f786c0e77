"""Turns the rocprofv3 outputs of one profiling session into the files kept under profiles/:

    python tools/parse_rocprof.py gpurun_out/prof_r01 profiles r01 [--current] [--pairs 8192] [--calibration profiles/<tag>_pmc_traffic.json]

--pairs is the number of frame pairs each profiled launch aligned (bench.py --pairs): the level kernels draw pairs
from a work queue, so the grid size no longer says how many there were.

Inputs (directories written by the commands in profiles/README.md):
  stats/      --kernel-trace --stats                     -> <tag>_kernel_stats.csv (copied)
  pmc_fetch/  --pmc FETCH_SIZE   of the same bench command
  pmc_write/  --pmc WRITE_SIZE   of the same bench command
  cal_fetch/, cal_write/   the same counters on tools/pmc_calibrate.py (a copy kernel of known size)
  pmc_sq/     --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
              SQ_BUSY_CYCLES GRBM_GUI_ACTIVE of the same bench command (optional) -> <tag>_pmc_sq.json:
              where the wave cycles go, VALU instructions per launch and the effective clock
Output: <tag>_pmc_traffic.json (+ pmc_traffic.json for bench.py) with HBM bytes per launch of each
Gauss-Newton level kernel, corrected as MI355X_MICROARCH.md (HBM section) prescribes: the counters are
calibrated on a known byte count in the kernel's own access pattern (8 B per lane, coalesced) and the
measured factor is applied."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def counter_rows(d):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    if not f:
        return []
    return list(csv.DictReader(open(f[0])))


def mean_by_kernel(rows, skip_first=0):
    acc = defaultdict(list)
    for r in rows:
        acc[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: sum(v[skip_first:]) / max(1, len(v[skip_first:])) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def sq_summary(rows, pairs):
    """Per level kernel: mean of every SQ/GRBM counter over the timed dispatches (first two = warm-up) and the
    ratios MI355X_MICROARCH.md names: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (per wave, in
    quad-cycles); effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration."""
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for r in rows:
        if "gn_level_kernel" not in r["Kernel_Name"] and "gn_fused_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[r["Kernel_Name"]][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = []
    for k, d in sorted(acc.items()):
        m = {c: sum(v[2:]) / max(1, len(v[2:])) for c, v in d.items()}
        t = sorted(dur[k].items(), key=lambda kv: int(kv[0]))
        t = [v for _, v in t][2:]
        ns = sum(t) / max(1, len(t))
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        e = dict(kernel=k, pairs=pairs, dispatches=len(t), mean_duration_ns=ns, counters=m)
        if wc > 0:
            e["fraction_of_wave_cycles"] = {
                "parked (SQ_WAIT_ANY: s_waitcnt / barrier)": m.get("SQ_WAIT_ANY", 0.0) / wc,
                "issue stall (SQ_WAIT_INST_ANY)": m.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                "issuing (SQ_ACTIVE_INST_ANY)": m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                "issuing VALU (SQ_ACTIVE_INST_VALU)": m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc,
            }
            # the waves of a SIMD share one vector unit: the per-wave VALU share times the resident waves per SIMD is the
            # unit's busy fraction (4 for the 64- / 256- / 512- / 1024-thread level kernels and the fused kernel in a full
            # launch; 3 for the sliding-window kernel -- 768 threads, one workgroup per CU; 2 before its waves were split into
            # two kinds -- 2 for the bilinear kernel's record form (228 registers) and 3 for its LDS-landing form (168))
            if "gn_level_kernel_slide" in k:
                e["waves_per_simd"] = 3 if "<768" in k else 2
            else:
                e["waves_per_simd"] = 3 if "gn_level_kernel_bilinear_dma" in k else (2 if "gn_level_kernel_bilinear" in k else 4)
            if "gn_level_kernel<64," in k and pairs:
                # one wave per pair, 16 per CU: a launch of fewer than 4096 pairs leaves SIMDs with fewer than four waves
                e["waves_per_simd"] = min(4.0, pairs / 1024.0)
            e["simd_valu_busy"] = e["waves_per_simd"] * m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
        if ns > 0 and "GRBM_GUI_ACTIVE" in m:
            e["effective_clock_GHz"] = m["GRBM_GUI_ACTIVE"] / 8.0 / ns
        out.append(e)
    return out


def launch_table(bench, stats_csv, traffic, sq, path):
    """<tag>_launches.json: every launch of one step as bench.py recorded it (levels, kernel, algorithmic bytes from the
    iteration counts the pairs reported) with the rocprofv3 average duration of that kernel (kernel trace of the same
    command), its HBM-side traffic and vector-unit share (separate --pmc passes), and the fraction of the 8 TB/s roofline.
    `whole_step` takes the un-profiled HIP-event span of bench.py (first launch to last) for the same bytes."""
    import re
    ns = {r["Name"]: (float(r["AverageNs"]), int(r["Calls"])) for r in csv.DictReader(open(stats_csv))} if os.path.exists(stats_csv) else {}
    rows = []
    for l in bench["roofline"]["launches"]:
        fam, threads, types = re.match(r"([\w +]+)<(\d+), \.\.\. (.*)>", l["kernel"]).groups()

        def same(name):
            m = re.search(r"(\w+)<(\d+),", name)
            return bool(m) and m.group(1) == fam and int(m.group(2)) == int(threads) and types in name
        row = dict(levels=l["levels"], kernel=l["kernel"], workgroups=l["workgroups"], algorithmic_bytes=l["algorithmic_bytes"],
                   iterations_per_pair=l["iterations_per_pair"], bench_event_ms=l["avg_launch_ms"])
        shared = sum(1 for o in bench["roofline"]["launches"] if o["kernel"] == l["kernel"])
        # (two instantiations can share family, thread count and storage -- the 1024-thread level kernel with the owner map
        # in LDS and the one with the map in HBM that follows a sliding-window launch, usually with no pair to finish: the
        # launch is the one the trace spent time in)
        def longest(cands, key):
            cands = [c for c in cands]
            return [max(cands, key=key)] if cands else []
        for name, (avg, calls) in longest([(n_, v) for n_, v in ns.items() if same(n_)], lambda c: c[1][0] * c[1][1]):
            if same(name) and shared == 1:
                row.update(rocprof_avg_us=avg / 1e3, rocprof_calls=calls,
                           frac_of_8TBs=l["algorithmic_bytes"] / (avg * 1e-9) / 8e12)
            elif same(name):     # one instantiation, several launches per step (levels): the trace's average mixes them
                row.update(rocprof_avg_us_over_its_launches=avg / 1e3, rocprof_calls=calls,
                           frac_of_8TBs=l["algorithmic_bytes"] / (l["avg_launch_ms"] * 1e-3) / 8e12,
                           frac_from="bench.py's HIP-event span of this launch")
        for kd in longest([k for k in traffic.get("kernels", []) if same(k["kernel"])], lambda k: k["hbm_bytes_per_launch"]):
            if same(kd["kernel"]) and shared == 1:
                row.update(hbm_bytes_per_launch=kd["hbm_bytes_per_launch"],
                           traffic_over_algorithmic=kd["hbm_bytes_per_launch"] / l["algorithmic_bytes"])
        for kd in longest([k for k in (sq or []) if same(k["kernel"])], lambda k: k.get("mean_duration_ns", 0.0)):
            if same(kd["kernel"]) and "simd_valu_busy" in kd:
                row.update(simd_valu_busy=kd["simd_valu_busy"], parked_share=kd["fraction_of_wave_cycles"]["parked (SQ_WAIT_ANY: s_waitcnt / barrier)"],
                           valu_instructions_per_launch=kd["counters"].get("SQ_INSTS_VALU"))
        rows.append(row)
    total = sum(r["algorithmic_bytes"] for r in rows)
    out = dict(command=bench["config"]["workload"], pairs=bench["config"]["pairs_per_gpu"], launches=rows,
               whole_step=dict(algorithmic_bytes=total, bench_value=bench["value"], pipelined=bench.get("pipelined"),
                               all_launches_frac=bench["roofline"]["all_levels_frac"],
                               throughput_frac=total * bench["value"] / bench["config"]["pairs_per_gpu"] / 8e12,
                               one_enqueue_at_a_time=bench.get("one_enqueue_at_a_time")))
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


def build_stamp(src):
    """Which build the counters belong to: the sha256 of the libphovo_hip.so the profiled command loaded (written on the GPU
    box by tools/profile_round.sh) and the commit this tree stands at when the summary is made.  bench.py takes a stored
    profile's traffic for its own `roofline.traffic` only when that sha256 is the loaded library's."""
    import subprocess
    stamp = {}
    f = os.path.join(src, "library.sha256")
    if os.path.exists(f):
        stamp["library_sha256"] = open(f).read().split()[0]
    f = os.path.join(src, "source.sha256")
    if os.path.exists(f):
        stamp["source_sha256"] = open(f).read().split()[0]
    # the commit the tree stood at when the raw data was FIRST summarised is kept beside the raw data, so that summarising
    # it again later (a fix in this parser) does not move the stamp to a commit the counters were not collected on
    kept = os.path.join(src, "commit.json")
    if os.path.exists(kept):
        stamp.update(json.load(open(kept)))
        return stamp
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        stamp["commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "HEAD"], text=True).strip()
        stamp["tree_dirty"] = bool(subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "photoconsistency-visual-odometry_amd/csrc", "include"], text=True).strip())
        json.dump(dict(commit=stamp["commit"], tree_dirty=stamp["tree_dirty"]), open(kept, "w"))
    except Exception:
        pass
    return stamp


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    pairs = int(sys.argv[sys.argv.index("--pairs") + 1]) if "--pairs" in sys.argv else 8192
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))

    # calibration: k_resize_level<double> at level 0 = 640x480 doubles read and written once
    known = 640 * 480 * 8
    cal = {}
    for name, d in (("FETCH_SIZE", "cal_fetch"), ("WRITE_SIZE", "cal_write")):
        m, _ = mean_by_kernel(counter_rows(os.path.join(src, d)))
        vals = [v for (k, g), v in m.items() if "k_resize_level<double>" in k and g == 768 * 480]
        cal[name] = dict(counter_KB=vals[0] if vals else None, known_bytes=known,
                         factor=(known / (vals[0] * 1024.0)) if vals and vals[0] > 0 else None)

    # a session whose calibration passes were skipped (PROFILE_SKIP_CAL=1) takes the factors of the session named by
    # --calibration <its *_pmc_traffic.json> (same box, same call: tools/profile_r03.sh)
    if "--calibration" in sys.argv and not (cal["FETCH_SIZE"]["factor"] and cal["WRITE_SIZE"]["factor"]):
        ref = sys.argv[sys.argv.index("--calibration") + 1]
        cal = json.load(open(ref))["calibration"]
        cal["taken_from"] = os.path.basename(ref)
    out = dict(tag=tag, calibration=cal, kernels=[])
    out.update(build_stamp(src))
    fetch, nf = mean_by_kernel(counter_rows(os.path.join(src, "pmc_fetch")), skip_first=2)
    write, _ = mean_by_kernel(counter_rows(os.path.join(src, "pmc_write")), skip_first=2)
    for (k, g), v in sorted(fetch.items()):
        if "gn_level_kernel" not in k and "gn_fused_kernel" not in k:
            continue
        w = write.get((k, g), 0.0)
        ff = cal["FETCH_SIZE"]["factor"] or 1.0
        wf = cal["WRITE_SIZE"]["factor"] or 1.0
        out["kernels"].append(dict(kernel=k, grid_size=g, dispatches=nf[(k, g)],
                                   FETCH_SIZE_KB=v, WRITE_SIZE_KB=w,
                                   hbm_read_bytes=v * 1024.0 * ff, hbm_write_bytes=w * 1024.0 * wf,
                                   hbm_bytes_per_launch=v * 1024.0 * ff + w * 1024.0 * wf))
    import re
    for kd in out["kernels"]:
        m = re.search(r"gn_(?:level|fused)_kernel\w*<(\d+)", kd["kernel"])
        kd["threads"] = int(m.group(1)) if m else None
        kd["workgroups"] = kd["grid_size"] // kd["threads"] if kd["threads"] else None
        kd["pairs"] = pairs
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    sq = sq_summary(counter_rows(os.path.join(src, "pmc_sq")), pairs)
    if sq:
        json.dump(dict(tag=tag, kernels=sq, **build_stamp(src)), open(os.path.join(dst, f"{tag}_pmc_sq.json"), "w"), indent=1)
        print(json.dumps(sq, indent=1))
    if "--bench" in sys.argv:        # per-launch table: bench.py's own launch rows joined with the trace and the counters
        launch_table(json.loads(open(sys.argv[sys.argv.index("--bench") + 1]).read().strip().splitlines()[-1]),
                     os.path.join(dst, f"{tag}_kernel_stats.csv"), out, sq, os.path.join(dst, f"{tag}_launches.json"))
    if "--current" in sys.argv:      # the file bench.py reads for roofline.traffic
        json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

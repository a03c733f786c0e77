#!/usr/bin/env python3
"""When every pair of ONE enqueue started and ended (diagnostic build with per-pair wall-clock stamps, `make timeline`):
    python3 tools/timeline.py [--pairs 8192] [--distinct 1024] [--yml config_4_level_optimization_analytic.yml]
Prints the enqueue's device time, the makespan seen by the stamps, when the last pair was drawn, how many workgroups are busy
over time (deciles of the makespan) and the pairs that end last with their start times -- i.e. what the end of a batch with
data-dependent termination looks like (profiles/r05_runs/).  The stamps sit in report slots of levels 14 and 15, which the
4- and 5-level files do not use."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: E402
_variant.use("timeline")
sys.path.insert(0, _variant.ROOT)
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=8192)
ap.add_argument("--distinct", type=int, default=1024)
ap.add_argument("--yml", default="config_4_level_optimization_analytic.yml")
ap.add_argument("--scene", default="plane")
args = ap.parse_args()

cfg = native.read_config_file(os.path.join(_variant.ROOT, "config_files", args.yml))
nl = cfg.num_levels
seq = synthetic.make_sequence(100, args.distinct + 1, 640, 480, holes=0.01, scene=args.scene, workers=min(16, os.cpu_count() or 1))
reps = (args.pairs + args.distinct - 1) // args.distinct
with odometry.AlignmentEngine() as eng:
    eng.set_batch_invariant(True)
    eng.set_config(cfg)
    eng.set_intrinsic_matrix(seq["K"])
    eng.reserve_frames(reps * (args.distinct + 1), 640, 480)
    src, tgt = [], []
    for r in range(reps):
        base = r * (args.distinct + 1)
        eng.upload_frames(base, seq["gray"], seq["depth"])
        src += [base + t for t in range(args.distinct)]
        tgt += [base + t + 1 for t in range(args.distinct)]
    src, tgt = np.array(src[:args.pairs], dtype=np.int32), np.array(tgt[:args.pairs], dtype=np.int32)
    for k in (0,):
        for _ in range(3):
            eng.align_pairs(src, tgt)
        ms = []
        for _ in range(5):
            eng.enqueue_align(src, tgt)
            eng.synchronize()
            ms.append(eng.last_align_ms()[0])
        _, rep = eng.fetch_results(len(src), want_reports=True)
        launches = eng.last_launches()
        TICK = 1e-2                                   # microseconds per tick of the 100 MHz wall clock
        begin = np.array([r.valid_pixels[15] for r in rep], dtype=np.int64)
        end = np.array([r.valid_pixels[14] for r in rep], dtype=np.int64)
        aside = back = np.zeros(len(rep), dtype=np.int64)
        its = np.array([list(r.iterations[:nl]) for r in rep])
        t0 = begin.min()
        b, e = (begin - t0) * TICK, (end - t0) * TICK                       # (a wrap of the 31-bit stamp inside one enqueue: every 21 s)
        span = e.max()
        was_aside = aside != 0
        print(f"\n== launches {[(l['kind'], l['levels'], l['workgroups']) for l in launches]}")
        print(f"device time of the enqueue (HIP events) {np.median(ms):.3f} ms [{min(ms):.3f} .. {max(ms):.3f}]; "
              f"first start to last end {span / 1e3:.3f} ms; last unseen pair drawn at {b.max() / 1e3:.3f} ms; "
              f"{int(was_aside.sum())} pairs set aside")
        mean_its = its.mean(axis=0)
        print("mean iterations per level", [round(float(v), 2) for v in mean_its], "max", its.max(axis=0).tolist())
        # busy workgroups over time: a pair occupies its workgroup from start to end, minus the time it waited on a list
        grid = np.linspace(0.0, span, 201)
        busy = np.zeros_like(grid)
        a_s, a_e = (aside - t0) * TICK, (back - t0) * TICK
        for i in range(len(b)):
            if was_aside[i]:
                busy += ((grid >= b[i]) & (grid < a_s[i])) | ((grid >= a_e[i]) & (grid < e[i]))
            else:
                busy += (grid >= b[i]) & (grid < e[i])
        dec = [int(busy[int(q * 200 / 10)]) for q in range(10)] + [int(busy[199])]
        print("busy workgroups at 0,10,...,90,99.5 % of the span:", dec)
        idle_tail = float(np.trapezoid(np.maximum(busy.max() - busy, 0), grid) / busy.max())
        print(f"workgroup-time idle inside the span: {idle_tail / 1e3:.3f} ms-equivalents of the full grid "
              f"({100 * idle_tail / span:.1f} % of the span)")
        order = np.argsort(-(e))[:8]
        for i in order:
            print(f"  pair {i:5d}: start {b[i] / 1e3:.3f} ms end {e[i] / 1e3:.3f} ms iterations {its[i].tolist()}"
                  + (f" set aside at {a_s[i] / 1e3:.3f}, continued at {a_e[i] / 1e3:.3f}" if was_aside[i] else ""))

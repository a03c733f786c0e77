"""Diagnostic (phase-stamp build only): when each pair of a level launch began and ended on the GPU's 100 MHz wall
clock, with the shipped thresholds -- how full the chip is over the launch, and what the last workgroups are doing.

    make -C photoconsistency-visual-odometry_amd/csrc stamps
    PHOVO_HIP_LIBRARY=photoconsistency-visual-odometry_amd/libphovo_hip_stamps.so python tools/queue_timeline.py [level=2]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2048          # [pairs] [fixed iterations per pair: one uncapped launch]
fixed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
distinct = 32
seq = synthetic.make_sequence(100, distinct + 1, 640, 480, holes=0.01)
cfg = native.read_config_file(os.path.join(os.path.dirname(__file__), "..", "config_files",
                                           "config_4_level_optimization_analytic.yml"))
if level == 3:                      # the stamps of the last level launched are the ones that stay in the report
    cfg.max_num_iterations[2] = 0
if fixed:
    for l in range(4):
        cfg.min_gradient_norm[l] = 0.0
    cfg.max_num_iterations[level] = fixed
with odometry.AlignmentEngine() as eng:
    eng.set_config(cfg)
    eng.set_intrinsic_matrix(seq["K"])
    reps = pairs // distinct
    eng.reserve_frames(reps * (distinct + 1), 640, 480)
    src, tgt = [], []
    for r in range(reps):
        eng.upload_frames(r * (distinct + 1), seq["gray"], seq["depth"])
        src += [r * (distinct + 1) + t for t in range(distinct)]
        tgt += [r * (distinct + 1) + t + 1 for t in range(distinct)]
    eng.align_pairs(src, tgt)
    _, rp = eng.align_pairs(src, tgt, want_reports=True)
    ms = eng.last_align_ms()[1][level]
it = np.array([r.iterations[level] for r in rp])
b = np.array([r.iterations[13] for r in rp], dtype=np.int64) & 0xFFFFFFFF
e = np.array([r.iterations[14] for r in rp], dtype=np.int64) & 0xFFFFFFFF
wg = np.array([r.iterations[15] for r in rp])
t0 = b.min()
b, e = (b - t0) / 100.0, (e - t0) / 100.0          # microseconds
print(f"level {level}: launch {ms * 1e3:.0f} us by HIP events; first begin 0, last end {e.max():.0f} us; "
      f"workgroups {wg.max() + 1}; iterations mean {it.mean():.2f} max {it.max()}")
dur = e - b
cyc = np.array([[r.iterations[8 + j] for j in range(5)] for r in rp], dtype=np.float64).sum(axis=1)      # wave 0's cycles per iteration
print(f"   wave 0 counts {cyc.mean():.0f} cycles per iteration over {np.mean(dur / np.maximum(it, 1)):.2f} us of wall clock: "
      f"{cyc.mean() / np.mean(dur / np.maximum(it, 1)) / 1e3:.3f} GHz effective")
for k in sorted(set(it)):
    m = it == k
    print(f"   {k:3d} iterations: {m.sum():5d} pairs, {dur[m].mean():7.1f} us per pair = {dur[m].mean() / k:6.1f} us per iteration; "
          f"begin {b[m].min():6.0f}..{b[m].max():6.0f} us")
edges = np.linspace(0, e.max(), 21)
for lo, hi in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (lo + hi)
    active = int(((b <= mid) & (e > mid)).sum())
    print(f"   t = {mid:7.0f} us: {active:4d} pairs in flight")
gap = []
for w in range(wg.max() + 1):
    m = np.where(wg == w)[0]
    o = m[np.argsort(b[m])]
    gap += list(b[o][1:] - e[o][:-1])
print(f"   between a workgroup's pairs (write-back, draw, prologue): mean {np.mean(gap):.1f} us, max {np.max(gap):.1f} us, n {len(gap)}")
# where that time goes (thread 0's stamps, valid_pixels[8..12] of the stamp build): end of the pair before -> its ticket
# drawn (write-back + atomic) -> loop-top barrier passed -> state loaded -> pose constants written -> prologue barrier
vp = np.array([[r.valid_pixels[8 + j] for j in range(5)] for r in rp], dtype=np.int64) & 0xFFFFFFFF
b_raw = np.array([r.iterations[13] for r in rp], dtype=np.int64) & 0xFFFFFFFF
m = vp[:, 0] != 0                                  # pairs that had a predecessor on their workgroup
if m.any():
    pts = np.concatenate([vp[m], b_raw[m, None]], axis=1).astype(np.float64) / 100.0
    d = np.diff(pts, axis=1)
    keep = (pts[:, -1] - pts[:, 0]) < 200.0        # (a workgroup that waited for the next launch is not a gap)
    names = ["write-back + draw", "to loop-top barrier", "frame indices + state load", "sincos + pose constants", "prologue barrier"]
    print(f"   gap breakdown over {int(keep.sum())} pairs (us): " +
          ", ".join(f"{nm} {d[keep, j].mean():.2f}" for j, nm in enumerate(names)) +
          f"; total {(pts[keep, -1] - pts[keep, 0]).mean():.2f}")

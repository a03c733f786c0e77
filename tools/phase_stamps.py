#!/usr/bin/env python3
"""Where one pair's iterations spend their time inside the level kernels (diagnostic build of the library):
    python3 tools/phase_stamps.py [pairs [shipped]]        (builds csrc/build_stamps/libphovo_hip_stamps.so: `make stamps`)
Workgroup 0 prints, for every pair it draws and four of its waves, the 10 ns ticks spent in pass 1, at the barrier behind it,
in pass 2, in the butterfly, at the barrier in front of the solve and in / waiting for the solve (csrc/gn_kernels.hip,
PHOVO_STAMP).  Fixed-iteration mode of the 4-level configuration: 50 iterations at 80x60, 20 at 160x120.  (The printing
workgroup is slowed by its own printf round trips and draws fewer pairs than the others; with the shipped thresholds its
later pairs run at the tail of the batch, next to an idle neighbour -- read the fixed-iteration figures, where every pair of
a workgroup sees the same load.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: E402
_variant.use("stamps")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import phovo_amd  # noqa: F401,E402
from phovo_amd import native, odometry, synthetic  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cfg = native.read_config_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "config_files",
                                          "config_4_level_optimization_analytic.yml"))
shipped = len(sys.argv) > 2 and sys.argv[2] == "shipped"      # keep the yml's gradient thresholds: the fused launch
if not shipped:
    for l in range(cfg.num_levels):
        cfg.min_gradient_norm[l] = 0.0
pairs = [synthetic.make_pair(i, 640, 480) for i in range(8)]
with odometry.AlignmentEngine() as eng:
    eng.set_config(cfg)
    eng.set_intrinsic_matrix(pairs[0]["K"])
    eng.reserve_frames(16, 640, 480)
    for i, p in enumerate(pairs):
        eng.upload_frame(2 * i, p["gray0"], p["depth0"], roles=native.ROLE_SOURCE)
        eng.upload_frame(2 * i + 1, p["gray1"], None, roles=native.ROLE_TARGET)
    src = [2 * (k % 8) for k in range(n_pairs)]
    tgt = [2 * (k % 8) + 1 for k in range(n_pairs)]
    eng.align_pairs(src, tgt)
    print("launches:", [(r["kind"], r["levels"], r["threads"]) for r in eng.last_launches()], flush=True)

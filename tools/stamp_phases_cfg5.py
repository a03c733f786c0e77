"""Diagnostic (phase-stamp build): per-phase cycles of the 1280x960 6-level configuration's levels, one level at a
time (level 2 = 320x240 is the one whose owner map lives in HBM).

    PHOVO_HIP_LIBRARY=photoconsistency-visual-odometry_amd/libphovo_hip_stamps.so python tools/stamp_phases_cfg5.py [pairs=1024]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seq = synthetic.make_sequence(100, 9, 1280, 960, holes=0.01)
cfg = native.read_config_file(os.path.join(os.path.dirname(__file__), "..", "config_files",
                                           "config_6_level_optimization_analytic.yml"))
nl = cfg.num_levels
full = list(cfg.max_num_iterations[:nl])
names = ["pass1", "barrier1", "pass2", "reduce+barrier", "serial+barrier"]
for lvl in range(nl):
    if full[lvl] <= 0:
        continue
    for l in range(nl):
        cfg.max_num_iterations[l] = full[l] if l == lvl else 0
        cfg.min_gradient_norm[l] = 0.0
    with odometry.AlignmentEngine() as eng:
        eng.set_config(cfg)
        eng.set_intrinsic_matrix(seq["K"])
        reps = pairs // 8
        eng.reserve_frames(reps * 9, 1280, 960)
        src, tgt = [], []
        for r in range(reps):
            eng.upload_frames(r * 9, seq["gray"], seq["depth"])
            src += [r * 9 + t for t in range(8)]
            tgt += [r * 9 + t + 1 for t in range(8)]
        eng.align_pairs(src, tgt)
        _, rp = eng.align_pairs(src, tgt, want_reports=True)
        ms = eng.last_align_ms()[1][lvl]
        info = eng.level_launch_info(lvl)
        w, h = eng.level_size(lvl)
    st = np.array([[r.iterations[8 + j] for j in range(5)] for r in rp], dtype=np.float64)
    tot = st.sum(axis=1).mean()
    print(f"level {lvl} ({w}x{h}, {full[lvl]} iterations, {info['threads']} threads, owner in LDS {info['owner_in_lds']}): "
          f"{ms:.3f} ms for {len(src)} pairs; cycles/iteration {tot:.0f}")
    for j, nme in enumerate(names):
        print(f"   {nme:16s} {st[:, j].mean():9.0f} cycles  {100 * st[:, j].mean() / max(tot, 1):5.1f} %")

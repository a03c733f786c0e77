"""Diagnostic: per-phase cycles of one GN iteration (wave 0's view), from the phase-stamp build.

    make -C photoconsistency-visual-odometry_amd/csrc stamps
    PHOVO_HIP_LIBRARY=photoconsistency-visual-odometry_amd/libphovo_hip_stamps.so python tools/stamp_phases.py

Shares are what matters; the stamp build's run time is not quoted anywhere.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seq = synthetic.make_sequence(100, 9, 640, 480, holes=0.01)
cfg = native.read_config_file(os.path.join(os.path.dirname(__file__), "..", "config_files",
                                           "config_4_level_optimization_analytic.yml"))
for l in range(4):
    cfg.min_gradient_norm[l] = 0.0
names = ["pass1", "barrier1", "pass2", "reduce+barrier", "serial+barrier"]
for levels in ([0, 0, 20, 0], [0, 0, 0, 50]):
    for l in range(4):
        cfg.max_num_iterations[l] = levels[l]
    with odometry.AlignmentEngine() as eng:
        eng.set_config(cfg)
        eng.set_intrinsic_matrix(seq["K"])
        reps = pairs // 8
        eng.reserve_frames(reps * 9, 640, 480)
        src, tgt = [], []
        for r in range(reps):
            for f in range(9):
                eng.upload_frame(r * 9 + f, seq["gray"][f], seq["depth"][f])
            src += [r * 9 + t for t in range(8)]
            tgt += [r * 9 + t + 1 for t in range(8)]
        eng.align_pairs(src, tgt)
        _, reps_ = eng.align_pairs(src, tgt, want_reports=True)
        ms, per = eng.last_align_ms()
        st = np.array([[r.iterations[8 + j] for j in range(5)] for r in reps_], dtype=np.float64)
        tot = st.sum(axis=1).mean()
        print(f"levels {levels}: kernel {ms:.3f} ms for {len(src)} pairs; cycles/iteration {tot:.0f}")
        for j, nme in enumerate(names):
            print(f"   {nme:16s} {st[:, j].mean():9.0f} cycles  {100 * st[:, j].mean() / tot:5.1f} %")
        sub = np.array([[r.iterations[5 + j] for j in range(3)] for r in reps_], dtype=np.float64).mean(axis=0)
        print(f"   serial section of wave 0: sum+broadcast {sub[0]:.0f}, solve+update {sub[1]:.0f}, pose constants {sub[2]:.0f} cycles")

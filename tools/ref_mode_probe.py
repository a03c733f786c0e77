"""Per-level kernel time with the SHIPPED thresholds (data-dependent iteration counts), 2048 pairs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402

pairs, distinct = 2048, 32
seq = synthetic.make_sequence(100, distinct + 1, 640, 480, holes=0.01)
cfg = native.read_config_file(os.path.join(os.path.dirname(__file__), "..", "config_files",
                                           "config_4_level_optimization_analytic.yml"))
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
    for _ in range(3):
        s, reps_ = eng.align_pairs(src, tgt, want_reports=True)
        tot, per = eng.last_align_ms()
        it = np.array([list(r.iterations[:4]) for r in reps_])
        print(f"total {tot:.3f} ms  level2 {per[2]:.3f} ms  level3 {per[3]:.3f} ms   mean it {it.mean(axis=0)}  "
              f"px-it per pair {(it[:, 2] * 19200 + it[:, 3] * 4800).mean():.0f}")

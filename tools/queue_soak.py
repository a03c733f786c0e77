"""Soak test of the work queue's pair hand-over: N launches of 2500 shuffled copies of three problems (5-level
configuration, 5 fixed iterations per active level); every copy of a problem must come out bit-identical in every launch.
    python tools/queue_soak.py [launches=500]
Before the explicit LDS wait in front of the loop-head barrier about one launch in ten failed."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 500
ncfg = native.read_config_file(os.path.join(os.path.dirname(__file__), "..", "config_files",
                                            "config_5_level_optimization_analytic.yml"))
nl = ncfg.num_levels
ncfg = native.make_config(num_levels=nl, max_iter=[min(m, 5) for m in ncfg.max_num_iterations[:nl]], min_grad=[0.0] * nl)
probs = [synthetic.make_pair(21, 640, 480, holes=0.02, trans=0.004, rot=0.002),
         synthetic.make_pair(22, 640, 480, holes=0.0, trans=0.03, rot=0.015),
         synthetic.make_pair(23, 640, 480, holes=0.05, trans=0.06, rot=0.03)]
order = np.random.RandomState(5).randint(0, 3, size=2500)
bad = 0
with odometry.AlignmentEngine() as eng:
    eng.set_config(ncfg)
    eng.set_intrinsic_matrix(probs[0]["K"])
    eng.reserve_frames(6, 640, 480)
    for i, p in enumerate(probs):
        eng.upload_frame(2 * i, p["gray0"], p["depth0"], roles=native.ROLE_SOURCE)
        eng.upload_frame(2 * i + 1, p["gray1"], None, roles=native.ROLE_TARGET)
    src, tgt = [2 * int(i) for i in order], [2 * int(i) + 1 for i in order]
    ref = eng.align_pairs(src, tgt)
    for i in range(3):
        idx = np.where(order == i)[0]
        assert all(np.array_equal(ref[idx[0]], ref[k]) for k in idx), "first launch already inconsistent"
    for t in range(launches):
        s = eng.align_pairs(src, tgt)
        if not np.array_equal(s, ref):
            bad += 1
            print(f"launch {t}: {int((s != ref).any(axis=1).sum())} pairs differ")
print(f"{launches} launches of 2500 pairs, {bad} with deviations")
sys.exit(1 if bad else 0)

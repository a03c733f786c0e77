"""Per-level device time for small batches (1..256 pairs) under the launch geometry the environment selects:
    python tools/latency_probe.py            default plan (wide form for <= 32 pairs on >= 16384 px)
    PHOVO_PROBE_WIDE_POLICY=-1 ...           never the wide form (persistent kernels only)
    PHOVO_GN_NO_QUAD=1 / PHOVO_GN_FORCE_WIDE=1   tuning aids of gn_plan_level (512 / 1024-thread workgroups)
Fixed iterations, 640x480 pyramids: level 2 = 160x120 x 20 iterations, level 3 = 80x60 x 50 iterations."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402

policy = int(os.environ.get("PHOVO_PROBE_WIDE_POLICY", "0"))
seq = synthetic.make_sequence(7, 9, 640, 480, holes=0.01)
cfg = native.make_config(num_levels=4, max_iter=[0, 0, 20, 50], min_grad=[0.0] * 4)
if os.environ.get("PHOVO_PROBE_SHIPPED"):      # the yml thresholds: a handful of iterations per level
    cfg = native.read_config_file(os.path.join(os.path.dirname(__file__), "..", "config_files",
                                               "config_4_level_optimization_analytic.yml"))
with odometry.AlignmentEngine() as eng:
    eng.set_config(cfg)
    eng.set_intrinsic_matrix(seq["K"])
    eng.set_wide_policy(policy)
    eng.reserve_frames(9, 640, 480)
    eng.upload_frames(0, seq["gray"], seq["depth"])
    info = {l: eng.level_launch_info(l) for l in (2, 3)}
    print("plan", {l: (v["threads"], v["lds_bytes"]) for l, v in info.items()}, "wide policy", policy)
    for n in (1, 8, 32, 64, 256):
        src = [i % 8 for i in range(n)]
        tgt = [i % 8 + 1 for i in range(n)]
        best = None
        for _ in range(4):
            eng.align_pairs(src, tgt)
            tot, per = eng.last_align_ms()
            cur = (tot, per[2], per[3])
            best = cur if best is None or cur[0] < best[0] else best
        _, reps = eng.align_pairs(src, tgt, want_reports=True)
        print("   iterations of pair 0:", list(reps[0].iterations[:4]))
        print(f"pairs {n:4d}: total {best[0]:7.3f} ms  level2 {best[1]:7.3f} ms ({best[1] / 20 * 1e3:6.1f} us/it)  "
              f"level3 {best[2]:7.3f} ms ({best[2] / 50 * 1e3:6.1f} us/it)")

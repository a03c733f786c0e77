"""Where the host time of one small Optimize() goes: wall time of enqueue / synchronize / fetch for a batch of 1, 8 and 64
pairs (4-level configuration, shipped thresholds), next to the device time of the enqueue.
    python tools/host_overhead_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402

cfg = native.read_config_file(os.path.join(os.path.dirname(__file__), "..", "config_files",
                                           "config_4_level_optimization_analytic.yml"))
seq = synthetic.make_sequence(100, 9, 640, 480, holes=0.01)
with odometry.AlignmentEngine() as eng:
    eng.set_config(cfg)
    eng.set_intrinsic_matrix(seq["K"])
    eng.reserve_frames(9, 640, 480)
    eng.upload_frames(0, seq["gray"], seq["depth"])
    for n in (1, 8, 64):
        src = [i % 8 for i in range(n)]
        tgt = [s + 1 for s in src]
        for _ in range(20):
            eng.align_pairs(src, tgt)
        t = np.zeros(3)
        dev = 0.0
        reps = 300
        for _ in range(reps):
            t0 = time.perf_counter()
            eng.enqueue_align(src, tgt)
            t1 = time.perf_counter()
            eng.synchronize()
            t2 = time.perf_counter()
            eng.fetch_results(n)
            t3 = time.perf_counter()
            t += (t1 - t0, t2 - t1, t3 - t2)
            dev += eng.last_align_ms()[0]
        t *= 1e6 / reps
        print(f"{n:3d} pairs: enqueue {t[0]:6.1f} us, synchronize {t[1]:6.1f} us, fetch {t[2]:6.1f} us, total {t.sum():6.1f} us; "
              f"device {1e3 * dev / reps:6.1f} us")

"""PCIe-inclusive rates: host buffers in, device pyramids built, poses out.
Never bench.py's `value` -- that one starts with the pyramids resident in HBM."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 257
seq = synthetic.make_sequence(3, 17, 640, 480, holes=0.01)
gray = [seq["gray"][f % 17] for f in range(F)]
depth = [seq["depth"][f % 17] for f in range(F)]
d16 = [np.rint(d * 5000.0).astype(np.uint16) for d in depth]
cfg = native.read_config_file(os.path.join(os.path.dirname(__file__), "..", "config_files",
                                           "config_4_level_optimization_analytic.yml"))
with odometry.AlignmentEngine() as eng:
    eng.set_config(cfg)
    eng.set_intrinsic_matrix(seq["K"])
    eng.reserve_frames(F, 640, 480)
    for f in range(8):
        eng.upload_frame(f, gray[f], depth[f])
    t0 = time.perf_counter()
    for f in range(F):
        eng.upload_frame(f, gray[f], depth[f])
    t_f64 = time.perf_counter() - t0
    t0 = time.perf_counter()
    for f in range(F):
        eng.upload_frame_u16(f, gray[f], d16[f], 1.0 / 5000.0)
    t_u16 = time.perf_counter() - t0
    G, D16, D64 = np.stack(gray), np.stack(d16), np.stack(depth)
    t0 = time.perf_counter()
    eng.upload_frames(0, G, D64)
    t_bf64 = time.perf_counter() - t0
    t0 = time.perf_counter()
    eng.upload_frames(0, G, D16, depth_scale=1.0 / 5000.0)
    t_bu16 = time.perf_counter() - t0
    src, tgt = list(range(F - 1)), list(range(1, F))
    eng.align_pairs(src, tgt)
    t0 = time.perf_counter()
    eng.upload_frames(0, G, D16, depth_scale=1.0 / 5000.0)
    eng.align_pairs(src, tgt)
    t_e2e = time.perf_counter() - t0
print(f"upload+pyramids fp64 depth : {F / t_f64:9.0f} frames/s  ({2.76 * F / t_f64 / 1e3:.2f} GB/s over PCIe)")
print(f"upload+pyramids u16 depth  : {F / t_u16:9.0f} frames/s  ({0.92 * F / t_u16 / 1e3:.2f} GB/s over PCIe)")
print(f"batched upload fp64 depth   : {F / t_bf64:9.0f} frames/s  ({2.76 * F / t_bf64 / 1e3:.2f} GB/s over PCIe)")
print(f"batched upload u16 depth    : {F / t_bu16:9.0f} frames/s  ({0.92 * F / t_bu16 / 1e3:.2f} GB/s over PCIe)")
print(f"end-to-end sequence (batched u16 upload + pyramids + Optimize, shipped thresholds): {(F - 1) / t_e2e:9.0f} alignments/s")

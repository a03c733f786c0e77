"""Soak test of the double-buffered batched upload (two staging halves, copy stream beside the engine's stream): F frames
uploaded in one batched call, `rounds` times with fresh content, every plane of every level compared bit for bit with the
frame-by-frame upload of the same data into a second engine.

    python tools/upload_soak.py [frames=200] [rounds=10]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w, h, nl = 320, 240, 3
cfg = native.make_config(num_levels=nl, max_iter=[1] * nl, min_grad=[0.0] * nl, blur=[3, 0, 5])
bad = 0
with odometry.AlignmentEngine() as a, odometry.AlignmentEngine() as b:
    for e in (a, b):
        e.set_config(cfg)
        e.reserve_frames(F, w, h)
    for r in range(rounds):
        rs = np.random.RandomState(r)
        gray = rs.randint(0, 256, size=(F, h, w)).astype(np.uint8)
        d16 = rs.randint(0, 30000, size=(F, h, w)).astype(np.uint16)
        a.upload_frames(0, gray, d16, depth_scale=1.0 / 5000.0)
        for f in range(F):
            b.upload_frame_u16(f, gray[f], d16[f], 1.0 / 5000.0)
        for f in range(F):
            for l in range(nl):
                for x, y in zip(a.get_level_planes(f, l), b.get_level_planes(f, l)):
                    if not np.array_equal(x, y):
                        bad += 1
        print(f"round {r}: {F} frames compared, {bad} differing planes so far", flush=True)
print(f"{rounds} rounds x {F} frames, {bad} differing planes")
sys.exit(1 if bad else 0)

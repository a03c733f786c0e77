"""Probe for the wide level form: one 640x480 pair, level 0 only, 20 fixed iterations (run under rocprofv3)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402

p = synthetic.make_pair(1, 640, 480, holes=0.01, trans=0.01, rot=0.005)
cfg = native.make_config(num_levels=1, max_iter=[20], min_grad=[0.0])
with odometry.AlignmentEngine() as eng:
    eng.set_config(cfg)
    eng.set_intrinsic_matrix(p["K"])
    eng.reserve_frames(2, 640, 480)
    eng.upload_frame(0, p["gray0"], p["depth0"])
    eng.upload_frame(1, p["gray1"], p["depth1"])
    for _ in range(5):
        eng.align_pairs([0], [1])
        print("device ms", eng.last_align_ms()[0])

"""Calibration workload for the FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md, HBM section):
uploads frames with every pyramid level built, so that k_resize_level<double> at level 0 runs as a pure
8-byte-per-lane coalesced copy of a known size (640*480*8 = 2,457,600 B read and written per dispatch) --
the same access width the Gauss-Newton kernel uses for its plane loads."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402

seq = synthetic.make_sequence(5, 4, 640, 480)
cfg = native.make_config(num_levels=2, max_iter=[1, 1])
with odometry.AlignmentEngine() as eng:
    eng.set_config(cfg)
    eng.set_build_all_levels(True)
    eng.reserve_frames(64, 640, 480)
    for f in range(64):
        eng.upload_frame(f, seq["gray"][f % 4], seq["depth"][f % 4])
print("calibration uploads done")

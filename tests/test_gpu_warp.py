"""phovo_warp_image (device) against the oracle's restatement of phovo::warpImage
(phovo/include/CPhotoconsistencyOdometry.h:73-134).  u8 output: bit-exact."""
import ctypes as C

import numpy as np
import pytest

import phovo_amd  # noqa: F401
from phovo_amd import native, odometry, se3, synthetic
from oracle import oracle

pytestmark = pytest.mark.gpu

K640 = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1.0]])


@pytest.mark.parametrize("state", [
    (0, 0, 0, 0, 0, 0),                                   # identity: every valid pixel maps to itself
    (0.02, -0.01, 0.015, 0.01, -0.008, 0.006),            # a typical inter-frame motion
    (0.3, 0.2, 0.8, 0.2, -0.15, 0.4),                     # large: many collisions (last raster writer wins) and holes
    (-0.1, 0.05, -1.2, 0.0, 0.0, 3.0),                    # camera pushed behind part of the scene: Z <= 0, wild coordinates
])
@pytest.mark.parametrize("level", [0, 2])
def test_warp_image_is_bit_exact(state, level):
    p = synthetic.make_pair(11, 640, 480, holes=0.03)
    d = p["depth0"].copy()
    d[5, 7] = -1.0                                         # negative depth fails the > 0 gate
    d[9, 9] = np.nan
    rt = se3.eigen_pose(np.array(state, dtype=np.float64))
    got = odometry.warpImage(p["gray0"], d, rt, K640, level=level)
    exp = oracle.warp_image(p["gray0"], d, rt, K640, level=level)
    assert got.dtype == np.uint8 and got.shape == exp.shape
    assert np.array_equal(got, exp)
    if state == (0, 0, 0, 0, 0, 0) and level == 0:
        valid = d > 0
        # not a tautology: the truncating cast moves pixels whose reprojection rounds to just under the integer
        assert np.mean(got[valid] == p["gray0"][valid]) > 0.5


def test_warp_image_small_odd_size_and_strides():
    rng = np.random.RandomState(3)
    h, w = 37, 53
    g = rng.randint(0, 256, size=(h, w + 11)).astype(np.uint8)        # padded rows: strides longer than the row
    d = rng.uniform(0.5, 4.0, size=(h, w + 5))
    d[rng.rand(h, w + 5) < 0.1] = 0.0
    K = np.array([[60.0, 0, 26.0], [0, 61.0, 18.0], [0, 0, 1.0]])
    rt = se3.eigen_pose(np.array([0.05, -0.02, 0.1, 0.05, 0.02, -0.04]))
    out = np.full((h, w + 3), 77, dtype=np.uint8)
    dp = C.POINTER(C.c_double)
    rtf, kf = np.ascontiguousarray(rt).reshape(16), np.ascontiguousarray(K).reshape(9)
    native.check(native.lib().phovo_warp_image(0, g.ctypes.data, g.strides[0], d.ctypes.data, d.strides[0], w, h,
                                               rtf.ctypes.data_as(dp), kf.ctypes.data_as(dp), 0,
                                               out.ctypes.data, out.strides[0]), "phovo_warp_image")
    exp = oracle.warp_image(np.ascontiguousarray(g[:, :w]), np.ascontiguousarray(d[:, :w]), rt, K)
    assert np.array_equal(out[:, :w], exp)
    assert np.all(out[:, w:] == 77)                                   # padding untouched


def test_warp_image_rejects_bad_arguments():
    L = native.lib()
    g = np.zeros((4, 4), dtype=np.uint8)
    d = np.ones((4, 4))
    dp = C.POINTER(C.c_double)
    rt, k = np.eye(4).reshape(16), np.eye(3).reshape(9)
    out = np.zeros((4, 4), dtype=np.uint8)
    args = lambda w, h, gs: (0, g.ctypes.data, gs, d.ctypes.data, 32, w, h, rt.ctypes.data_as(dp),   # noqa: E731
                             k.ctypes.data_as(dp), 0, out.ctypes.data, 4)
    assert L.phovo_warp_image(*args(0, 4, 4)) == 3           # PHOVO_E_SHAPE
    assert L.phovo_warp_image(*args(4, 4, 2)) == 3           # stride shorter than a row
    assert L.phovo_warp_image(0, None, 4, d.ctypes.data, 32, 4, 4, rt.ctypes.data_as(dp), k.ctypes.data_as(dp), 0,
                              out.ctypes.data, 4) == 1       # PHOVO_E_INVALID_ARGUMENT

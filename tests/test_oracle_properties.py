"""Size-independent properties and edge cases of the oracle."""
import numpy as np

import phovo_amd  # noqa: F401
from phovo_amd import se3, synthetic
from oracle import oracle, numpy_twin as twin


def _cfg(**kw):
    base = dict(num_levels=3, max_iter=[2, 3, 5], min_grad=[0, 0, 0], lam=[1, 1, 1],
                grad_scale=[0.0625] * 3, blur=[0, 0, 0])
    base.update(kw)
    return oracle.make_config(**base)


def test_defaults_match_reference_constructor():
    # ...Analytic.h:430-443
    cfg = oracle.make_config()
    assert cfg.num_levels == 5
    assert list(cfg.max_num_iterations[:5]) == [0, 0, 5, 20, 50]
    assert list(cfg.min_gradient_norm[:5]) == [300.0] * 5
    assert list(cfg.image_gradients_scaling_factor[:5]) == [0.0625] * 5
    assert (cfg.min_depth, cfg.max_depth) == (0.3, 5.0)


def test_identical_frames_keep_zero_state():
    p = synthetic.make_pair(3, 64, 48)
    state, iters = oracle.align_frames(_cfg(), p["K"], p["gray0"], p["depth0"], p["gray0"])
    assert np.all(state == 0.0)
    assert iters == [2, 3, 5]


def test_levels_with_zero_iterations_leave_state_untouched():
    p = synthetic.make_pair(4, 64, 48)
    init = np.array([0.01, -0.02, 0.005, 0.003, -0.002, 0.001])
    state, iters = oracle.align_frames(_cfg(max_iter=[0, 0, 0]), p["K"], p["gray0"], p["depth0"],
                                       p["gray1"], init_state=init)
    assert np.array_equal(state, init)
    assert iters == [1, 1, 1]            # the loop body runs once per level (:510,547-549)


def test_scatter_last_writer_wins_and_jacobian_at_source_index():
    # 4x1 image, all four source pixels are sent to column 2: the LAST one (c=3) must own r[2].
    w, h = 4, 1
    K = np.array([[1.0, 0, 0.0], [0, 1.0, 0.0], [0, 0, 1]])
    i0 = np.array([[0.1, 0.2, 0.3, 0.4]])
    i1 = np.array([[0.5, 0.6, 0.7, 0.8]])
    # depth z, state x: tc = (c*z + x)/z ; choose x so that every pixel lands near 2
    d0 = np.array([[1.0, 1.0, 1.0, 1.0]])
    gx = np.array([[1.0, 2.0, 3.0, 4.0]])
    gy = np.zeros((1, 4))
    # collapse all columns: use a large depth-dependent trick instead -> state x = 0 keeps identity,
    # so build the collision with depth: px = c*z, X = px + x, tc = X/z.  z=1 except pixel 0 (z=0.5, x=1 -> tc=2)
    d0 = np.array([[0.5, 1.0, 1.0, 1.0]])
    state = np.array([1.0, 0, 0, 0, 0, 0])
    r, J = oracle.compute_residuals_and_jacobians(i0, d0, i1, gx, gy, 0, K, state, 0.3, 5.0)
    # pixel 0: tc = (0*0.5+1)/0.5 = 2 ; pixel 1: tc = 2 ; pixel 2: tc = 3 ; pixel 3: tc = 4 (out)
    assert r[2] == i1[0, 2] - i0[0, 1]          # pixel 1 (later) wins over pixel 0
    assert r[3] == i1[0, 3] - i0[0, 2]
    assert r[0] == 0 and r[1] == 0
    # Jacobian rows live at the SOURCE index; pixel 3 is out of bounds -> zero row
    assert np.all(J[:, 3] == 0)
    assert J[0, 0] == gx[0, 0] * (1.0 / 0.5) and J[0, 1] == gx[0, 1] * 1.0
    g, Hm, r2, J2 = twin.normal_equations((i0, d0, i1, gx, gy), 0, K, state)
    np.testing.assert_allclose(r, r2, atol=0)
    np.testing.assert_allclose(J.T, J2, rtol=1e-15)


def test_depth_gate_is_strict():
    # minD < z < maxD, strict on both sides (:280)
    K = np.eye(3)
    i0 = np.full((1, 3), 0.5)
    i1 = np.full((1, 3), 0.25)
    d0 = np.array([[0.3, 1.0, 5.0]])
    ones = np.ones((1, 3))
    r, J = oracle.compute_residuals_and_jacobians(i0, d0, i1, ones, ones, 0, K, np.zeros(6), 0.3, 5.0)
    assert r[0] == 0 and r[2] == 0 and r[1] == -0.25
    assert np.all(J[:, 0] == 0) and np.all(J[:, 2] == 0) and np.any(J[:, 1] != 0)


def test_c_round_half_away_from_zero():
    # tc = c + x; x = 0.5 sends pixel c to c+1 (round(0.5)=1, round(1.5)=2), not banker's rounding
    K = np.eye(3)
    i0 = np.array([[0.1, 0.2, 0.3, 0.4]])
    i1 = np.array([[0.5, 0.6, 0.7, 0.9]])
    d0 = np.ones((1, 4))
    z = np.zeros((1, 4))
    r, _ = oracle.compute_residuals_and_jacobians(i0, d0, i1, z, z, 0, K,
                                                  np.array([0.5, 0, 0, 0, 0, 0]), 0.3, 5.0)
    np.testing.assert_allclose(r, [0, 0.6 - 0.1, 0.7 - 0.2, 0.9 - 0.3], atol=1e-16)


def test_transcription_bug_is_reproduced():
    # With x != 0 the (u,z) entry uses px*(cp*cy + x) instead of px*cp*cy + x  (:253,:325)
    K = np.array([[100.0, 0, 2.0], [0, 100.0, 2.0], [0, 0, 1]])
    n = 5
    i0 = np.zeros((n, n)); i1 = np.zeros((n, n))
    d0 = np.full((n, n), 2.0)
    gx = np.ones((n, n)); gy = np.zeros((n, n))
    state = np.array([0.01, 0.0, 0.0, 0.0, 0.0, 0.0])
    _, J = oracle.compute_residuals_and_jacobians(i0, d0, i1, gx, gy, 0, K, state, 0.3, 5.0)
    c = 3; r_ = 2; i = r_ * n + c
    px = (c - 2.0) * 2.0 / 100.0
    Z = 2.0
    buggy = -100.0 * (2.0 * 0.0 + 0.0 + px * (1.0 + 0.01)) / Z ** 2
    true = -100.0 * (px * 1.0 + 0.01) / Z ** 2
    assert abs(J[2, i] - buggy) < 1e-15
    assert abs(J[2, i] - true) > 1e-6


def test_oracle_and_twin_agree_on_random_problem():
    p = synthetic.make_pair(11, 80, 60, holes=0.03)
    cfg = _cfg()
    s1, it1 = oracle.align_frames(cfg, p["K"], p["gray0"], p["depth0"], p["gray1"])
    pyr = twin.build_pyramids(p["gray0"], p["depth0"], p["gray1"], 3, [0.0625] * 3)
    # 80x60 is divisible by 4
    s2, it2, _ = twin.optimize(pyr, p["K"], dict(num_levels=3, lam=[1, 1, 1], max_iter=[2, 3, 5],
                                                  min_grad=[0, 0, 0]))
    assert it1 == it2
    assert se3.state_distance(s1, s2) < 1e-9


def test_warp_image_truncates_and_gates_on_positive_depth():
    # CPhotoconsistencyOdometry.h:107,119-122
    K = np.eye(3)
    g = np.array([[10, 20, 30, 40]], dtype=np.uint8)
    d = np.array([[1.0, 0.0, 1.0, 1.0]])
    rt = np.eye(4); rt[0, 3] = 0.9            # tc = c + 0.9 -> truncated to c
    out = oracle.warp_image(g, d, rt, K)
    assert list(out[0]) == [10, 0, 30, 40]
    rt[0, 3] = 1.0
    out = oracle.warp_image(g, d, rt, K)
    assert list(out[0]) == [0, 10, 0, 30]


def test_level_zero_blur_is_in_place_and_feeds_the_later_levels():
    """BuildPyramid: `imgAux = img` (...Analytic.h:136) is a shallow cv::Mat alias, so GaussianBlur(imgAux, imgAux)
    (:146-147) blurs the converted image itself and every later cv::resize(img, ...) (:132) reads the blurred
    level 0.  Oracle against the independent twin (scipy separable correlation), and against the naive reading
    (resize from the unblurred image), which must differ.  Parity unpinned: OpenCV is not in this image and every
    shipped configuration has blurFilterSize = 0."""
    p = synthetic.make_pair(12, 160, 120)
    blur = [5, 3, 0]
    cfg = oracle.make_config(num_levels=3, blur=blur, max_iter=[1, 1, 1], min_grad=[0, 0, 0])
    i1p, gxp, gyp = oracle.build_target_pyramids(p["gray1"], cfg)
    i0p, d0p = oracle.build_source_pyramids(p["gray0"], p["depth0"], cfg)
    tp = twin.intensity_pyramid(p["gray1"], 3, blur)
    for l in range(3):
        np.testing.assert_allclose(i1p[l], tp[l], rtol=0, atol=1e-14)
        gx, gy = twin.scharr(tp[l], 0.0625)
        np.testing.assert_allclose(gxp[l], gx, rtol=0, atol=1e-13)
    raw = oracle.convert_intensity(p["gray1"])
    # level 0: blurred once (twice the filter, not four times); level 1: resize OF the blurred level 0, then its own blur
    np.testing.assert_array_equal(i1p[0], oracle.gaussian_blur_twice(raw, 5))
    np.testing.assert_array_equal(i1p[1], oracle.gaussian_blur_twice(oracle.resize_level(i1p[0], 1), 3))
    np.testing.assert_array_equal(i1p[2], oracle.resize_level(i1p[0], 2))
    naive1 = oracle.gaussian_blur_twice(oracle.resize_level(raw, 1), 3)
    assert np.max(np.abs(naive1 - i1p[1])) > 1e-4
    # depth is built with applyBlur = false (:475): never blurred
    np.testing.assert_array_equal(d0p[1], oracle.resize_level(p["depth0"], 1))
    # blur only at a later level: level 0 untouched, later levels resized from the raw image
    cfg2 = oracle.make_config(num_levels=3, blur=[0, 3, 0], max_iter=[1, 1, 1], min_grad=[0, 0, 0])
    j1p, _, _ = oracle.build_target_pyramids(p["gray1"], cfg2)
    np.testing.assert_array_equal(j1p[0], raw)
    np.testing.assert_array_equal(j1p[2], oracle.resize_level(raw, 2))


def half_pixel_problem(w=64, h=48, sign=1.0):
    K, i0, d0, i1, state = synthetic.half_pixel_problem(w, h, sign)
    gx, gy = twin.scharr(i1, 0.0625)
    return K, i0, d0, i1, gx, gy, state


def test_exact_half_pixel_projections_round_half_away_from_zero():
    """Known answer for the rounding rule the whole scatter hangs on: with every projection on an exact half, the
    residual slot of target (r + 1, c + 1) holds I1(r + 1, c + 1) - I0(r, c) (positive halves round up), and with the
    sign flipped target (r, c) holds I1(r, c) - I0(r, c) except in row 0 / column 0, whose sources project to -0.5,
    round to -1 and are dropped."""
    K, i0, d0, i1, gx, gy, state = half_pixel_problem()
    h, w = i0.shape
    r, J = oracle.compute_residuals_and_jacobians(i0, d0, i1, gx, gy, 0, K, state)
    r = r.reshape(h, w)
    np.testing.assert_array_equal(r[1:, 1:], i1[1:, 1:] - i0[:-1, :-1])
    assert np.all(r[0, :] == 0) and np.all(r[:, 0] == 0)
    J = J.reshape(6, h, w)
    assert np.all(J[:, -1, :] == 0) and np.all(J[:, :, -1] == 0)         # last row / column project out of bounds
    assert np.all(np.any(J[:, 1:-1, 1:-1] != 0, axis=0))               # (row 0 / column 0: the reflected Scharr gradient can be 0)
    K, i0, d0, i1, gx, gy, state = half_pixel_problem(sign=-1.0)
    r, J = oracle.compute_residuals_and_jacobians(i0, d0, i1, gx, gy, 0, K, state)
    r, J = r.reshape(h, w), J.reshape(6, h, w)
    np.testing.assert_array_equal(r[1:, 1:], i1[1:, 1:] - i0[1:, 1:])
    assert np.all(r[0, :] == 0) and np.all(r[:, 0] == 0)                 # round(-0.5) = -1: out of bounds
    assert np.all(J[:, 0, :] == 0) and np.all(J[:, :, 0] == 0)
    # the twin (np.sign * floor(|x| + 0.5)) agrees
    g, H, rt, Jt = twin.normal_equations((i0, d0, i1, gx, gy), 0, K, state)
    np.testing.assert_allclose(rt.reshape(h, w), r, atol=0)


def test_two_instruction_round_of_the_device_is_c_round_exactly():
    """The kernels round a projected coordinate v > -0.5 (all that the bounds test admits, ...Analytic.h:297-303) as
    floor(v + p), p = the largest double below 0.5 (csrc/gn_device.hpp, round_half_up_from): two fp64 instructions.
    This is C round() -- half away from zero -- for EVERY such double, not approximately: checked here against exact
    rational arithmetic on every neighbour (six doubles either side) of every tie, integer and quarter point up to
    image sizes and beyond, and on 400 000 random values.  The obvious floor(v + 0.5) fails at 0.49999999999999994."""
    import math
    from fractions import Fraction
    p = np.nextafter(0.5, 0.0)

    def exact_round(v):
        f = Fraction(float(v))
        return math.floor(f + Fraction(1, 2)) if f >= 0 else -math.floor(-f + Fraction(1, 2))

    cands = []
    for n in list(range(0, 70)) + [127, 128, 255, 256, 511, 512, 1023, 1024, 1279, 1280, 2047, 2048, 65535, 65536, 2 ** 20, 2 ** 30]:
        for base in (n, n + 0.25, n + 0.5, n + 0.75, n + 1.0):
            for k in range(-6, 7):
                y = np.float64(base)
                for _ in range(abs(k)):
                    y = np.nextafter(y, np.inf if k > 0 else -np.inf)
                cands.append(float(y))
    rng = np.random.RandomState(0)
    cands += list(rng.uniform(-0.5, 2000.0, 200000)) + list(rng.uniform(-0.5, 1.5, 200000))
    cands += [float(np.nextafter(-0.5, 0.0)), -0.25, -1e-300, 0.0, 5e-324]
    checked = 0
    for v in cands:
        if v > -0.5:
            assert float(np.floor(np.float64(v) + p)) == exact_round(v), repr(v)
            checked += 1
    assert checked > 400000
    bad = float(np.nextafter(0.5, 0.0))
    assert float(np.floor(np.float64(bad) + 0.5)) == 1.0 and exact_round(bad) == 0          # why p is not 0.5


def test_no_baseline_shape_reaches_what_the_oracle_cannot_vouch_for():
    """oracle/phovo_oracle.c restates OpenCV's resize / GaussianBlur from recalled behaviour; the branches nothing here can
    check -- the clipped 2x2 block of an odd-sized scale-2 resize, a bilinear tap clipped to the last row / column, the
    Gaussian blur -- are tagged UNVERIFIED-vs-OpenCV and count their executions.  Every shape BASELINE.json names (640x480
    with each shipped analytic yml, 1280x960 with the 6-level file: every level of the file, active or not, source and target
    pyramids) stays out of all three; the counters themselves work (an odd size, a 7th level of 480 rows and a blur hit one
    each)."""
    import glob
    import os
    from phovo_amd import native
    cfg_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config_files")
    ymls = sorted(glob.glob(os.path.join(cfg_dir, "*analytic.yml")))
    assert len(ymls) == 4
    shapes = [(640, 480, y) for y in ymls] + [(1280, 960, os.path.join(cfg_dir, "config_6_level_optimization_analytic.yml"))]
    oracle.unverified_hits(reset=True)
    rs = np.random.RandomState(0)
    for w, h, yml in shapes:
        n = native.read_config_file(yml)
        nl = n.num_levels
        cfg = oracle.make_config(num_levels=nl, blur=list(n.blur_filter_size[:nl]),
                                 grad_scale=list(n.image_gradients_scaling_factor[:nl]),
                                 max_iter=list(n.max_num_iterations[:nl]), min_grad=list(n.min_gradient_norm[:nl]))
        gray = rs.randint(0, 256, size=(h, w)).astype(np.uint8)
        depth = rs.uniform(0.5, 4.0, size=(h, w))
        oracle.build_source_pyramids(gray, depth, cfg)
        oracle.build_target_pyramids(gray, cfg)
        assert oracle.unverified_hits() == (0, 0, 0), (w, h, os.path.basename(yml), oracle.unverified_hits())
    # ... and the tagged branches are real: shapes outside BASELINE do reach them
    img = rs.uniform(size=(53, 75))
    oracle.resize_level(img, 1)                                    # odd size, scale 2: the clipped block
    assert oracle.unverified_hits(reset=True)[0] > 0
    oracle.resize_level(rs.uniform(size=(480, 640)), 6)            # 480 / 64 = 7.5 -> 8 rows: the last one is a clipped tap
    assert oracle.unverified_hits(reset=True)[1] > 0
    cfg = oracle.make_config(num_levels=1, blur=[5], max_iter=[1], min_grad=[0.0])
    oracle.build_target_pyramids(rs.randint(0, 256, size=(48, 64)).astype(np.uint8), cfg)
    assert oracle.unverified_hits(reset=True) == (0, 0, 1)

"""GPU tests of the extensions that are NOT in the reference (BASELINE.json configs[4]): narrow plane storage
and Huber IRLS weights.  Parity is against the oracle extended identically (never against the reference):
the oracle is fed the planes exactly as the device stored them (read back as fp64), so the only differences are
the usual fp64 ones -- same 1e-9 bar as the reference-exact path."""
import os

import numpy as np
import pytest

import phovo_amd  # noqa: F401
from phovo_amd import native, odometry, se3, synthetic
from oracle import oracle

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_DIR = os.path.join(ROOT, "config_files")
POSE_TOL = 1e-9


def _cfg_pair(yml, fixed_cap=None):
    n = native.read_config_file(os.path.join(CFG_DIR, yml))
    nl = n.num_levels
    mi = list(n.max_num_iterations[:nl])
    mg = list(n.min_gradient_norm[:nl])
    if fixed_cap is not None:
        mi = [min(m, fixed_cap) for m in mi]
        mg = [0.0] * nl
    return (native.make_config(num_levels=nl, max_iter=mi, min_grad=mg),
            oracle.make_config(num_levels=nl, max_iter=mi, min_grad=mg), nl, mi)


def _stored_planes(eng, src, tgt, nl, mi, w, h):
    """Oracle inputs = exactly what the device holds (rounded to the storage type), levels it does not hold: zeros."""
    i0p, d0p, i1p, gxp, gyp = [], [], [], [], []
    for l in range(nl):
        if mi[l] > 0:
            i0, d0, _, _ = eng.get_level_planes(src, l)
            i1, _, gx, gy = eng.get_level_planes(tgt, l)
        else:
            lw, lh = oracle.level_size(w, h, l)
            i0 = d0 = i1 = gx = gy = np.zeros((lh, lw))
        i0p.append(i0); d0p.append(d0); i1p.append(i1); gxp.append(gx); gyp.append(gy)
    return i0p, d0p, i1p, gxp, gyp


@pytest.mark.parametrize("storage", [native.STORAGE_F32, native.STORAGE_F16])
def test_narrow_storage_rounds_once_and_matches_oracle(storage):
    p = synthetic.make_pair(21, 640, 480, holes=0.02)
    ncfg, ocfg, nl, mi = _cfg_pair("config_4_level_optimization_analytic.yml")
    ref_i0, ref_d0 = oracle.build_source_pyramids(p["gray0"], p["depth0"], ocfg)
    ref_i1, ref_gx, ref_gy = oracle.build_target_pyramids(p["gray1"], ocfg)
    with odometry.AlignmentEngine() as eng:
        eng.set_config(ncfg)
        eng.set_extensions(native.make_extensions(plane_storage=storage))
        eng.set_intrinsic_matrix(p["K"])
        eng.reserve_frames(2, 640, 480)
        eng.upload_frame(0, p["gray0"], p["depth0"])
        eng.upload_frame(1, p["gray1"], p["depth1"])
        i0p, d0p, i1p, gxp, gyp = _stored_planes(eng, 0, 1, nl, mi, 640, 480)
        # stored planes = fp64 pyramids rounded ONCE to the storage type (fp16 goes through fp32)
        img_t = (lambda a: a.astype(np.float32).astype(np.float64)) if storage == native.STORAGE_F32 else \
                (lambda a: a.astype(np.float32).astype(np.float16).astype(np.float64))
        dep_t = lambda a: a.astype(np.float32).astype(np.float64)
        for l in range(nl):
            if mi[l] == 0:
                continue
            np.testing.assert_array_equal(i0p[l], img_t(ref_i0[l]))
            np.testing.assert_array_equal(d0p[l], dep_t(ref_d0[l]))
            np.testing.assert_array_equal(gxp[l], img_t(ref_gx[l]))
            np.testing.assert_array_equal(gyp[l], img_t(ref_gy[l]))
        es, eits = oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp)
        s, reps = eng.align_pairs([0], [1], want_reports=True)
        # set_level_planes rounds the same way: feeding the unrounded oracle planes gives the same stored bits
        eng.set_level_planes(0, 3, intensity=ref_i0[3], depth=ref_d0[3])
        a, b, _, _ = eng.get_level_planes(0, 3)
        np.testing.assert_array_equal(a, i0p[3])
        np.testing.assert_array_equal(b, d0p[3])
    assert list(reps[0].iterations[:nl]) == eits
    assert se3.state_distance(s[0], es) < POSE_TOL
    # and the narrow storage moved the answer only slightly away from the fp64 one
    s64, _ = oracle.optimize(ocfg, p["K"], ref_i0, ref_d0, ref_i1, ref_gx, ref_gy)
    assert se3.state_distance(s[0], s64) < (1e-4 if storage == native.STORAGE_F32 else 5e-2)


def test_huber_weights_match_extended_oracle():
    p = synthetic.make_pair(22, 640, 480, holes=0.02)
    g1 = p["gray1"].copy()
    g1[100:180, 200:330] = 255                                  # an occluder-like outlier block
    ncfg, ocfg, nl, mi = _cfg_pair("config_4_level_optimization_analytic.yml", fixed_cap=8)
    delta = [0.0, 0.0, 0.04, 0.06]
    i0p, d0p = oracle.build_source_pyramids(p["gray0"], p["depth0"], ocfg)
    i1p, gxp, gyp = oracle.build_target_pyramids(g1, ocfg)
    es, eits, etr = oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp, want_trace=True, huber_delta=delta)
    e0, _ = oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp)
    with odometry.AlignmentEngine() as eng:
        eng.set_config(ncfg)
        eng.set_intrinsic_matrix(p["K"])
        eng.reserve_frames(2, 640, 480)
        eng.upload_frame(0, p["gray0"], p["depth0"])
        eng.upload_frame(1, g1, p["depth1"])
        plain = eng.align_pairs([0], [1])
        eng.set_extensions(native.make_extensions(huber_delta=delta))     # same storage: the pool survives
        s, reps = eng.align_pairs([0], [1], want_reports=True)
        eng.set_extensions(native.make_extensions(huber_delta=[1e9] * nl))
        huge = eng.align_pairs([0], [1])
    assert list(reps[0].iterations[:nl]) == eits
    assert se3.state_distance(s[0], es) < POSE_TOL
    g_last = np.linalg.norm(etr[-1]["gradient"])
    assert abs(reps[0].gradient_norm - g_last) <= 1e-9 * max(1.0, g_last)
    assert se3.state_distance(plain[0], e0) < POSE_TOL
    assert se3.state_distance(s[0], plain[0]) > 1e-5             # the weights changed the estimate
    assert se3.state_distance(huge[0], plain[0]) < 1e-12         # delta -> inf is the least-squares path


def test_config5_shape_huber_fp16_through_yml_and_class_surface(tmp_path):
    """BASELINE.json configs[4]: 1280x960, config_6_level_optimization_analytic.yml + Huber + fp16 pyramids,
    switched on by the two optional yml keys and driven through the class-shaped surface."""
    base = open(os.path.join(CFG_DIR, "config_6_level_optimization_analytic.yml")).read()
    yml = tmp_path / "config_6_level_huber_fp16.yml"
    delta = [0.0, 0.0, 0.05, 0.05, 0.08, 0.1]
    yml.write_text(base + "huber_delta (at each level): " + str(delta) + "\nplane_storage_bits: 16\n")
    p = synthetic.make_pair(23, 1280, 960, holes=0.02)
    n = native.read_config_file(str(yml))
    nl = n.num_levels
    mi = list(n.max_num_iterations[:nl])
    ocfg = oracle.make_config(num_levels=nl, max_iter=mi, min_grad=list(n.min_gradient_norm[:nl]))
    with odometry.AlignmentEngine() as eng:                      # the stored (fp16 / fp32) planes for the oracle
        eng.read_configuration_file(str(yml))
        assert eng.get_extensions().plane_storage == native.STORAGE_F16
        eng.reserve_frames(2, 1280, 960)
        eng.upload_frame(0, p["gray0"], p["depth0"])
        eng.upload_frame(1, p["gray1"], p["depth1"])
        planes = _stored_planes(eng, 0, 1, nl, mi, 1280, 960)
    es, eits = oracle.optimize(ocfg, p["K"], *planes, huber_delta=delta)
    with odometry.CPhotoconsistencyOdometryAnalytic() as po:
        po.ReadConfigurationFile(str(yml))
        po.SetIntrinsicMatrix(p["K"])
        po.SetSourceFrame(p["gray0"], p["depth0"])
        po.SetTargetFrame(p["gray1"], p["depth1"])
        po.SetInitialStateVector(np.zeros(6))
        po.Optimize()
        s = po.GetOptimalStateVector()
        rep = po.GetReport()
    assert list(rep.iterations[:nl]) == eits
    assert se3.state_distance(s, es) < POSE_TOL


@pytest.mark.parametrize("corrected,storage,huber", [
    (False, native.STORAGE_F64, None),
    (True, native.STORAGE_F64, None),
    (True, native.STORAGE_F32, None),
    (True, native.STORAGE_F16, [0.0, 0.0, 0.05, 0.05]),
])
def test_bilinear_sampling_matches_extended_oracle(corrected, storage, huber):
    """PHOVO_SAMPLING_BILINEAR (single-pass kernel, no owner map) against the oracle's bilinear mode, alone and
    combined with the corrected Jacobian, narrow storages and Huber weights."""
    p = synthetic.make_pair(24, 640, 480, holes=0.02)
    ncfg, ocfg, nl, mi = _cfg_pair("config_4_level_optimization_analytic.yml", fixed_cap=10)
    with odometry.AlignmentEngine() as eng:
        eng.set_config(ncfg)
        eng.set_extensions(native.make_extensions(plane_storage=storage, huber_delta=huber,
                                                  sampling=native.SAMPLING_BILINEAR, jacobian_corrected=corrected))
        eng.set_intrinsic_matrix(p["K"])
        eng.reserve_frames(2, 640, 480)
        eng.upload_frame(0, p["gray0"], p["depth0"])
        eng.upload_frame(1, p["gray1"], p["depth1"])
        planes = _stored_planes(eng, 0, 1, nl, mi, 640, 480)
        s, reps = eng.align_pairs([0, 0], [1, 1], want_reports=True)
        assert [r["kind"] for r in eng.last_launches()] == ["bilinear"] * 2
    es, eits, etr = oracle.optimize(ocfg, p["K"], *planes, want_trace=True, huber_delta=huber,
                                    bilinear=True, corrected=corrected)
    assert list(reps[0].iterations[:nl]) == eits
    assert se3.state_distance(s[0], es) < POSE_TOL
    assert np.array_equal(s[0], s[1])
    # bilinear alignment converges: the last gradient is a small difference of large sums, so the fp64 noise of the
    # sums (relative to the FIRST gradient's size) is what bounds the error, not the last gradient's own size
    g_last = np.linalg.norm(etr[-1]["gradient"])
    g_scale = max(np.linalg.norm(e["gradient"]) for e in etr)
    assert abs(reps[0].gradient_norm - g_last) <= 1e-8 * max(1.0, g_scale)


def test_bilinear_handles_levels_of_any_size_and_rejects_bad_combinations():
    p = synthetic.make_pair(25, 640, 480, holes=0.01, trans=0.004, rot=0.002)
    ncfg = native.make_config(num_levels=1, max_iter=[5], min_grad=[0.0])        # 307200 px in one level
    ocfg = oracle.make_config(num_levels=1, max_iter=[5], min_grad=[0.0])
    i0p, d0p = oracle.build_source_pyramids(p["gray0"], p["depth0"], ocfg)
    i1p, gxp, gyp = oracle.build_target_pyramids(p["gray1"], ocfg)
    es, eits = oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp, bilinear=True, corrected=True)
    with odometry.AlignmentEngine() as eng:
        eng.set_config(ncfg)
        eng.set_extensions(native.make_extensions(sampling=native.SAMPLING_BILINEAR, jacobian_corrected=True))
        eng.set_intrinsic_matrix(p["K"])
        eng.reserve_frames(2, 640, 480)
        eng.upload_frame(0, p["gray0"], p["depth0"])
        eng.upload_frame(1, p["gray1"], p["depth1"])
        s, reps = eng.align_pairs([0], [1], want_reports=True)
        with pytest.raises(native.PhovoError) as ei:         # the scatter path stays reference-exact
            eng.set_extensions(native.make_extensions(jacobian_corrected=True))
        assert ei.value.status == 7
    assert list(reps[0].iterations[:1]) == eits
    assert se3.state_distance(s[0], es) < POSE_TOL


@pytest.mark.parametrize("storage", [native.STORAGE_F64, native.STORAGE_F32, native.STORAGE_F16])
@pytest.mark.parametrize("side", [-1.0, 1.0])
def test_bilinear_taps_in_the_outer_half_pixel_band(storage, side):
    """Pixels whose warped column lies in (-1/2, 0) or [W - 1, W - 1/2) take the edge pixel for both taps of a row.  The three
    storages do that in three ways (fp64: a 16-byte pair loaded one column inside with the row's weight set to 0 or 1; fp32:
    clamped tap addresses; fp16: the tap record of the edge pixel): the initial state shifts the whole image by a third of
    a pixel to one side, so every row has a pixel in that band, on a small image where the band is 1 / 72 of all pixels."""
    w, h = 72, 56
    p = synthetic.make_pair(31, w, h, holes=0.0, trans=0.002, rot=0.001)
    ncfg = native.make_config(num_levels=2, max_iter=[4, 4], min_grad=[0.0, 0.0])
    ocfg = oracle.make_config(num_levels=2, max_iter=[4, 4], min_grad=[0.0, 0.0])
    z = p["depth0"][p["depth0"] > 0]
    tx = side * 0.3 * float(np.median(z)) / p["K"][0, 0]
    init = np.array([tx, 0.0, 0.0, 0.0, 0.0, 0.0])
    shift = p["K"][0, 0] * tx / z
    assert np.mean((np.abs(shift) > 0.05) & (np.abs(shift) < 0.5)) > 0.9     # the edge column of nearly every row is in the band
    with odometry.AlignmentEngine() as eng:
        eng.set_config(ncfg)
        eng.set_extensions(native.make_extensions(plane_storage=storage, sampling=native.SAMPLING_BILINEAR,
                                                  jacobian_corrected=True))
        eng.set_intrinsic_matrix(p["K"])
        eng.reserve_frames(2, w, h)
        eng.upload_frame(0, p["gray0"], p["depth0"])
        eng.upload_frame(1, p["gray1"], p["depth1"])
        planes = _stored_planes(eng, 0, 1, 2, [4, 4], w, h)
        s, reps = eng.align_pairs([0], [1], init_states=init[None, :], want_reports=True)
        assert [r["kind"] for r in eng.last_launches()] == ["bilinear"] * 2
    es, eits = oracle.optimize(ocfg, p["K"], *planes, init_state=init, bilinear=True, corrected=True)
    assert list(reps[0].iterations[:2]) == eits
    assert se3.state_distance(s[0], es) < POSE_TOL


@pytest.mark.parametrize("storage", [native.STORAGE_F64, native.STORAGE_F32, native.STORAGE_F16])
@pytest.mark.parametrize("w,h", [(1, 40), (2, 40), (3, 37), (40, 1), (40, 2), (5, 5)])
def test_bilinear_on_images_a_few_pixels_wide(storage, w, h):
    """One- to three-column (and -row) images: every tap is an edge tap, the fp64 form's 16-byte pair does not fit a one-column
    row at all (both taps are that column; the pair is loaded at column 0 and weighted 1 : 0).  Such normal equations are rank
    deficient more often than not: then both sides must say so (non-finite state, PAIR_NONFINITE) instead of agreeing on
    numbers."""
    p = synthetic.make_pair(33, max(w, 8), max(h, 8), holes=0.0, trans=0.002, rot=0.001)
    g0, d0, g1 = p["gray0"][:h, :w].copy(), p["depth0"][:h, :w].copy(), p["gray1"][:h, :w].copy()
    K = synthetic.intrinsics(w, h)
    ncfg = native.make_config(num_levels=1, max_iter=[3], min_grad=[0.0])
    ocfg = oracle.make_config(num_levels=1, max_iter=[3], min_grad=[0.0])
    with odometry.AlignmentEngine() as eng:
        eng.set_config(ncfg)
        eng.set_extensions(native.make_extensions(plane_storage=storage, sampling=native.SAMPLING_BILINEAR,
                                                  jacobian_corrected=True))
        eng.set_intrinsic_matrix(K)
        eng.reserve_frames(2, w, h)
        eng.upload_frame(0, g0, d0)
        eng.upload_frame(1, g1, d0)
        planes = _stored_planes(eng, 0, 1, 1, [3], w, h)
        s, reps = eng.align_pairs([0, 0, 0], [1, 1, 1], want_reports=True)
    es, eits = oracle.optimize(ocfg, K, *planes, bilinear=True, corrected=True)
    assert np.array_equal(s[0], s[1], equal_nan=True) and np.array_equal(s[0], s[2], equal_nan=True)
    if np.all(np.isfinite(es)):
        assert list(reps[0].iterations[:1]) == eits
        # (a handful of pixels: the normal equations are as ill-conditioned as they come; the bar scales as in tests/tools/fuzz_parity.py)
        assert se3.state_distance(s[0], es) < 1e-5
    else:
        assert reps[0].flags & native.PAIR_NONFINITE and not np.all(np.isfinite(s[0]))

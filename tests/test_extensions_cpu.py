"""Extensions that are NOT in the reference (BASELINE.json configs[4]: Huber weights, narrow plane storage):
CPU-side checks -- the oracle's Huber IRLS against the independent numpy twin, and the optional yml keys."""
import os

import numpy as np
import pytest

import phovo_amd  # noqa: F401
from phovo_amd import native, se3, synthetic
from oracle import oracle, numpy_twin as twin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "config_files")


def _problem():
    p = synthetic.make_pair(17, 96, 72, holes=0.03)
    # a few gross outliers in the target, the thing robust weights are for
    g1 = p["gray1"].copy()
    g1[10:20, 30:44] = 255
    return p, g1


def test_oracle_huber_matches_twin():
    p, g1 = _problem()
    nl, max_iter, delta = 3, [2, 4, 6], [0.02, 0.05, 0.05]
    ocfg = oracle.make_config(num_levels=nl, max_iter=max_iter, min_grad=[0.0] * nl)
    i0p, d0p = oracle.build_source_pyramids(p["gray0"], p["depth0"], ocfg)
    i1p, gxp, gyp = oracle.build_target_pyramids(g1, ocfg)
    s1, it1, tr1 = oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp, want_trace=True, huber_delta=delta)
    pyr = twin.build_pyramids(p["gray0"], p["depth0"], g1, nl, [0.0625] * nl)
    s2, it2, tr2 = twin.optimize(pyr, p["K"], dict(num_levels=nl, lam=[1.0] * nl, max_iter=max_iter,
                                                    min_grad=[0.0] * nl, huber_delta=delta))
    assert it1 == it2 and len(tr1) == len(tr2)
    for a, b in zip(tr1, tr2):
        np.testing.assert_allclose(a["gradient"], b["gradient"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(a["hessian"], b["hessian"], rtol=1e-9, atol=1e-9)
    assert se3.state_distance(s1, s2) < 1e-9
    # the weights do something: the plain least-squares result differs
    s0, _ = oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp)
    assert se3.state_distance(s0, s1) > 1e-5


def test_huber_off_and_huge_delta_equal_the_reference_path():
    p, g1 = _problem()
    nl = 3
    ocfg = oracle.make_config(num_levels=nl, max_iter=[2, 3, 4], min_grad=[0.0] * nl)
    i0p, d0p = oracle.build_source_pyramids(p["gray0"], p["depth0"], ocfg)
    i1p, gxp, gyp = oracle.build_target_pyramids(g1, ocfg)
    s0, it0 = oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp)
    s1, it1 = oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp, huber_delta=[0.0, -1.0, 0.0])
    s2, it2 = oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp, huber_delta=[1e9] * nl)
    assert it0 == it1 == it2
    assert np.array_equal(s0, s1) and np.array_equal(s0, s2)


def test_extension_keys_are_optional_in_the_yml(tmp_path):
    for name in sorted(os.listdir(CFG)):                       # the reference's files: everything off
        ext = native.read_extensions_file(os.path.join(CFG, name))
        assert ext.plane_storage == native.STORAGE_F64
        assert all(v == 0.0 for v in ext.huber_delta)
    base = open(os.path.join(CFG, "config_6_level_optimization_analytic.yml")).read()
    p = tmp_path / "config_6_level_huber_fp16.yml"
    p.write_text(base + "huber_delta (at each level): [0.1, 0.1, 0.05, 0.05, 0.05, 0.05]\nplane_storage_bits: 16\n")
    cfg = native.read_config_file(str(p))                       # the reference keys still parse
    assert cfg.num_levels == 6 and list(cfg.max_num_iterations[:6]) == [0, 0, 5, 20, 50, 50]
    ext = native.read_extensions_file(str(p))
    assert ext.plane_storage == native.STORAGE_F16
    assert list(ext.huber_delta[:6]) == [0.1, 0.1, 0.05, 0.05, 0.05, 0.05]
    bad = tmp_path / "bad.yml"
    bad.write_text(base + "plane_storage_bits: 8\n")
    with pytest.raises(native.PhovoError) as ei:
        native.read_extensions_file(str(bad))
    assert ei.value.status == 2


@pytest.mark.parametrize("corrected", [False, True])
def test_oracle_bilinear_matches_twin(corrected):
    """Bilinear forward-additive mode: the oracle (reference temps, optional `+x` slip) against the twin, which
    builds the Jacobian from the rotated point's derivatives instead -- two derivations, one answer."""
    p = synthetic.make_pair(31, 96, 72, holes=0.03)
    nl, mi = 3, [3, 5, 8]
    ocfg = oracle.make_config(num_levels=nl, max_iter=mi, min_grad=[0.0] * nl)
    i0p, d0p = oracle.build_source_pyramids(p["gray0"], p["depth0"], ocfg)
    i1p, gxp, gyp = oracle.build_target_pyramids(p["gray1"], ocfg)
    pyr = twin.build_pyramids(p["gray0"], p["depth0"], p["gray1"], nl, [0.0625] * nl)
    s1, it1, tr1 = oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp, want_trace=True,
                                   bilinear=True, corrected=corrected, huber_delta=[0.0, 0.03, 0.03])
    s2, it2, tr2 = twin.optimize(pyr, p["K"], dict(num_levels=nl, lam=[1.0] * nl, max_iter=mi, min_grad=[0.0] * nl,
                                                    bilinear=True, corrected=corrected, huber_delta=[0.0, 0.03, 0.03]))
    assert it1 == it2
    for a, b in zip(tr1, tr2):
        np.testing.assert_allclose(a["gradient"], b["gradient"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(a["hessian"], b["hessian"], rtol=1e-9, atol=1e-8)
    assert se3.state_distance(s1, s2) < 1e-9
    # and it is the better estimator on this synthetic pair than the reference's nearest/scatter formulation
    s0, _ = oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp)
    assert se3.state_distance(s1, p["motion"]) < se3.state_distance(s0, p["motion"])


def test_sampling_keys_in_yml(tmp_path):
    base = open(os.path.join(CFG, "config_4_level_optimization_analytic.yml")).read()
    p = tmp_path / "b.yml"
    p.write_text(base + "sampling_bilinear: 1\njacobian_corrected: 1\n")
    ext = native.read_extensions_file(str(p))
    assert ext.sampling == native.SAMPLING_BILINEAR and ext.jacobian_corrected == 1
    bad = tmp_path / "bad2.yml"
    bad.write_text(base + "sampling_bilinear: 2\n")
    with pytest.raises(native.PhovoError):
        native.read_extensions_file(str(bad))

"""The C oracle (oracle/phovo_oracle.c) against the committed golden fixtures.

The fixtures were produced by the independent numpy restatement (tests/golden/make_golden.py);
the reference holds none, so this is what pins the oracle ("parity unpinned" w.r.t. the
reference itself -- see oracle/phovo_oracle.h).
"""
import glob
import os

import numpy as np
import pytest

import phovo_amd  # noqa: F401
from phovo_amd import se3
from oracle import oracle

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "case_*.npz")))


def _cfg(d):
    nl = int(d["num_levels"])
    return oracle.make_config(num_levels=nl, blur=[0] * nl, grad_scale=d["grad_scale"],
                              lam=d["lam"], max_iter=d["max_iter"], min_grad=d["min_grad"],
                              min_depth=float(d["min_depth"]), max_depth=float(d["max_depth"]))


def test_fixtures_present():
    assert len(CASES) >= 3


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_pyramids_match_twin(path):
    d = np.load(path)
    cfg = _cfg(d)
    i0p, d0p = oracle.build_source_pyramids(d["gray0"], d["depth0"], cfg)
    i1p, gxp, gyp = oracle.build_target_pyramids(d["gray1"], cfg)
    for l in (cfg.num_levels - 1, 1):
        # bit-exact: same operations in the same order on both sides
        np.testing.assert_array_equal(i0p[l], d[f"exp_L{l}_i0"])
        np.testing.assert_array_equal(d0p[l], d[f"exp_L{l}_d0"])
        np.testing.assert_array_equal(gxp[l], d[f"exp_L{l}_gx1"])
        np.testing.assert_array_equal(gyp[l], d[f"exp_L{l}_gy1"])


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_optimize_matches_twin(path):
    d = np.load(path)
    cfg = _cfg(d)
    state, iters, trace = oracle.align_frames(cfg, d["K"], d["gray0"], d["depth0"], d["gray1"],
                                              init_state=d["init_state"], want_trace=True)
    assert iters == list(d["exp_iters"])
    assert len(trace) == len(d["exp_trace_level"])
    for k, e in enumerate(trace):
        assert e["level"] == d["exp_trace_level"][k]
        assert e["iteration"] == d["exp_trace_iteration"][k]
        # fp64 sums over <= 12288 pixels in two different orders: 1e-11 relative
        np.testing.assert_allclose(e["gradient"], d["exp_trace_gradient"][k], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(e["hessian"], d["exp_trace_hessian"][k], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(e["state"], d["exp_trace_state"][k], rtol=0, atol=1e-9)
    assert se3.state_distance(state, d["exp_state"]) < 1e-9

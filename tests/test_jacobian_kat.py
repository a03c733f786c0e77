"""Known-answer test of the warp Jacobian against the reference's own derivation.

phovo/Maxima/derivatives_photoconsistency.wxm:5-19 defines the forward model the analytic
Jacobian was transcribed from (Rt(x,y,z,yaw,pitch,roll), transformedPoint2D).  The model is
restated here in sympy and differentiated; the oracle's per-pixel Jacobian
(...Analytic.h:312-342) must equal the true derivative for 9 of the 12 entries, and for
(u,z), (u,pitch), (u,roll) must differ from it by exactly the `temp11 = cos(pitch)*cos(yaw)+x`
transcription bug (:253): reference - true = -fx * x*(px-1)/Z^2 * {1, dZ/dpitch, dZ/droll}.
This is the one fixture the reference tree itself offers for this path.
"""
import numpy as np
import pytest

sympy = pytest.importorskip("sympy")

from oracle import oracle  # noqa: E402


def _model():
    x, y, z, yaw, pitch, roll, px, py, pz, fx, fy, ox, oy = sympy.symbols(
        "x y z yaw pitch roll px py pz fx fy ox oy", real=True)
    c, s = sympy.cos, sympy.sin
    Rt = sympy.Matrix([
        [c(yaw) * c(pitch), c(yaw) * s(pitch) * s(roll) - s(yaw) * c(roll), c(yaw) * s(pitch) * c(roll) + s(yaw) * s(roll), x],
        [s(yaw) * c(pitch), s(yaw) * s(pitch) * s(roll) + c(yaw) * c(roll), s(yaw) * s(pitch) * c(roll) - c(yaw) * s(roll), y],
        [-s(pitch), c(pitch) * s(roll), c(pitch) * c(roll), z],
        [0, 0, 0, 1]])
    P = Rt * sympy.Matrix([px, py, pz, 1])
    u = P[0] * fx / P[2] + ox
    v = P[1] * fy / P[2] + oy
    params = (x, y, z, yaw, pitch, roll)
    syms = params + (px, py, pz, fx, fy, ox, oy)
    Ju = [sympy.diff(u, p) for p in params]
    Jv = [sympy.diff(v, p) for p in params]
    dZ = [sympy.diff(P[2], p) for p in params]
    f = sympy.lambdify(syms, Ju + Jv + dZ + [P[2]], "math")
    return f


def test_oracle_jacobian_against_maxima_model():
    f = _model()
    rs = np.random.RandomState(5)
    W, H = 9, 7
    K = np.array([[60.0, 0, 4.0], [0, 55.0, 3.0], [0, 0, 1]])
    for trial in range(6):
        state = np.concatenate([rs.uniform(-0.02, 0.02, 3), rs.uniform(-0.02, 0.02, 3)])
        d0 = rs.uniform(1.0, 3.0, (H, W))
        z = np.zeros((H, W))
        ones = np.ones((H, W))
        _, Ju = oracle.compute_residuals_and_jacobians(z, d0, z, ones, z, 0, K, state)
        _, Jv = oracle.compute_residuals_and_jacobians(z, d0, z, z, ones, 0, K, state)
        checked = 0
        for r in range(H):
            for c in range(W):
                i = r * W + c
                if not np.any(Ju[:, i]):
                    continue                       # warped out of bounds: zero row
                pz = d0[r, c]
                px = (c - K[0, 2]) * pz / K[0, 0]
                py = (r - K[1, 2]) * pz / K[1, 1]
                out = f(*state, px, py, pz, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
                tu, tv, dZ, Z = np.array(out[0:6]), np.array(out[6:12]), np.array(out[12:18]), out[18]
                bug = -K[0, 0] * state[0] * (px - 1.0) / Z ** 2
                expect_u = tu.copy()
                expect_u[2] += bug * 1.0           # dZ/dz = 1
                expect_u[4] += bug * dZ[4]
                expect_u[5] += bug * dZ[5]
                np.testing.assert_allclose(Ju[:, i], expect_u, rtol=1e-11, atol=1e-11)
                np.testing.assert_allclose(Jv[:, i], tv, rtol=1e-11, atol=1e-11)
                # the bug is real: the three entries differ from the true derivative
                if abs(state[0]) > 1e-3 and abs(px - 1.0) > 1e-2:
                    assert abs(Ju[2, i] - tu[2]) > 1e-9
                checked += 1
        assert checked > 20


def test_bug_vanishes_at_zero_translation_x():
    f = _model()
    K = np.array([[60.0, 0, 4.0], [0, 55.0, 3.0], [0, 0, 1]])
    state = np.array([0.0, 0.01, -0.01, 0.01, -0.02, 0.015])
    H, W = 7, 9
    d0 = np.full((H, W), 2.0)
    z = np.zeros((H, W)); ones = np.ones((H, W))
    _, Ju = oracle.compute_residuals_and_jacobians(z, d0, z, ones, z, 0, K, state)
    r, c = 3, 5
    pz = 2.0
    px = (c - 4.0) * pz / 60.0
    py = (r - 3.0) * pz / 55.0
    out = f(*state, px, py, pz, 60.0, 55.0, 4.0, 3.0)
    np.testing.assert_allclose(Ju[:, r * W + c], out[0:6], rtol=1e-11, atol=1e-12)

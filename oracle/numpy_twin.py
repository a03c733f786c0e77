"""Independent numpy restatement of the analytic alignment path (vectorised, gather form).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (the reference holds no tests or golden
vectors; see oracle/phovo_oracle.h).  This twin exists so that the C oracle is checked
against something written separately from it: it resolves the residual scatter with an
explicit "largest source index wins" owner map instead of a serial loop, forms
J^T J / J^T r with numpy reductions and solves the 6x6 system with numpy.linalg.
tests/golden/make_golden.py runs it to produce the committed fixtures.

Follows phovo/include/CPhotoconsistencyOdometryAnalytic.h:191-367 (per pixel),
:376-392 (termination), :500-563 (GN loop), CPhotoconsistencyOdometry.h:47-71 (pose).
"""
import numpy as np


def level_size(w, h, level):
    f = 1.0 / (1 << level)
    return int(np.rint(w * f)), int(np.rint(h * f))


def resize_level(img, level):
    """cv::resize by 2^-L from level 0 (INTER_LINEAR; 2x -> area fast path).  Sizes divisible by 2^L only: the twin does
    not restate the clipped-border branches of phovo_oracle.c (tagged UNVERIFIED-vs-OpenCV there: recalled OpenCV behaviour
    that nothing in this repository can check and that no BASELINE shape reaches), so those are pinned by nothing."""
    if level == 0:
        return img.copy()
    s = 1 << level
    h, w = img.shape
    assert w % s == 0 and h % s == 0, "twin covers the divisible case only"
    if s == 2:
        a, b = img[0::2, 0::2], img[0::2, 1::2]
        c, d = img[1::2, 0::2], img[1::2, 1::2]
        return (((a + b) + c) + d) * 0.25
    o = s // 2 - 1
    a, b = img[o::s, o::s], img[o::s, o + 1::s]
    c, d = img[o + 1::s, o::s], img[o + 1::s, o + 1::s]
    return (a * 0.5 + b * 0.5) * 0.5 + (c * 0.5 + d * 0.5) * 0.5


def scharr(img, scale):
    """cv::Scharr dx / dy, reflect-101 border, smoothing kernel scaled."""
    k3, k10 = 3.0 * scale, 10.0 * scale
    p = np.pad(img, 1, mode="reflect")
    left, mid, right = p[:, :-2], p[:, 1:-1], p[:, 2:]
    tx = right - left                      # rows still padded vertically
    ty = ((k3 * left) + (k10 * mid)) + (k3 * right)
    gx = (k10 * tx[1:-1]) + (k3 * (tx[2:] + tx[:-2]))
    gy = ty[2:] - ty[:-2]
    return gx, gy


def gaussian_blur_twice(img, ksize):
    """cv::GaussianBlur(img, img, Size(k, k), 3) twice (...Analytic.h:146-147): separable, kernel
    exp(-(i - (k-1)/2)^2 / (2*3^2)) normalised, BORDER_REFLECT_101 (scipy's "mirror").  UNVERIFIED vs OpenCV like the C
    oracle's (kernel, normalisation, summation order, border are recalled); no shipped analytic yml sets blurFilterSize > 0."""
    if ksize <= 1:
        return img.copy()
    from scipy.ndimage import correlate1d
    x = np.arange(ksize) - (ksize - 1) * 0.5
    k = np.exp(-0.5 / 9.0 * x * x)
    k /= k.sum()
    out = img
    for _ in range(2):
        out = correlate1d(correlate1d(out, k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    return out


def intensity_pyramid(gray, num_levels, blur=None):
    """BuildPyramid(intensity, applyBlur = true) (:115-163).  Level 0 is a shallow alias of the converted image
    (`imgAux = img`, :136) and is blurred in place, so with blur[0] > 0 the later levels are resized from the
    blurred image, then blurred with their own filter size."""
    img = gray.astype(np.float64) * (1.0 / 255)
    pyr = []
    for l in range(num_levels):
        lv = resize_level(img, l)
        if blur is not None and blur[l] > 0:
            lv = gaussian_blur_twice(lv, int(blur[l]))
            if l == 0:
                img = lv
        pyr.append(lv)
    return pyr


def build_pyramids(gray0, depth0, gray1, num_levels, grad_scale, blur=None):
    p0 = intensity_pyramid(gray0, num_levels, blur)
    p1 = intensity_pyramid(gray1, num_levels, blur)
    out = []
    for l in range(num_levels):
        d = resize_level(depth0.astype(np.float64), l)
        gx, gy = scharr(p1[l], grad_scale[l])
        out.append((p0[l], d, p1[l], gx, gy))
    return out


def c_round(x):
    """C round(): half away from zero."""
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


def normal_equations(planes, level, K, state, min_depth=0.3, max_depth=5.0):
    """One pass of ComputeResidualsAndJacobians + J^T r, J^T J, in gather form.

    Returns (g[6], H[6,6], r[N], J[N,6])."""
    i0, d0, i1, gx1, gy1 = planes
    H_, W_ = i0.shape
    n = H_ * W_
    sf = 1.0 / 2 ** level
    fx, fy, ox, oy = K[0, 0] * sf, K[1, 1] * sf, K[0, 2] * sf, K[1, 2] * sf
    ifx, ify = 1.0 / fx, 1.0 / fy
    x, y, z, yaw, pitch, roll = state
    sy_, cy_ = np.sin(yaw), np.cos(yaw)
    sp, cp = np.sin(pitch), np.cos(pitch)
    sr, cr = np.sin(roll), np.cos(roll)
    R = np.array([[cy_ * cp, cy_ * sp * sr - sy_ * cr, cy_ * sp * cr + sy_ * sr],
                  [sy_ * cp, sy_ * sp * sr + cy_ * cr, sy_ * sp * cr - cy_ * sr],
                  [-sp, cp * sr, cp * cr]])
    t1, t2, t3 = cp * sr, cp * cr, sp
    t4 = sr * sy_ + sp * cr * cy_
    t5 = sp * sr * cy_ - cr * sy_
    t6 = sp * sr * sy_ + cr * cy_
    t7 = -sp * sr * sy_ - cr * cy_
    t8 = sr * cy_ - sp * cr * sy_
    t9 = sp * cr * sy_ - sr * cy_
    t10 = cp * sr * cy_
    t11 = cp * cy_ + x                       # reference transcription bug (:253)
    t12 = cp * cr * cy_
    t13, t14, t15 = sp * cy_, cp * sy_, cp * cy_
    t16, t17 = sp * sr, sp * cr
    t18, t19, t20 = cp * sr * sy_, cp * cr * sy_, sp * sy_
    t21 = cr * sy_ - sp * sr * cy_
    t22, t23, t24 = cp * cr, cp * sr, cp

    cc, rr = np.meshgrid(np.arange(W_, dtype=np.float64), np.arange(H_, dtype=np.float64))
    pz = d0.reshape(-1)
    valid = (min_depth < pz) & (pz < max_depth)
    with np.errstate(all="ignore"):
        px = (cc.reshape(-1) - ox) * pz * ifx
        py = (rr.reshape(-1) - oy) * pz * ify
        X = R[0, 0] * px + R[0, 1] * py + R[0, 2] * pz + x
        Y = R[1, 0] * px + R[1, 1] * py + R[1, 2] * pz + y
        Z = R[2, 0] * px + R[2, 1] * py + R[2, 2] * pz + z
        iz = 1.0 / Z
        tc = (X * fx) * iz + ox
        tr = (Y * fy) * iz + oy
        tri, tci = c_round(tr), c_round(tc)
        inb = valid & np.isfinite(tri) & np.isfinite(tci) & \
            (tri >= 0) & (tri < H_) & (tci >= 0) & (tci < W_)
        t25 = 1.0 / (z + py * t1 + pz * t2 - px * t3)
        t26 = t25 * t25
        A = pz * t4 + py * t5 + px * t11
        B = py * t6 + pz * t9 + px * t14 + y
        Cc = -py * t16 - pz * t17 - px * t24
        D = py * t22 - pz * t23
        zero = np.zeros(n)
        Ju = [fx * t25, zero, -fx * A * t26,
              fx * (py * t7 + pz * t8 - px * t14) * t25,
              fx * (py * t10 + pz * t12 - px * t13) * t25 - fx * Cc * A * t26,
              fx * (py * t4 + pz * t21) * t25 - fx * D * A * t26]
        Jv = [zero, fy * t25, -fy * B * t26,
              fy * (pz * t4 + py * t5 + px * t15) * t25,
              fy * (py * t18 + pz * t19 - px * t20) * t25 - fy * Cc * B * t26,
              fy * (pz * t7 + py * t9) * t25 - fy * D * B * t26]
        gxi, gyi = gx1.reshape(-1), gy1.reshape(-1)        # gradient at the SOURCE index (:346-347)
        J = np.stack([gxi * Ju[j] + gyi * Jv[j] for j in range(6)], axis=1)
    J[~inb] = 0.0

    # scatter r[W*tri+tci] = I1(tri,tci) - I0(i), later source pixels win (:358)
    src = np.nonzero(inb)[0]
    tgt = (tri[src] * W_ + tci[src]).astype(np.int64)
    owner = np.full(n, -1, dtype=np.int64)
    np.maximum.at(owner, tgt, src)
    r = np.zeros(n)
    has = owner >= 0
    r[has] = i1.reshape(-1)[has] - i0.reshape(-1)[owner[has]]

    g = J.T @ r
    Hm = J.T @ J
    return g, Hm, r, J


def scatter_statistics(planes, level, K, state, min_depth=0.3, max_depth=5.0):
    """How hard one pass works the reference's scatter (`residuals(nCols*tr+tc) = ...`, ...Analytic.h:358, last raster
    writer wins) on these planes: the number of source pixels that land in bounds, the target pixels hit, the fraction of
    hit targets that more than one source lands on, the largest number of sources on one target, and the largest distance
    (pixels) between two sources that collide: adjacent pixels of one surface collide at distance 1-2, pixels of
    DIFFERENT depth layers (occlusion) from further apart -- `far_collisions` counts the targets whose sources are more
    than 2 pixels apart.  Test coverage bookkeeping, not part of the algorithm."""
    i0, d0 = planes[0], planes[1]
    H_, W_ = i0.shape
    sf = 1.0 / 2 ** level
    fx, fy, ox, oy = K[0, 0] * sf, K[1, 1] * sf, K[0, 2] * sf, K[1, 2] * sf
    x, y, z, yaw, pitch, roll = state
    sy_, cy_, sp, cp, sr, cr = np.sin(yaw), np.cos(yaw), np.sin(pitch), np.cos(pitch), np.sin(roll), np.cos(roll)
    R = np.array([[cy_ * cp, cy_ * sp * sr - sy_ * cr, cy_ * sp * cr + sy_ * sr],
                  [sy_ * cp, sy_ * sp * sr + cy_ * cr, sy_ * sp * cr - cy_ * sr],
                  [-sp, cp * sr, cp * cr]])
    cc, rr = np.meshgrid(np.arange(W_, dtype=np.float64), np.arange(H_, dtype=np.float64))
    pz = d0.reshape(-1)
    valid = (min_depth < pz) & (pz < max_depth)
    with np.errstate(all="ignore"):
        px = (cc.reshape(-1) - ox) * pz / fx
        py = (rr.reshape(-1) - oy) * pz / fy
        P = np.stack([px, py, pz]) 
        X, Y, Z = R @ P + np.array([[x], [y], [z]])
        tci, tri = c_round(X * fx / Z + ox), c_round(Y * fy / Z + oy)
        inb = valid & np.isfinite(tri) & np.isfinite(tci) & (tri >= 0) & (tri < H_) & (tci >= 0) & (tci < W_)
    src = np.nonzero(inb)[0]
    tgt = (tri[src] * W_ + tci[src]).astype(np.int64)
    n = H_ * W_
    count = np.bincount(tgt, minlength=n)
    lo = np.full(n, n, dtype=np.int64)
    hi = np.full(n, -1, dtype=np.int64)
    np.minimum.at(lo, tgt, src)
    np.maximum.at(hi, tgt, src)
    hit = count > 0
    multi = count > 1
    # distance between the first and the last source of a target, in pixels (Chebyshev: rows or columns, whichever is larger)
    span = np.where(multi, np.maximum(hi // W_ - lo // W_, np.abs(hi % W_ - lo % W_)), 0)
    return dict(landed=int(inb.sum()), valid_fraction=float(valid.mean()), targets_hit=int(hit.sum()),
                collision_fraction=float(multi.sum()) / max(1, int(hit.sum())), max_sources_per_target=int(count.max()),
                max_collision_distance=int(span.max()), far_collisions=int((span > 2).sum()))


def normal_equations_bilinear(planes, level, K, state, min_depth=0.3, max_depth=5.0, corrected=False):
    """EXTENSION (not in the reference's analytic path): forward-additive residuals / Jacobians with bilinear
    sampling at the real-valued warped position; rows belong to the source pixel.  Returns (r[N], J[N,6])."""
    i0, d0, i1, gx1, gy1 = planes
    H_, W_ = i0.shape
    sf = 1.0 / 2 ** level
    fx, fy, ox, oy = K[0, 0] * sf, K[1, 1] * sf, K[0, 2] * sf, K[1, 2] * sf
    x, y, z, yaw, pitch, roll = state
    sy_, cy_, sp, cp, sr, cr = np.sin(yaw), np.cos(yaw), np.sin(pitch), np.cos(pitch), np.sin(roll), np.cos(roll)
    R = np.array([[cy_ * cp, cy_ * sp * sr - sy_ * cr, cy_ * sp * cr + sy_ * sr],
                  [sy_ * cp, sy_ * sp * sr + cy_ * cr, sy_ * sp * cr - cy_ * sr],
                  [-sp, cp * sr, cp * cr]])
    cc, rr = np.meshgrid(np.arange(W_, dtype=np.float64), np.arange(H_, dtype=np.float64))
    pz = d0.reshape(-1)
    valid = (min_depth < pz) & (pz < max_depth)
    with np.errstate(all="ignore"):
        px = (cc.reshape(-1) - ox) * pz / fx
        py = (rr.reshape(-1) - oy) * pz / fy
        P = R @ np.stack([px, py, pz]) + np.array([[x], [y], [z]])
        iz = 1.0 / P[2]
        tc, tr = P[0] * fx * iz + ox, P[1] * fy * iz + oy
        inb = valid & (tc > -0.5) & (tc < W_ - 0.5) & (tr > -0.5) & (tr < H_ - 0.5)
        tc, tr = np.where(inb, tc, 0.0), np.where(inb, tr, 0.0)
        fc, fr = np.floor(tc), np.floor(tr)
        ax, ay = tc - fc, tr - fr
        c0, c1 = np.clip(fc, 0, W_ - 1).astype(np.int64), np.clip(fc + 1, 0, W_ - 1).astype(np.int64)
        r0, r1 = np.clip(fr, 0, H_ - 1).astype(np.int64), np.clip(fr + 1, 0, H_ - 1).astype(np.int64)

        def smp(Pl):
            return (1 - ay) * ((1 - ax) * Pl[r0, c0] + ax * Pl[r0, c1]) + ay * ((1 - ax) * Pl[r1, c0] + ax * Pl[r1, c1])
        res = smp(i1) - i0.reshape(-1)
        gxs, gys = smp(gx1), smp(gy1)
        # true derivative of the projection w.r.t. (x,y,z,yaw,pitch,roll) via the rotated point and its derivatives
        Xr, Yr = P[0] - x, P[1] - y                        # rotated, untranslated
        dP_dyaw = np.stack([-Yr, Xr, np.zeros_like(Xr)])
        Zr = P[2] - z
        dP_dpitch = np.stack([cy_ * Zr, sy_ * Zr, -(cp * px + sp * sr * py + sp * cr * pz)])
        dP_droll = np.stack([R[0, 2] * py - R[0, 1] * pz, R[1, 2] * py - R[1, 1] * pz, R[2, 2] * py - R[2, 1] * pz])
        Xb = P[0] if corrected else (Xr + px * x)          # the reference's slip: px*(cp*cy + x) instead of px*cp*cy + x
        Ju = [fx * iz, 0 * iz, -fx * Xb * iz ** 2]
        Jv = [0 * iz, fy * iz, -fy * P[1] * iz ** 2]
        for dP in (dP_dyaw, dP_dpitch, dP_droll):
            Ju.append(fx * dP[0] * iz - fx * Xb * dP[2] * iz ** 2)
            Jv.append(fy * dP[1] * iz - fy * P[1] * dP[2] * iz ** 2)
        J = np.stack([gxs * Ju[j] + gys * Jv[j] for j in range(6)], axis=1)
    J[~inb] = 0.0
    res = np.where(inb, res, 0.0)
    return res, J


def optimize(pyr, K, cfg, init_state=None):
    """cfg: dict(num_levels, lam, max_iter, min_grad, min_depth, max_depth[, huber_delta]).
    huber_delta[L] > 0 (extension, not in the reference): IRLS weights of the Huber loss.
    Returns (state, iterations_per_level, trace list)."""
    state = np.zeros(6) if init_state is None else np.array(init_state, dtype=np.float64)
    g = np.zeros(6)
    iters = [0] * cfg["num_levels"]
    trace = []
    for level in range(cfg["num_levels"] - 1, -1, -1):
        it = 0
        while True:
            if cfg["max_iter"][level] > 0:
                if cfg.get("bilinear", False):
                    r, J = normal_equations_bilinear(pyr[level], level, K, state, cfg.get("min_depth", 0.3),
                                                     cfg.get("max_depth", 5.0), cfg.get("corrected", False))
                    g, Hm = J.T @ r, J.T @ J
                else:
                    g, Hm, r, J = normal_equations(pyr[level], level, K, state,
                                                   cfg.get("min_depth", 0.3), cfg.get("max_depth", 5.0))
                delta = cfg.get("huber_delta", [0.0] * cfg["num_levels"])[level]
                if delta > 0:
                    ar = np.abs(r)
                    wgt = np.where(ar <= delta, 1.0, delta / np.maximum(ar, 1e-300))
                    g = J.T @ (wgt * r)
                    Hm = J.T @ (J * wgt[:, None])
                state = state - cfg["lam"][level] * np.linalg.solve(Hm, g)
                trace.append(dict(level=level, iteration=it + 1, gradient=g.copy(),
                                  hessian=Hm.copy(), state=state.copy()))
            it += 1
            if it >= cfg["max_iter"][level]:
                break
            if np.linalg.norm(g) < cfg["min_grad"][level]:
                break
        iters[level] = it
    return state, iters, trace

"""ctypes loader for the CPU oracle (oracle/phovo_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see oracle/phovo_oracle.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libphovo_oracle.so")
MAX_LEVELS = 16


class Config(C.Structure):
    _fields_ = [
        ("num_levels", C.c_int),
        ("blur_filter_size", C.c_int * MAX_LEVELS),
        ("image_gradients_scaling_factor", C.c_double * MAX_LEVELS),
        ("lambda_optimization_step", C.c_double * MAX_LEVELS),
        ("max_num_iterations", C.c_int * MAX_LEVELS),
        ("min_gradient_norm", C.c_double * MAX_LEVELS),
        ("min_depth", C.c_double),
        ("max_depth", C.c_double),
    ]


class Level(C.Structure):
    _fields_ = [
        ("w", C.c_int), ("h", C.c_int),
        ("i0", C.POINTER(C.c_double)), ("d0", C.POINTER(C.c_double)),
        ("i1", C.POINTER(C.c_double)), ("gx1", C.POINTER(C.c_double)),
        ("gy1", C.POINTER(C.c_double)),
    ]


class TraceEntry(C.Structure):
    _fields_ = [
        ("level", C.c_int), ("iteration", C.c_int),
        ("gradient", C.c_double * 6), ("hessian", C.c_double * 36),
        ("state", C.c_double * 6),
        ("valid_pixels", C.c_int), ("reserved", C.c_int),
    ]


def build(force=False):
    """Compile the oracle with its Makefile (gcc).  Building the checker is not using it."""
    src = os.path.join(_HERE, "phovo_oracle.c")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= os.path.getmtime(src)
            and os.path.getmtime(_SO) >= os.path.getmtime(os.path.join(_HERE, "phovo_oracle.h"))):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        L.phovo_oracle_default_config.argtypes = [C.POINTER(Config)]
        L.phovo_oracle_eigen_pose.argtypes = [dp, dp]
        L.phovo_oracle_convert_intensity.argtypes = [C.c_void_p, C.c_int, dp]
        L.phovo_oracle_level_size.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.phovo_oracle_resize_level.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp]
        L.phovo_oracle_gaussian_blur_twice.argtypes = [dp, C.c_int, C.c_int, C.c_int]
        L.phovo_oracle_scharr.argtypes = [dp, C.c_int, C.c_int, C.c_double, dp, dp]
        L.phovo_oracle_compute_residuals_and_jacobians.argtypes = [
            C.POINTER(Level), C.c_int, dp, dp, C.c_double, C.c_double, dp, dp, dp]
        L.phovo_oracle_optimize.argtypes = [
            C.POINTER(Config), dp, C.POINTER(Level), dp, C.POINTER(C.c_int),
            C.POINTER(TraceEntry), C.c_int]
        L.phovo_oracle_optimize.restype = C.c_int
        L.phovo_oracle_optimize_huber.argtypes = [
            C.POINTER(Config), dp, C.POINTER(Level), dp, C.POINTER(C.c_int),
            C.POINTER(TraceEntry), C.c_int, dp]
        L.phovo_oracle_optimize_huber.restype = C.c_int
        L.phovo_oracle_optimize_ext.argtypes = [
            C.POINTER(Config), dp, C.POINTER(Level), dp, C.POINTER(C.c_int),
            C.POINTER(TraceEntry), C.c_int, dp, C.c_int, C.c_int]
        L.phovo_oracle_optimize_ext.restype = C.c_int
        L.phovo_oracle_unverified_hits.argtypes = [C.c_int]
        L.phovo_oracle_unverified_hits.restype = C.c_long
        L.phovo_oracle_unverified_reset.argtypes = []
        L.phovo_oracle_unverified_reset.restype = None
        L.phovo_oracle_warp_image.argtypes = [
            C.c_void_p, dp, C.c_int, C.c_int, dp, dp, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def make_config(num_levels=None, blur=None, grad_scale=None, lam=None, max_iter=None,
                min_grad=None, min_depth=0.3, max_depth=5.0):
    """Config with the constructor defaults (...Analytic.h:430-443) overridden per argument."""
    cfg = Config()
    lib().phovo_oracle_default_config(C.byref(cfg))
    if num_levels is not None:
        cfg.num_levels = int(num_levels)
    for name, vals in (("blur_filter_size", blur),
                       ("image_gradients_scaling_factor", grad_scale),
                       ("lambda_optimization_step", lam),
                       ("max_num_iterations", max_iter),
                       ("min_gradient_norm", min_grad)):
        if vals is not None:
            arr = getattr(cfg, name)
            for i, v in enumerate(list(vals)[:MAX_LEVELS]):
                arr[i] = v
    cfg.min_depth = float(min_depth)
    cfg.max_depth = float(max_depth)
    return cfg


def eigen_pose(state):
    s = _f64(state)
    rt = np.zeros(16)
    lib().phovo_oracle_eigen_pose(_dp(s), _dp(rt))
    return rt.reshape(4, 4)


def level_size(w, h, level):
    lw, lh = C.c_int(), C.c_int()
    lib().phovo_oracle_level_size(w, h, level, C.byref(lw), C.byref(lh))
    return lw.value, lh.value


def convert_intensity(gray_u8):
    g = np.ascontiguousarray(gray_u8, dtype=np.uint8)
    out = np.empty(g.shape, dtype=np.float64)
    lib().phovo_oracle_convert_intensity(g.ctypes.data, g.size, _dp(out))
    return out


def resize_level(img, level):
    img = _f64(img)
    h, w = img.shape
    lw, lh = level_size(w, h, level)
    out = np.empty((lh, lw), dtype=np.float64)
    lib().phovo_oracle_resize_level(_dp(img), w, h, level, _dp(out))
    return out


def gaussian_blur_twice(img, ksize):
    out = _f64(img).copy()
    h, w = out.shape
    lib().phovo_oracle_gaussian_blur_twice(_dp(out), w, h, int(ksize))
    return out


def scharr(img, scale):
    img = _f64(img)
    h, w = img.shape
    gx = np.empty_like(img)
    gy = np.empty_like(img)
    lib().phovo_oracle_scharr(_dp(img), w, h, float(scale), _dp(gx), _dp(gy))
    return gx, gy


def _intensity_pyramid(gray_u8, cfg):
    """BuildPyramid(intensity, applyBlur = true) (...Analytic.h:115-163).  Level 0 is `imgAux = img` (:136), a shallow
    cv::Mat alias of the converted image, and cv::GaussianBlur(imgAux, imgAux) (:146-147) writes through it: with
    blurFilterSize[0] > 0 the image every later cv::resize(img, ...) (:132) reads is the BLURRED level 0.  (All
    shipped configurations have blurFilterSize = 0; parity unpinned -- OpenCV is not in this image.)"""
    img = convert_intensity(gray_u8)
    pyr = []
    for level in range(cfg.num_levels):
        lv = resize_level(img, level) if level else img
        if cfg.blur_filter_size[level] > 0:
            lv = gaussian_blur_twice(lv, cfg.blur_filter_size[level])
            if level == 0:
                img = lv                                   # the in-place blur of the aliased level 0
        pyr.append(lv)
    return pyr


UNVERIFIED_BRANCHES = ("resize by 2: clipped block at an odd border", "resize by >= 4: tap clipped to the last row / column",
                       "GaussianBlur")


def unverified_hits(reset=False):
    """How often the pyramid branches tagged UNVERIFIED-vs-OpenCV in phovo_oracle.c (restated from recalled OpenCV behaviour
    and reached by no BASELINE shape) have run since the last reset: a tuple in the order of UNVERIFIED_BRANCHES."""
    L = lib()
    hits = tuple(int(L.phovo_oracle_unverified_hits(i)) for i in range(len(UNVERIFIED_BRANCHES)))
    if reset:
        L.phovo_oracle_unverified_reset()
    return hits


def build_source_pyramids(gray_u8, depth, cfg):
    """SetSourceFrame (...Analytic.h:466-476): intensity (blurred if configured) and depth pyramids
    (depth: applyBlur = false, :475, so the caller's image is never written)."""
    depth = _f64(depth)
    return _intensity_pyramid(gray_u8, cfg), [resize_level(depth, level) for level in range(cfg.num_levels)]


def build_target_pyramids(gray_u8, cfg):
    """SetTargetFrame (...Analytic.h:479-491): intensity pyramid and its Scharr gradients."""
    ipyr = _intensity_pyramid(gray_u8, cfg)
    gxp, gyp = [], []
    for level, lv in enumerate(ipyr):
        gx, gy = scharr(lv, cfg.image_gradients_scaling_factor[level])
        gxp.append(gx)
        gyp.append(gy)
    return ipyr, gxp, gyp


def _levels_array(i0p, d0p, i1p, gxp, gyp):
    n = len(i0p)
    keep = []
    arr = (Level * n)()
    for l in range(n):
        planes = [_f64(p[l]) for p in (i0p, d0p, i1p, gxp, gyp)]
        keep.append(planes)
        h, w = planes[0].shape
        arr[l].w, arr[l].h = w, h
        arr[l].i0, arr[l].d0, arr[l].i1, arr[l].gx1, arr[l].gy1 = [_dp(p) for p in planes]
    return arr, keep


def compute_residuals_and_jacobians(i0, d0, i1, gx1, gy1, level, K, state,
                                    min_depth=0.3, max_depth=5.0, want_warped=False):
    arr, keep = _levels_array([i0], [d0], [i1], [gx1], [gy1])
    h, w = keep[0][0].shape
    n = w * h
    r = np.zeros(n)
    J = np.zeros((6, n))            # column-major N x 6  ==  6 contiguous planes
    warped = np.zeros(n) if want_warped else None
    Kf, sf = _f64(K).reshape(9), _f64(state)
    lib().phovo_oracle_compute_residuals_and_jacobians(
        C.byref(arr[0]), int(level), _dp(Kf), _dp(sf), float(min_depth), float(max_depth),
        _dp(r), _dp(J), _dp(warped) if want_warped else None)
    return (r, J, warped) if want_warped else (r, J)


def optimize(cfg, K, i0p, d0p, i1p, gxp, gyp, init_state=None, want_trace=False,
             trace_capacity=8192, huber_delta=None, bilinear=False, corrected=False):
    """Optimize() on prebuilt pyramids.  Returns (state, iterations_per_level[, trace]).
    Extensions not in the reference: huber_delta (per level, Huber IRLS weights where > 0), bilinear
    (forward-additive alignment with bilinear sampling), corrected (true Jacobian, bilinear mode)."""
    arr, keep = _levels_array(i0p, d0p, i1p, gxp, gyp)
    state = np.zeros(6) if init_state is None else _f64(init_state).copy()
    iters = (C.c_int * MAX_LEVELS)()
    Kf = _f64(K).reshape(9)
    tr = (TraceEntry * trace_capacity)() if want_trace else None
    if bilinear or corrected:
        hd = np.zeros(MAX_LEVELS)
        if huber_delta is not None:
            hd[:len(huber_delta)] = huber_delta
        n = lib().phovo_oracle_optimize_ext(C.byref(cfg), _dp(Kf), arr, _dp(state), iters,
                                            tr, trace_capacity if want_trace else 0, _dp(hd),
                                            int(bool(bilinear)), int(bool(corrected)))
    elif huber_delta is None:
        n = lib().phovo_oracle_optimize(C.byref(cfg), _dp(Kf), arr, _dp(state), iters,
                                        tr, trace_capacity if want_trace else 0)
    else:
        hd = np.zeros(MAX_LEVELS)
        hd[:len(huber_delta)] = huber_delta
        n = lib().phovo_oracle_optimize_huber(C.byref(cfg), _dp(Kf), arr, _dp(state), iters,
                                              tr, trace_capacity if want_trace else 0, _dp(hd))
    its = [iters[l] for l in range(cfg.num_levels)]
    if not want_trace:
        return state, its
    trace = []
    for e in tr[:min(n, trace_capacity)]:
        trace.append(dict(level=e.level, iteration=e.iteration,
                          gradient=np.array(e.gradient[:]),
                          hessian=np.array(e.hessian[:]).reshape(6, 6),
                          state=np.array(e.state[:]), valid_pixels=int(e.valid_pixels)))
    return state, its, trace


def valid_pixels_per_level(trace, num_levels):
    """What the device path reports as phovo_pair_report.valid_pixels: the rows of J filled by the LAST executed
    iteration of each level (0 for a level that executed none)."""
    out = [0] * num_levels
    for e in trace:
        out[e["level"]] = e["valid_pixels"]
    return out


def align_frames(cfg, K, gray0, depth0, gray1, init_state=None, want_trace=False):
    """SetSourceFrame + SetTargetFrame + Optimize (apps/PhotoconsistencyFrameAlignment/...cpp:92-101)."""
    i0p, d0p = build_source_pyramids(gray0, depth0, cfg)
    i1p, gxp, gyp = build_target_pyramids(gray1, cfg)
    return optimize(cfg, K, i0p, d0p, i1p, gxp, gyp, init_state, want_trace)


def warp_image(gray_u8, depth, rt, K, level=0):
    g = np.ascontiguousarray(gray_u8, dtype=np.uint8)
    d = _f64(depth)
    h, w = g.shape
    out = np.zeros_like(g)
    rtf, Kf = _f64(rt).reshape(16), _f64(K).reshape(9)
    lib().phovo_oracle_warp_image(g.ctypes.data, _dp(d), w, h, _dp(rtf), _dp(Kf), int(level),
                                  out.ctypes.data)
    return out

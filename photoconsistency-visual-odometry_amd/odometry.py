"""Python host mirror of the reference's operator surface, over the C ABI.

`CPhotoconsistencyOdometryAnalytic` has the method names, argument meaning and call order of
phovo::Analytic::CPhotoconsistencyOdometryAnalytic<unsigned char,double>
(phovo/include/CPhotoconsistencyOdometryAnalytic.h:428-607); `AlignmentEngine` is the batched form
(one process per GPU, many independent frame pairs per launch).  Everything that computes runs in
libphovo_hip.so on the GPU; numpy only carries buffers across the boundary.
"""
import ctypes as C

import numpy as np

from . import native
from .native import check


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("intensity image must be 2-D (gray)")
    return a


def _f64img(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2:
        raise ValueError("depth image must be 2-D")
    return a


def warpImage(intensityImage, depthImage, Rt, intrinsicMatrix, level=0, device=0):
    """phovo::warpImage (phovo/include/CPhotoconsistencyOdometry.h:73-134) on the device: forward warp of the source
    intensities into the target view (depth > 0 gate, truncating cast, last raster writer wins, zeros elsewhere).
    Returns the warped u8 image instead of filling an output argument."""
    g, d = _u8(intensityImage), _f64img(depthImage)
    if g.shape != d.shape:
        raise ValueError("intensity and depth must have the same size")
    h, w = g.shape
    rt = np.ascontiguousarray(Rt, dtype=np.float64).reshape(16)
    k = np.ascontiguousarray(intrinsicMatrix, dtype=np.float64).reshape(9)
    out = np.empty((h, w), dtype=np.uint8)
    dp = C.POINTER(C.c_double)
    check(native.lib().phovo_warp_image(int(device), g.ctypes.data, w, d.ctypes.data, w * 8, w, h,
                                        rt.ctypes.data_as(dp), k.ctypes.data_as(dp), int(level),
                                        out.ctypes.data, w), "warpImage")
    return out


class CPhotoconsistencyOdometryAnalytic:
    """One frame pair at a time; 1:1 with the reference class."""

    def __init__(self, device=0):
        self._lib = native.lib()
        self._h = C.c_void_p()
        check(self._lib.phovo_odometry_create(int(device), C.byref(self._h)), "phovo_odometry_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.phovo_odometry_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- configuration --------------------------------------------------------------------
    def ReadConfigurationFile(self, fileName):
        check(self._lib.phovo_odometry_read_configuration_file(self._h, str(fileName).encode()),
              "ReadConfigurationFile")

    def SetConfiguration(self, cfg):
        check(self._lib.phovo_odometry_set_config(self._h, C.byref(cfg)), "SetConfiguration")

    def SetExtensions(self, ext):
        """Not in the reference: plane storage / Huber weights (native.make_extensions)."""
        check(self._lib.phovo_odometry_set_extensions(self._h, C.byref(ext)), "SetExtensions")

    def SetLatencyForms(self, on=True):
        """Not in the reference: Optimize() may take the forms that finish soonest for one pair (last bits may then differ
        from the same pair aligned in a batch); default off."""
        check(self._lib.phovo_odometry_set_latency_forms(self._h, 1 if on else 0), "SetLatencyForms")

    def SetMinDepth(self, minD):
        check(self._lib.phovo_odometry_set_min_depth(self._h, float(minD)), "SetMinDepth")

    def SetMaxDepth(self, maxD):
        check(self._lib.phovo_odometry_set_max_depth(self._h, float(maxD)), "SetMaxDepth")

    def SetIntrinsicMatrix(self, intrinsicMatrix):
        k = np.ascontiguousarray(intrinsicMatrix, dtype=np.float64).reshape(9)
        check(self._lib.phovo_odometry_set_intrinsic_matrix(self._h, k.ctypes.data_as(C.POINTER(C.c_double))),
              "SetIntrinsicMatrix")

    # -- frames ---------------------------------------------------------------------------
    def SetSourceFrame(self, intensityImage, depthImage):
        g, d = _u8(intensityImage), _f64img(depthImage)
        if g.shape != d.shape:
            raise ValueError("intensity and depth sizes differ")
        h, w = g.shape
        check(self._lib.phovo_odometry_set_source_frame(self._h, g.ctypes.data, g.strides[0],
                                                        d.ctypes.data, d.strides[0], w, h), "SetSourceFrame")

    def SetTargetFrame(self, intensityImage, depthImage=None):
        g = _u8(intensityImage)
        h, w = g.shape
        d = _f64img(depthImage) if depthImage is not None else None
        check(self._lib.phovo_odometry_set_target_frame(
            self._h, g.ctypes.data, g.strides[0],
            d.ctypes.data if d is not None else None, d.strides[0] if d is not None else 0, w, h),
            "SetTargetFrame")

    def SetInitialStateVector(self, initialStateVector):
        s = np.ascontiguousarray(initialStateVector, dtype=np.float64).reshape(6)
        check(self._lib.phovo_odometry_set_initial_state_vector(self._h, s.ctypes.data_as(C.POINTER(C.c_double))),
              "SetInitialStateVector")

    # -- optimisation ---------------------------------------------------------------------
    def Optimize(self):
        check(self._lib.phovo_odometry_optimize(self._h), "Optimize")

    def GetOptimalStateVector(self):
        s = np.zeros(6)
        check(self._lib.phovo_odometry_get_optimal_state_vector(self._h, s.ctypes.data_as(C.POINTER(C.c_double))),
              "GetOptimalStateVector")
        return s

    def GetOptimalRigidTransformationMatrix(self):
        rt = np.zeros(16)
        check(self._lib.phovo_odometry_get_optimal_rigid_transformation_matrix(
            self._h, rt.ctypes.data_as(C.POINTER(C.c_double))), "GetOptimalRigidTransformationMatrix")
        return rt.reshape(4, 4)

    def GetReport(self):
        rep = native.PairReport()
        check(self._lib.phovo_odometry_get_report(self._h, C.byref(rep)), "GetReport")
        return rep

    def LastOptimizeMilliseconds(self):
        ms = C.c_double()
        check(self._lib.phovo_odometry_last_optimize_ms(self._h, C.byref(ms)), "LastOptimizeMilliseconds")
        return ms.value


class AlignmentEngine:
    """Batched alignment: a pool of frames resident in HBM, pairs aligned one launch per level."""

    def __init__(self, device=0):
        self._lib = native.lib()
        self._h = C.c_void_p()
        check(self._lib.phovo_engine_create(int(device), C.byref(self._h)), "phovo_engine_create")
        self.n_frames = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.phovo_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_config(self, cfg):
        check(self._lib.phovo_engine_set_config(self._h, C.byref(cfg)), "phovo_engine_set_config")

    def read_configuration_file(self, path):
        self.set_extensions(native.read_extensions_file(path))
        self.set_config(native.read_config_file(path))

    def set_extensions(self, ext):
        check(self._lib.phovo_engine_set_extensions(self._h, C.byref(ext)), "phovo_engine_set_extensions")

    def get_extensions(self):
        ext = native.Extensions()
        check(self._lib.phovo_engine_get_extensions(self._h, C.byref(ext)), "phovo_engine_get_extensions")
        return ext

    def get_config(self):
        cfg = native.Config()
        check(self._lib.phovo_engine_get_config(self._h, C.byref(cfg)), "phovo_engine_get_config")
        return cfg

    def set_intrinsic_matrix(self, K):
        k = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        check(self._lib.phovo_engine_set_intrinsic_matrix(self._h, k.ctypes.data_as(C.POINTER(C.c_double))),
              "phovo_engine_set_intrinsic_matrix")

    def set_depth_range(self, min_depth, max_depth):
        check(self._lib.phovo_engine_set_depth_range(self._h, float(min_depth), float(max_depth)),
              "phovo_engine_set_depth_range")

    def set_build_all_levels(self, on):
        check(self._lib.phovo_engine_set_build_all_levels(self._h, int(bool(on))), "phovo_engine_set_build_all_levels")

    def set_wide_policy(self, policy):
        """0 automatic, 1 wide form wherever possible, -1 persistent form only."""
        check(self._lib.phovo_engine_set_wide_policy(self._h, int(policy)), "phovo_engine_set_wide_policy")

    def set_level_fusion(self, mode):
        """native.FUSION_AUTO (default): consecutive levels that fit the 512-thread scatter kernel are ONE launch when a
        gradient threshold makes their iteration counts data-dependent; FUSION_OFF: one launch per level; FUSION_SPLIT: one
        launch per level in the fused launch's geometry (bit-identical to AUTO)."""
        check(self._lib.phovo_engine_set_level_fusion(self._h, int(mode)), "phovo_engine_set_level_fusion")

    def set_batch_invariant(self, on=True):
        """Every batch, whatever its size, takes the same kernels and geometries: a pair's result does not depend on how
        many other pairs are aligned with it (what the sequence drivers set, so that a sequence cut into shards of any
        sizes gives bit-identical poses)."""
        check(self._lib.phovo_engine_set_batch_invariant(self._h, int(bool(on))), "phovo_engine_set_batch_invariant")

    def set_latency_forms(self, on=True):
        """A handful of pairs may take the forms that finish soonest also on levels of <= ~39 k pixels (last bits may then
        differ from the batch forms); default off: one arithmetic per pair on those levels."""
        check(self._lib.phovo_engine_set_latency_forms(self._h, 1 if on else 0), "phovo_engine_set_latency_forms")

    def set_slide_policy(self, policy):
        """0 automatic (sliding-window kernel on levels whose owner map exceeds LDS), -1 exact kernel only."""
        check(self._lib.phovo_engine_set_slide_policy(self._h, int(policy)), "phovo_engine_set_slide_policy")

    def level_uses_wide(self, level, n_pairs):
        return bool(self._lib.phovo_engine_level_uses_wide(self._h, int(level), int(n_pairs)))

    def reserve_frames(self, n_frames, width, height):
        check(self._lib.phovo_engine_reserve_frames(self._h, int(n_frames), int(width), int(height)),
              "phovo_engine_reserve_frames")
        self.n_frames = int(n_frames)

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        check(self._lib.phovo_engine_level_size(self._h, int(level), C.byref(w), C.byref(h)), "phovo_engine_level_size")
        return w.value, h.value

    def level_is_stored(self, level):
        return bool(self._lib.phovo_engine_level_is_stored(self._h, int(level)))

    def upload_frame(self, frame, gray, depth=None, roles=native.ROLE_BOTH):
        g = _u8(gray)
        d = _f64img(depth) if depth is not None else None
        check(self._lib.phovo_engine_upload_frame(
            self._h, int(frame), int(roles), g.ctypes.data, g.strides[0],
            d.ctypes.data if d is not None else None, d.strides[0] if d is not None else 0),
            "phovo_engine_upload_frame")

    def upload_frame_u16(self, frame, gray, depth_u16, depth_scale, roles=native.ROLE_BOTH):
        g = _u8(gray)
        d = np.ascontiguousarray(depth_u16, dtype=np.uint16)
        check(self._lib.phovo_engine_upload_frame_u16(
            self._h, int(frame), int(roles), g.ctypes.data, g.strides[0], d.ctypes.data, d.strides[0],
            float(depth_scale)), "phovo_engine_upload_frame_u16")

    def upload_frames(self, first_frame, gray, depth=None, depth_scale=None, roles=native.ROLE_BOTH):
        """Batched upload of gray [F,H,W] u8 with depth [F,H,W] fp64 (metres) or u16 (with depth_scale)."""
        g = np.ascontiguousarray(gray, dtype=np.uint8)
        if g.ndim != 3:
            raise ValueError("gray must be [frames, height, width]")
        if depth is None:
            check(self._lib.phovo_engine_upload_frames(self._h, int(first_frame), g.shape[0], int(roles), g.ctypes.data,
                                                       g.strides[1], g.strides[0], None, 0, 0), "phovo_engine_upload_frames")
        elif depth_scale is None:
            d = np.ascontiguousarray(depth, dtype=np.float64)
            check(self._lib.phovo_engine_upload_frames(self._h, int(first_frame), g.shape[0], int(roles), g.ctypes.data,
                                                       g.strides[1], g.strides[0], d.ctypes.data, d.strides[1], d.strides[0]),
                  "phovo_engine_upload_frames")
        else:
            d = np.ascontiguousarray(depth, dtype=np.uint16)
            check(self._lib.phovo_engine_upload_frames_u16(self._h, int(first_frame), g.shape[0], int(roles), g.ctypes.data,
                                                           g.strides[1], g.strides[0], d.ctypes.data, d.strides[1], d.strides[0],
                                                           float(depth_scale)), "phovo_engine_upload_frames_u16")

    def set_level_planes(self, frame, level, intensity=None, depth=None, grad_x=None, grad_y=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64)
                for a in (intensity, depth, grad_x, grad_y)]
        w, h = self.level_size(level)
        for a in arrs:
            if a is not None and a.size != w * h:
                raise ValueError("plane size does not match the level")
        ptrs = [a.ctypes.data if a is not None else None for a in arrs]
        check(self._lib.phovo_engine_set_level_planes(self._h, int(frame), int(level), *ptrs),
              "phovo_engine_set_level_planes")

    def get_level_planes(self, frame, level):
        w, h = self.level_size(level)
        outs = [np.empty((h, w), dtype=np.float64) for _ in range(4)]
        check(self._lib.phovo_engine_get_level_planes(self._h, int(frame), int(level),
                                                      *[o.ctypes.data for o in outs]),
              "phovo_engine_get_level_planes")
        return tuple(outs)            # intensity, depth, grad_x, grad_y

    @staticmethod
    def _pairs(src, tgt):
        s = np.ascontiguousarray(src, dtype=np.int32).reshape(-1)
        t = np.ascontiguousarray(tgt, dtype=np.int32).reshape(-1)
        if s.size != t.size:
            raise ValueError("source / target lists differ in length")
        return s, t

    def align_pairs(self, src, tgt, init_states=None, want_reports=False):
        s, t = self._pairs(src, tgt)
        n = s.size
        init = None if init_states is None else np.ascontiguousarray(init_states, dtype=np.float64).reshape(n, 6)
        out = np.zeros((n, 6))
        reps = (native.PairReport * max(n, 1))() if want_reports else None
        ip = C.POINTER(C.c_int)
        check(self._lib.phovo_engine_align_pairs(
            self._h, n, s.ctypes.data_as(ip), t.ctypes.data_as(ip),
            init.ctypes.data if init is not None else None, out.ctypes.data,
            C.cast(reps, C.c_void_p) if reps is not None else None), "phovo_engine_align_pairs")
        return (out, list(reps)[:n]) if want_reports else out

    def enqueue_align(self, src, tgt, init_states=None):
        s, t = self._pairs(src, tgt)
        n = s.size
        init = None if init_states is None else np.ascontiguousarray(init_states, dtype=np.float64).reshape(n, 6)
        ip = C.POINTER(C.c_int)
        check(self._lib.phovo_engine_enqueue_align(
            self._h, n, s.ctypes.data_as(ip), t.ctypes.data_as(ip),
            init.ctypes.data if init is not None else None), "phovo_engine_enqueue_align")
        return n

    def synchronize(self):
        check(self._lib.phovo_engine_synchronize(self._h), "phovo_engine_synchronize")

    # -- pipelining (phovo_hip.h): PHOVO_ENQUEUE_DEPTH = 2 enqueues in flight, each under a ticket ----------------
    def last_ticket(self):
        return int(self._lib.phovo_engine_last_ticket(self._h))

    def wait(self, ticket):
        check(self._lib.phovo_engine_wait(self._h, int(ticket)), "phovo_engine_wait")

    def fetch(self, ticket, n, want_reports=False):
        out = np.zeros((n, 6))
        reps = (native.PairReport * max(n, 1))() if want_reports else None
        check(self._lib.phovo_engine_fetch(
            self._h, int(ticket), int(n), out.ctypes.data, C.cast(reps, C.c_void_p) if reps is not None else None),
            "phovo_engine_fetch")
        return (out, list(reps)[:n]) if want_reports else out

    def device_states(self, ticket):
        p = C.c_void_p()
        check(self._lib.phovo_engine_device_states(self._h, int(ticket), C.byref(p)), "phovo_engine_device_states")
        return p.value

    def align_ms(self, ticket):
        total = C.c_double()
        per = (C.c_double * native.MAX_LEVELS)()
        check(self._lib.phovo_engine_align_ms(self._h, int(ticket), C.byref(total), per), "phovo_engine_align_ms")
        return total.value, list(per)

    def fetch_results(self, n, want_reports=False):
        out = np.zeros((n, 6))
        reps = (native.PairReport * max(n, 1))() if want_reports else None
        check(self._lib.phovo_engine_fetch_results(
            self._h, int(n), out.ctypes.data, C.cast(reps, C.c_void_p) if reps is not None else None),
            "phovo_engine_fetch_results")
        return (out, list(reps)[:n]) if want_reports else out

    def results_device_ptr(self):
        p = C.c_void_p()
        check(self._lib.phovo_engine_results_device_ptr(self._h, C.byref(p)), "phovo_engine_results_device_ptr")
        return p.value

    def last_align_ms(self):
        total = C.c_double()
        per = (C.c_double * native.MAX_LEVELS)()
        check(self._lib.phovo_engine_last_align_ms(self._h, C.byref(total), per), "phovo_engine_last_align_ms")
        return total.value, list(per)

    def level_launch_info(self, level):
        t, l, o, s = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(self._lib.phovo_engine_level_launch_info(self._h, int(level), C.byref(t), C.byref(l),
                                                       C.byref(o), C.byref(s)), "phovo_engine_level_launch_info")
        return dict(threads=t.value, lds_bytes=l.value, owner_in_lds=bool(o.value), source_in_lds=bool(s.value))

    def last_launches(self):
        """The kernel launches of the last enqueue, in order: dicts with levels (coarse to fine), kind, threads,
        lds_bytes, workgroups."""
        if not hasattr(self._lib, "phovo_engine_last_launches"):      # an older build under PHOVO_HIP_LIBRARY (tools/)
            return []
        cap = 4 * native.MAX_LEVELS
        recs = (native.LaunchRecord * cap)()
        n = C.c_int()
        check(self._lib.phovo_engine_last_launches(self._h, C.cast(recs, C.c_void_p), cap, C.byref(n)),
              "phovo_engine_last_launches")
        return [dict(levels=list(range(r.level_first, r.level_last - 1, -1)), kind=native.LAUNCH_KINDS[r.kind],
                     threads=r.threads, lds_bytes=r.lds_bytes, workgroups=r.workgroups) for r in recs[:min(n.value, cap)]]

"""ctypes binding of the C ABI in include/phovo_hip.h (libphovo_hip.so, built by csrc/Makefile).

There is no fallback: if the library is missing it is built with hipcc, and if that fails, or if no
HIP device is present when an engine is created, the error is raised to the caller.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library, always -- unless a tools/ entry point has opted in to a diagnostic build (tools/_variant.py sets
# PHOVO_TOOLS_LIBRARY_OPT_IN=tools before the package is imported and names the build in PHOVO_HIP_LIBRARY: csrc/Makefile
# `variant`).  PHOVO_HIP_LIBRARY alone is ignored, so nothing in the environment can swap the library under tests, bench.py
# or an application.
_OVERRIDE = os.environ.get("PHOVO_HIP_LIBRARY") if os.environ.get("PHOVO_TOOLS_LIBRARY_OPT_IN") == "tools" else None
_SO = _OVERRIDE or os.path.join(_HERE, "libphovo_hip.so")
_CSRC = os.path.join(_HERE, "csrc")
MAX_LEVELS = 16

ROLE_SOURCE, ROLE_TARGET, ROLE_BOTH = 1, 2, 3
PAIR_NONFINITE = 1
PAIR_WINDOW_FALLBACK = 2
PAIR_RANK_DEFICIENT = 4
# phovo_status
OK, E_INVALID_ARGUMENT, E_CONFIG, E_SHAPE, E_HIP, E_NOT_READY, E_IO, E_UNSUPPORTED = range(8)


class PhovoError(RuntimeError):
    def __init__(self, status, where, message):
        super().__init__(f"{where}: status {status} ({message})")
        self.status = status


class Config(C.Structure):
    _fields_ = [
        ("num_levels", C.c_int),
        ("blur_filter_size", C.c_int * MAX_LEVELS),
        ("image_gradients_scaling_factor", C.c_double * MAX_LEVELS),
        ("lambda_optimization_step", C.c_double * MAX_LEVELS),
        ("max_num_iterations", C.c_int * MAX_LEVELS),
        ("min_gradient_norm", C.c_double * MAX_LEVELS),
        ("visualize_iterations", C.c_int),
    ]


STORAGE_F64, STORAGE_F32, STORAGE_F16 = 0, 1, 2
SAMPLING_NEAREST_SCATTER, SAMPLING_BILINEAR = 0, 1


class Extensions(C.Structure):
    _fields_ = [
        ("plane_storage", C.c_int),
        ("sampling", C.c_int),
        ("jacobian_corrected", C.c_int),
        ("reserved", C.c_int),
        ("huber_delta", C.c_double * MAX_LEVELS),
    ]


class PairReport(C.Structure):
    _fields_ = [
        ("iterations", C.c_int * MAX_LEVELS),
        ("gradient_norm", C.c_double),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
        ("valid_pixels", C.c_int32 * MAX_LEVELS),
    ]


FUSION_AUTO, FUSION_OFF, FUSION_SPLIT = 0, -1, -2
LAUNCH_KINDS = ("persistent", "fused", "slide", "slide_fallback", "wide", "bilinear")


class LaunchRecord(C.Structure):
    _fields_ = [
        ("level_first", C.c_int), ("level_last", C.c_int), ("kind", C.c_int),
        ("threads", C.c_int), ("lds_bytes", C.c_int), ("workgroups", C.c_int),
    ]


# name -> (restype, argtypes); every symbol include/phovo_hip.h declares
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p
SYMBOLS = {
    "phovo_version": (C.c_char_p, []),
    "phovo_status_string": (C.c_char_p, [C.c_int]),
    "phovo_last_error": (C.c_char_p, []),
    "phovo_device_count": (C.c_int, []),
    "phovo_config_default": (C.c_int, [C.POINTER(Config)]),
    "phovo_config_read_file": (C.c_int, [C.c_char_p, C.POINTER(Config)]),
    "phovo_extensions_default": (C.c_int, [C.POINTER(Extensions)]),
    "phovo_extensions_read_file": (C.c_int, [C.c_char_p, C.POINTER(Extensions)]),
    "phovo_eigen_pose": (C.c_int, [_dp, _dp]),
    "phovo_trajectory_chain": (C.c_int, [C.c_int, _vp, _dp, _vp]),
    "phovo_trajectory_format_pose": (C.c_int, [C.c_double, _dp, C.c_char_p, C.c_size_t]),
    "phovo_warp_image": (C.c_int, [C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, C.c_int, _dp, _dp, C.c_int,
                                   _vp, C.c_size_t]),
    "phovo_odometry_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "phovo_odometry_destroy": (C.c_int, [_vp]),
    "phovo_odometry_read_configuration_file": (C.c_int, [_vp, C.c_char_p]),
    "phovo_odometry_set_config": (C.c_int, [_vp, C.POINTER(Config)]),
    "phovo_odometry_set_extensions": (C.c_int, [_vp, C.POINTER(Extensions)]),
    "phovo_odometry_set_latency_forms": (C.c_int, [_vp, C.c_int]),
    "phovo_odometry_set_min_depth": (C.c_int, [_vp, C.c_double]),
    "phovo_odometry_set_max_depth": (C.c_int, [_vp, C.c_double]),
    "phovo_odometry_set_intrinsic_matrix": (C.c_int, [_vp, _dp]),
    "phovo_odometry_set_source_frame": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, C.c_int]),
    "phovo_odometry_set_target_frame": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, C.c_int]),
    "phovo_odometry_set_initial_state_vector": (C.c_int, [_vp, _dp]),
    "phovo_odometry_optimize": (C.c_int, [_vp]),
    "phovo_odometry_get_optimal_state_vector": (C.c_int, [_vp, _dp]),
    "phovo_odometry_get_optimal_rigid_transformation_matrix": (C.c_int, [_vp, _dp]),
    "phovo_odometry_get_report": (C.c_int, [_vp, C.POINTER(PairReport)]),
    "phovo_odometry_last_optimize_ms": (C.c_int, [_vp, _dp]),
    "phovo_engine_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "phovo_engine_destroy": (C.c_int, [_vp]),
    "phovo_engine_set_config": (C.c_int, [_vp, C.POINTER(Config)]),
    "phovo_engine_get_config": (C.c_int, [_vp, C.POINTER(Config)]),
    "phovo_engine_set_extensions": (C.c_int, [_vp, C.POINTER(Extensions)]),
    "phovo_engine_get_extensions": (C.c_int, [_vp, C.POINTER(Extensions)]),
    "phovo_engine_set_intrinsic_matrix": (C.c_int, [_vp, _dp]),
    "phovo_engine_set_depth_range": (C.c_int, [_vp, C.c_double, C.c_double]),
    "phovo_engine_set_build_all_levels": (C.c_int, [_vp, C.c_int]),
    "phovo_engine_set_wide_policy": (C.c_int, [_vp, C.c_int]),
    "phovo_engine_set_level_fusion": (C.c_int, [_vp, C.c_int]),
    "phovo_engine_set_batch_invariant": (C.c_int, [_vp, C.c_int]),
    "phovo_engine_set_latency_forms": (C.c_int, [_vp, C.c_int]),
    "phovo_engine_set_slide_policy": (C.c_int, [_vp, C.c_int]),
    "phovo_engine_level_uses_wide": (C.c_int, [_vp, C.c_int, C.c_int]),
    "phovo_host_register": (C.c_int, [_vp, C.c_size_t]),
    "phovo_host_unregister": (C.c_int, [_vp]),
    "phovo_engine_reserve_frames": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "phovo_engine_level_size": (C.c_int, [_vp, C.c_int, _ip, _ip]),
    "phovo_engine_level_is_stored": (C.c_int, [_vp, C.c_int]),
    "phovo_engine_upload_frame": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t]),
    "phovo_engine_upload_frame_u16": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, C.c_double]),
    "phovo_engine_upload_frames": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_size_t]),
    "phovo_engine_upload_frames_u16": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_size_t, C.c_double]),
    "phovo_engine_set_level_planes": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "phovo_engine_get_level_planes": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "phovo_engine_align_pairs": (C.c_int, [_vp, C.c_int, _ip, _ip, _vp, _vp, _vp]),
    "phovo_engine_enqueue_align": (C.c_int, [_vp, C.c_int, _ip, _ip, _vp]),
    "phovo_engine_synchronize": (C.c_int, [_vp]),
    "phovo_engine_fetch_results": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "phovo_engine_results_device_ptr": (C.c_int, [_vp, C.POINTER(_vp)]),
    "phovo_engine_last_align_ms": (C.c_int, [_vp, _dp, _dp]),
    "phovo_engine_level_launch_info": (C.c_int, [_vp, C.c_int, _ip, _ip, _ip, _ip]),
    "phovo_engine_last_launches": (C.c_int, [_vp, _vp, C.c_int, _ip]),
    "phovo_engine_last_ticket": (C.c_int, [_vp]),
    "phovo_engine_wait": (C.c_int, [_vp, C.c_int]),
    "phovo_engine_fetch": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "phovo_engine_device_states": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "phovo_engine_align_ms": (C.c_int, [_vp, C.c_int, _dp, _dp]),
}


def library_path():
    return _SO


def source_sha256():
    """sha256 over the sources libphovo_hip.so is built from (csrc/*.hip, *.hpp, *.cpp, the Makefile, include/phovo_hip.h), in
    name order: the stamp a rocprofv3 summary under profiles/ carries (tools/profile_round.sh) and bench.py compares with the
    tree it runs from.  (The library file itself is not reproducible bit for bit across build directories.)"""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC))
             if f.endswith((".hip", ".hpp", ".cpp")) or f == "Makefile"]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "phovo_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False):
    """Compile libphovo_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-s", "-C", _CSRC]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    if not os.path.exists(_SO):
        raise RuntimeError("libphovo_hip.so was not produced by csrc/Makefile")
    return _SO


_lib = None


def lib():
    """The loaded library.  Builds it if it is missing; raises if that is impossible."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        for name, (res, args) in SYMBOLS.items():
            if _OVERRIDE and not hasattr(L, name):
                continue                     # a diagnostic / older build named explicitly (tools/ A-B runs): bind what it has
            fn = getattr(L, name)            # AttributeError if the ABI and the header diverge
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, where):
    if status != 0:
        L = lib()
        msg = L.phovo_last_error().decode() or L.phovo_status_string(status).decode()
        raise PhovoError(status, where, msg)


def make_config(num_levels=None, blur=None, grad_scale=None, lam=None, max_iter=None,
                min_grad=None, visualize=0):
    """Constructor defaults (...Analytic.h:430-443) overridden per argument."""
    cfg = Config()
    check(lib().phovo_config_default(C.byref(cfg)), "phovo_config_default")
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
    cfg.visualize_iterations = int(visualize)
    return cfg


def make_extensions(plane_storage=STORAGE_F64, huber_delta=None, sampling=SAMPLING_NEAREST_SCATTER,
                    jacobian_corrected=False):
    ext = Extensions()
    check(lib().phovo_extensions_default(C.byref(ext)), "phovo_extensions_default")
    ext.plane_storage = int(plane_storage)
    ext.sampling = int(sampling)
    ext.jacobian_corrected = int(bool(jacobian_corrected))
    if huber_delta is not None:
        for i, v in enumerate(list(huber_delta)[:MAX_LEVELS]):
            ext.huber_delta[i] = float(v)
    return ext


def read_extensions_file(path):
    ext = Extensions()
    check(lib().phovo_extensions_read_file(os.fsencode(path), C.byref(ext)), "phovo_extensions_read_file")
    return ext


def read_config_file(path):
    cfg = Config()
    check(lib().phovo_config_read_file(os.fsencode(path), C.byref(cfg)), "phovo_config_read_file")
    return cfg

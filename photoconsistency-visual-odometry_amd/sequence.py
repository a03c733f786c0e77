"""One RGB-D sequence, its frame pairs sharded across the GPUs of a node, one trajectory file out.

The multi-GPU form of the reference's VisualOdometry loop
(apps/PhotoconsistencyVisualOdometry/PhotoconsistencyVisualOdometry.cpp:175,187-188,208-243): every pair starts from the
zero state and rebuilds its pyramids (:175,222-224), so pair t depends on frames t-1 and t only; the one sequential
step is `pose *= Rt^-1` (:233-234).  Hence, with one process per GPU:

  * rank r aligns the contiguous pair range shard_range(F-1, world, r) and decodes/uploads ONLY the frames
    frames_needed(...) of that range (one frame of overlap with its neighbour) -- no data-path collective;
  * ONE all_gather of the 6-vector states (RCCL: backend "nccl"; "gloo" for rehearsals and tests);
  * rank 0 chains the poses and writes `timestamp tx ty tz qx qy qz qw` lines through the same C-ABI functions the
    C++ app uses (phovo_trajectory_chain / phovo_trajectory_format_pose), so the file is byte-identical to
    `PhotoconsistencyVisualOdometry <cfg> <dir> <out> --batch` on one GPU.

    python -m torch.distributed.run --nproc-per-node N apps/PhotoconsistencyVisualOdometrySharded.py <cfg.yml> <dir> <out>
    python apps/PhotoconsistencyVisualOdometrySharded.py <cfg.yml> <dir> <out> --ranks N      (starts the ranks itself)

Dataset conventions are the app's (and the reference's): rgb.txt / depth.txt read in lock step, '#' lines skipped,
paths relative to the list file, intrinsics (517.3, 516.5, 318.6, 255.3), depth = u16 / 5000 (:163,170-173).
There is no CPU path: a rank without a HIP device fails in phovo_engine_create.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

from . import distributed, native, odometry

_HERE = os.path.dirname(os.path.abspath(__file__))
_APPS = os.path.join(os.path.dirname(_HERE), "apps")
_IO_SO = os.path.join(_APPS, "bin", "libphovo_io.so")

K_TUM = np.array([[517.3, 0.0, 318.6], [0.0, 516.5, 255.3], [0.0, 0.0, 1.0]])     # ...VisualOdometry.cpp:170-173
DEPTH_SCALE = 1.0 / 5000.0                                                         # :163

_io = None


def io_lib():
    """apps/io/png_io behind its C ABI: the decoder the C++ apps use (cv::imread(path, 0) / (path, -1) semantics)."""
    global _io
    if _io is None:
        if not os.path.exists(_IO_SO):
            subprocess.check_call(["make", "-s", "-C", _APPS, _IO_SO])
        L = C.CDLL(_IO_SO)
        ip = C.POINTER(C.c_int)
        L.phovo_io_read_gray8.argtypes = [C.c_char_p, ip, ip, C.POINTER(C.POINTER(C.c_uint8)), C.c_char_p, C.c_size_t]
        L.phovo_io_read_unchanged16.argtypes = [C.c_char_p, ip, ip, C.POINTER(C.POINTER(C.c_uint16)), C.c_char_p, C.c_size_t]
        L.phovo_io_free.argtypes = [C.c_void_p]
        L.phovo_io_free.restype = None
        _io = L
    return _io


def _read_png(fn, ptr_type, dtype, path):
    w, h, px = C.c_int(), C.c_int(), C.POINTER(ptr_type)()
    err = C.create_string_buffer(512)
    if fn(os.fsencode(path), C.byref(w), C.byref(h), C.byref(px), err, len(err)) != 0:
        raise IOError(err.value.decode() or f"cannot read {path}")
    try:
        return np.ctypeslib.as_array(px, shape=(h.value, w.value)).astype(dtype, copy=True)
    finally:
        io_lib().phovo_io_free(px)


def read_gray8(path):
    return _read_png(io_lib().phovo_io_read_gray8, C.c_uint8, np.uint8, path)


def read_depth16(path):
    return _read_png(io_lib().phovo_io_read_unchanged16, C.c_uint16, np.uint16, path)


def read_list(list_file):
    """[(timestamp, path)]: '#' and empty lines skipped, paths relative to the list file
    (phovo/include/CCameraRecord.h:74-108)."""
    base = os.path.dirname(os.path.abspath(list_file))
    out = []
    with open(list_file) as f:
        for line in f:
            if not line.strip() or line.startswith("#"):
                continue
            parts = line.split()
            if len(parts) < 2:
                continue
            try:
                ts = float(parts[0])
            except ValueError:
                continue
            out.append((ts, os.path.join(base, parts[1])))
    return out


def read_sequence_lists(dataset_dir):
    """rgb.txt and depth.txt in lock step -- line n with line n, no timestamp association, stopping at the shorter
    list (phovo/include/CMultiSensorDataSource.h:74-91)."""
    rgb = read_list(os.path.join(dataset_dir, "rgb.txt"))
    depth = read_list(os.path.join(dataset_dir, "depth.txt"))
    n = min(len(rgb), len(depth))
    return rgb[:n], depth[:n]


def chain_and_format(states, timestamps):
    """The trajectory file's text from per-pair states: pose *= Rt^-1 and one line per pair (:233-243), through the
    C ABI (the same code the C++ app runs)."""
    L = native.lib()
    states = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 6)
    n = states.shape[0]
    pose = np.eye(4).reshape(16).copy()
    poses = np.zeros((max(n, 1), 16))
    dp = C.POINTER(C.c_double)
    native.check(L.phovo_trajectory_chain(n, states.ctypes.data, pose.ctypes.data_as(dp), poses.ctypes.data),
                 "phovo_trajectory_chain")
    lines = ["# estimated trajectory", "# timestamp tx ty tz qx qy qz qw"]                # :187-188
    buf = C.create_string_buffer(256)
    for p in range(n):
        native.check(L.phovo_trajectory_format_pose(float(timestamps[p]), poses[p].ctypes.data_as(dp), buf, len(buf)),
                     "phovo_trajectory_format_pose")
        lines.append(buf.value.decode())
    return "\n".join(lines) + "\n"


def align_shard(config_file, rgb, depth, pair_start, pair_stop, device_index, log=None, want_reports=False):
    """Decode, upload and align the pairs [pair_start, pair_stop) of the sequence on one GPU.  Pair t is
    (frame t -> frame t+1).  Returns ([p, 6] states, frames decoded) -- with want_reports ([p, 6] states, frames
    decoded, per-pair reports)."""
    f0, f1 = distributed.frames_needed(pair_start, pair_stop)
    n_pairs = pair_stop - pair_start
    if n_pairs <= 0:
        return np.zeros((0, 6)), 0
    # decoding dominates a rank's wall time (6-20 ms per frame on one core); frames are independent and the decoder runs
    # outside the GIL (ctypes), so all of the rank's host threads decode at once
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(32, (os.cpu_count() or 1) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    with ThreadPoolExecutor(workers) as pool:
        gray = list(pool.map(lambda f: read_gray8(rgb[f][1]), range(f0, f1)))
        d16 = list(pool.map(lambda f: read_depth16(depth[f][1]), range(f0, f1)))
    h, w = gray[0].shape
    for f, (g, d) in enumerate(zip(gray, d16)):
        if g.shape != (h, w) or d.shape != (h, w):
            raise ValueError(f"frame {f0 + f} has a different size")
    if log:
        log(f"decoded frames [{f0}, {f1}) for pairs [{pair_start}, {pair_stop})")
    with odometry.AlignmentEngine(device_index) as eng:
        eng.read_configuration_file(config_file)
        eng.set_batch_invariant(True)        # a pair's pose must not depend on the size of the shard it falls into
        eng.set_intrinsic_matrix(K_TUM)
        eng.reserve_frames(f1 - f0, w, h)
        eng.upload_frames(0, np.stack(gray), np.stack(d16), depth_scale=DEPTH_SCALE)
        local = list(range(n_pairs))
        if want_reports:
            states, reports = eng.align_pairs(local, [i + 1 for i in local], want_reports=True)
            return states, f1 - f0, list(reports)
        states = eng.align_pairs(local, [i + 1 for i in local])
    return states, f1 - f0


def run(config_file, dataset_dir, trajectory_path, backend=None, log=None):
    """The body of one rank (or of the only process when no launcher set RANK/WORLD_SIZE)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    log = log or (lambda msg: print(msg, file=sys.stderr, flush=True))
    device = None
    dist = None
    if world > 1:
        # whatever launcher started this rank: the host driver only does dmabuf IPC (RCCL needs it), set before torch loads
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        import torch
        import torch.distributed as dist
        backend = backend or os.environ.get("PHOVO_SEQUENCE_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            device = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", device_id=device)
        else:                                            # rehearsal on fewer GPUs than ranks: ranks share the cards
            device = torch.device("cpu")
            dist.init_process_group(backend)
            local_rank = local_rank % max(native.lib().phovo_device_count(), 1)
        log(f"sequence: rank {rank}/{world} joined the {backend} group")
    try:
        rgb, depth = read_sequence_lists(dataset_dir)
        n_frames = len(rgb)
        n_pairs = max(n_frames - 1, 0)
        a, b = distributed.shard_range(n_pairs, world, rank)
        states, decoded = align_shard(config_file, rgb, depth, a, b, local_rank,
                                      log=lambda m: log(f"sequence: rank {rank}/{world} {m}"))
        log(f"sequence: rank {rank}/{world} aligned pairs [{a}, {b}) from {decoded} decoded frames of {n_frames}")
        if world > 1:
            states = distributed.gather_states(states, n_pairs, device=device)        # the ONE collective
        if rank == 0:
            out_dir = os.path.dirname(os.path.abspath(trajectory_path))
            os.makedirs(out_dir, exist_ok=True)
            with open(trajectory_path, "w") as f:
                f.write(chain_and_format(states, [rgb[t + 1][0] for t in range(n_pairs)]))   # CURRENT rgb stamp  :240
        if world > 1:
            dist.barrier()
    finally:
        if world > 1 and dist.is_initialized():
            dist.destroy_process_group()
    return 0

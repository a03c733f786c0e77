#!/usr/bin/env python3
"""bench.py -- frame-pair alignments per second of the analytic Gauss-Newton path on MI355X.

One "step" = Optimize() for one batch of independent 640x480 frame pairs whose pyramids are already
resident in HBM (the reference times exactly Optimize(), pyramids excluded:
apps/PhotoconsistencyFrameAlignment/PhotoconsistencyFrameAlignment.cpp:99-102).

Workload (BASELINE.json configs[1]): config_files/config_4_level_optimization_analytic.yml on synthetic
640x480 RGB-D.  The timed region runs it in FIXED-ITERATION mode (min_gradient_norm = 0, so every pair
executes exactly max_num_iterations = 50 + 20 iterations: deterministic work, the mode SURVEY.md section 8d
prescribes for the roofline figure).  The shipped thresholds (data-dependent early stop) are measured too and
reported under "reference_termination" -- that number is higher, it is not `value`.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); every rank aligns its own contiguous
shard of the pair list (weak scaling: `--pairs` per GPU) and ONE all_gather per step moves the 6-vector
results to every rank.  value = pairs of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The package (and with it libphovo_hip.so / the HIP runtime) is imported in main(), AFTER the self-launch decision:
# the parent of a `python bench.py --gpus N` run must never touch the GPU (launch_ranks below).
distributed = native = odometry = se3 = synthetic = None

PARITY_BAR = 1e-9             # ||log(T_gpu^-1 T_cpu)|| of the in-line self-check, the bar the GPU tests hold
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
# headline workload (BASELINE.json configs[1] / configs[3]); --workload cfg5 switches to configs[4]'s shape
WORKLOADS = {
    # BASELINE.json configs[0]'s configuration, batched: level 0 only (307 200 pixels: the sliding-window kernel).  The yml's
    # own max_num_iterations is 5000 with a threshold of 300; the fixed-iteration figure takes 5 iterations per pair.
    "cfg1": dict(yml="config_only_level_0_analytic.yml", size=(640, 480), pairs=512, max_iterations="5",
                 metric="frame-pair alignments/sec (640x480, level 0 only, 5 iterations; not the headline metric)"),
    "cfg2": dict(yml="config_4_level_optimization_analytic.yml", size=(640, 480),
                 metric="frame-pair alignments/sec (640x480, 4-level)"),
    "cfg3": dict(yml="config_5_level_optimization_analytic.yml", size=(640, 480),
                 metric="frame-pair alignments/sec (640x480, 5-level: the VisualOdometry app's configuration; not the headline metric)"),
    "cfg5": dict(yml="config_6_level_optimization_analytic.yml", size=(1280, 960),
                 metric="frame-pair alignments/sec (1280x960, 6-level; not the headline metric)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=None,
                    help="frame pairs per GPU per step (a level launch has ~0.13 ms of fixed cost: 2048 pairs run at "
                         "241 k alignments/s, 8192 at 253 k, 16384 at 254 k)")
    ap.add_argument("--distinct", type=int, default=1024,
                    help="distinct synthetic pairs generated per GPU (consecutive frames of one rendered sequence; the batch "
                         "repeats them, every repeat with its own copy of the planes in HBM).  The fixed-iteration headline "
                         "does not depend on it (profiles/r04_runs/distinct_pairs.txt: 32 ... 4096); with the shipped "
                         "thresholds the iteration counts are data, and 32 pairs are not a sample of them")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="bound of the CPU-oracle baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads of the all-cores CPU baseline (a one-GPU box's CPU share); 1 = skip it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-termination", action="store_true")
    # extensions that are NOT in the reference; the default run (driver's BENCH/SCALE) never sets them
    ap.add_argument("--storage", choices=["f64", "f32", "f16"], default="f64",
                    help="plane storage (f64 = reference-exact; arithmetic is fp64 in every mode)")
    ap.add_argument("--huber", type=float, default=0.0, help="Huber delta on every level (0 = off)")
    ap.add_argument("--bilinear", action="store_true", help="bilinear forward-additive sampling + corrected Jacobian")
    ap.add_argument("--max-iterations", default=None,
                    help="diagnostic, never the default: comma list overriding the yml's max_num_iterations, level 0 first "
                         "(e.g. 0,0,1,1 = every plane streamed exactly once: the HBM-only rate of the level kernels)")
    ap.add_argument("--thresholds", choices=["fixed", "shipped"], default="fixed",
                    help="fixed (default, the headline): min_gradient_norm = 0, every pair runs max_num_iterations; shipped: "
                         "diagnostic, the yml's own min_gradient_norm in the TIMED region (data-dependent early stop) -- what "
                         "tools/profile_round.sh profiles for the shipped configuration; `value` is then not the headline")
    ap.add_argument("--pipeline", choices=["auto", "on", "off"], default="auto",
                    help="two enqueues in flight (phovo_hip.h, Pipelining): step k + 1 is enqueued before step k is waited for. "
                         "auto = on where a gradient threshold makes a batch end with a few long pairs (--thresholds shipped and "
                         "the reference_termination leg), off for the fixed-iteration headline, whose per-kernel event spans "
                         "must not overlap")
    ap.add_argument("--scene", choices=["plane", "layered"], default="plane",
                    help="synthetic scene: the slanted textured plane (default) or the layered desk-like scene with depth "
                         "discontinuities, invalid regions and depth noise (synthetic.py)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2",
                    help="cfg1 = BASELINE.json configs[0]'s configuration batched (level 0 only); "
                         "cfg2 = the headline 640x480 4-level workload; cfg3 = BASELINE.json configs[2]'s configuration "
                         "(640x480, config_5_level); cfg5 = configs[4]'s shape (1280x960, config_6_level; combine with "
                         "--storage f16 --huber 0.05)")
    return ap.parse_args()


def algorithmic_bytes(level_sizes, iterations):
    """per GN iteration and pair, 5 fp64 planes of N_L pixels are read once."""
    return sum(5.0 * 8.0 * n * it for n, it in zip(level_sizes, iterations))


KERNEL_NAMES = {"persistent": "gn_level_kernel", "fused": "gn_fused_kernel", "slide": "gn_level_kernel_slide",
                "slide_fallback": "gn_level_kernel", "wide": "k_wide_pass1 + k_wide_pass2", "bilinear": "gn_level_kernel_bilinear"}


def launch_rows(launches, per_level_ms, steps, iters, level_sizes, plane_bytes, max_iter, n_pairs, storage_types):
    """One row per kernel launch of a step, from the engine's own launch records (phovo_engine_last_launches): the levels
    it covers, its kernel, its share of the algorithmic bytes and its average duration (the HIP-event span the engine
    reports at the launch's coarsest level; the exact launch behind a sliding-window launch is inside that span)."""
    rows = []
    for rec in launches:
        if rec["kind"] == "slide_fallback":
            rows[-1]["followed_by"] = f"gn_level_kernel<{rec['threads']}, ...> on the pairs that left the window"
            continue
        lv = [l for l in rec["levels"] if max_iter[l] > 0]
        pair_it = {str(l): float(iters[:, l].sum()) for l in lv}
        nbytes = sum(plane_bytes * level_sizes[l] * pair_it[str(l)] for l in lv)
        ms = per_level_ms[lv[0]] / steps
        name = KERNEL_NAMES[rec["kind"]]
        if rec["kind"] == "bilinear" and not storage_types.startswith("__half"):
            name += "_dma"              # fp64 / fp32 planes: the taps-through-LDS form (gn_bilinear_kernel.hip)
        rows.append(dict(levels=lv, pixels=[level_sizes[l] for l in lv], kind=rec["kind"],
                         kernel=f"{name}<{rec['threads']}, ... {storage_types}>",
                         threads=rec["threads"], lds_bytes=rec["lds_bytes"], workgroups=rec["workgroups"],
                         avg_launch_ms=ms, pair_iterations_per_level=pair_it,
                         iterations_per_pair={k: v / n_pairs for k, v in pair_it.items()},
                         algorithmic_bytes=nbytes, achieved_GBs=nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0))
    return rows


def library_sha256():
    """sha256 of the libphovo_hip.so this process loaded: a stored profile is this build's only if it carries the same one."""
    import hashlib
    h = hashlib.sha256()
    with open(native.library_path(), "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def profile_context(kernel, pairs):
    """What the committed rocprofv3 counter passes (profiles/, newest round first) hold for `kernel` (the name form of
    launch_rows) at `pairs` pairs per launch: HBM-side bytes per launch, vector-unit busy fraction and vector instructions
    per launch, and the rate at which the same kernel streams planes it reads exactly once (pure HBM).  These are NOT
    measurements of the running process (PMC passes cannot run inside it): every profile JSON carries the sha256 of the
    library it was collected on (tools/parse_rocprof.py), and `library_matches` says whether that is the library loaded now."""
    import csv
    import re
    out = {}
    fam, threads, types = re.match(r"([\w +]+)<(\d+), \.\.\. (.*)>", kernel).groups()

    def same(name):
        m = re.search(r"(\w+)<(\d+),", name)
        return bool(m) and m.group(1) == fam and int(m.group(2)) == int(threads) and types in name

    prof = os.path.join(ROOT, "profiles")
    for tag in ("r05", "r04", "r03"):
        for f in sorted(os.listdir(prof)):
            if not (f.startswith(tag + "_") and f.endswith("_pmc_traffic.json")) or "stream_once" in f:
                continue
            try:
                # (family, thread count and storage can name two instantiations -- the 1024-thread level kernel with the
                # owner map in LDS or in HBM: the one with the larger figure is the launch, the other ran behind a
                # sliding-window launch with few pairs or none)
                doc = json.load(open(os.path.join(prof, f)))
                for kd in sorted(doc.get("kernels", []), key=lambda k: -k.get("hbm_bytes_per_launch", 0.0)):
                    if same(kd.get("kernel", "")) and kd.get("pairs") == pairs and "traffic" not in out:
                        out["traffic"] = kd["hbm_bytes_per_launch"]
                        out["traffic_source"] = f"profiles/{f} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, calibrated)"
                        out["profile_library_sha256"] = doc.get("library_sha256")
                        out["profile_source_sha256"] = doc.get("source_sha256")
                        out["profile_commit"] = doc.get("commit")
                sq = os.path.join(prof, f.replace("_pmc_traffic.json", "_pmc_sq.json"))
                if os.path.exists(sq):
                    for kd in sorted(json.load(open(sq)).get("kernels", []), key=lambda k: -k.get("mean_duration_ns", 0.0)):
                        if same(kd.get("kernel", "")) and kd.get("pairs") == pairs and "valu" not in out:
                            share = kd["fraction_of_wave_cycles"]["issuing VALU (SQ_ACTIVE_INST_VALU)"]
                            waves = kd.get("waves_per_simd", 4)
                            out["valu"] = dict(busy=share * waves, waves_per_simd=waves,
                                               instructions_per_launch=kd["counters"].get("SQ_INSTS_VALU"),
                                               effective_clock_GHz=kd.get("effective_clock_GHz"),
                                               source=f"profiles/{os.path.basename(sq)} (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x waves per SIMD)")
            except Exception:
                pass
        so = os.path.join(prof, tag + "_stream_once_pmc_traffic.json")
        st = os.path.join(prof, tag + "_stream_once_kernel_stats.csv")
        if "hbm_stream_measured" not in out and os.path.exists(so) and os.path.exists(st):
            try:
                ns = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(st))}
                for kd in json.load(open(so)).get("kernels", []):
                    if same(kd.get("kernel", "")) and kd["kernel"] in ns:
                        out["hbm_stream_measured"] = dict(
                            GBs=kd["hbm_bytes_per_launch"] / ns[kd["kernel"]],
                            source=f"profiles/{tag}_stream_once_*: this kernel with one iteration per pair (every plane read "
                                   "exactly once per launch: pure HBM); the timed run's traffic above that rate is "
                                   "Infinity-Cache hits (DESIGN.md 5.2)")
            except Exception:
                pass
    return out


ZERO_COPY = {"ok": None}      # None = not probed yet; decided once by probe_zero_copy() outside the timed region


def probe_zero_copy(eng, n_local, device):
    """May the all_gather start from the engine's device buffer?  torch brings its own HIP runtime and the library
    uses the system's, so this is checked once, after the warm-up steps: the torch view of the device pointer must
    read back exactly what the library's own copy returns; anything else (or any exception) selects the host path."""
    ZERO_COPY["ok"] = False
    if device is None or device.type != "cuda":
        return
    try:
        host = eng.fetch_results(n_local)
        view = distributed.device_states_tensor(eng.results_device_ptr(), n_local, device)
        ZERO_COPY["ok"] = bool(view.device == device and np.array_equal(view.cpu().numpy(), host))
    except Exception:
        ZERO_COPY["ok"] = False


def run_steps(eng, src, tgt, steps, use_dist, device, n_global, pipelined=False):
    """Runs `steps` steps; returns (wall seconds of this rank, per-level kernel ms summed over the steps).

    pipelined: step k + 1 is enqueued before step k is waited for (phovo_hip.h, "Pipelining": two enqueues in flight, each on
    its own stream), so the next batch's kernels fill the CUs that the long pairs at the end of a batch leave idle; the
    results of every step are still gathered, one step behind.  The per-enqueue event spans then overlap and are not summed
    into anything: with pipelined=True only the wall time is meaningful."""
    per_level = np.zeros(native.MAX_LEVELS + 1)          # [level spans ..., whole enqueue]
    STEP_WALL.clear()                                    # wall time of every step of this call (min / median / max in the line)

    def finish(ticket):
        eng.wait(ticket)
        if use_dist:
            if ZERO_COPY["ok"]:               # RCCL: start the collective from the engine's own device buffer
                local = distributed.device_states_tensor(eng.device_states(ticket), len(src), device)
            else:
                local = eng.fetch(ticket, len(src))
            distributed.gather_states(local, n_global, device=device)
        total_ms, lv = eng.align_ms(ticket)
        per_level[:native.MAX_LEVELS] += np.array(lv)
        per_level[native.MAX_LEVELS] += total_ms      # first launch to last

    t0 = time.perf_counter()
    pending = None
    t_step = t0
    for _ in range(steps):
        eng.enqueue_align(src, tgt)
        ticket = eng.last_ticket()
        if pipelined:
            if pending is not None:
                finish(pending)
            pending = ticket
        else:
            finish(ticket)
        t_now = time.perf_counter()
        STEP_WALL.append(t_now - t_step)
        t_step = t_now
    if pending is not None:
        finish(pending)
    return time.perf_counter() - t0, per_level


STEP_WALL = []


def step_wall_summary():
    """min / median / max wall time of the steps of the last run_steps() call (one enqueue at a time: a step is enqueue +
    wait + gather; pipelined: the interval between two enqueues), so that a reader sees whether the timed region was steady."""
    if not STEP_WALL:
        return None
    a = np.array(STEP_WALL) * 1e3
    return dict(min=float(a.min()), median=float(np.median(a)), max=float(a.max()), steps=int(a.size))


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves.

    This process has not imported torch, the package or anything else that could initialise the GPU; it starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD (never an exec: a
    process that has touched the GPU must not be replaced, and this one stays around to relay the result), passes the
    children's stderr through, prints rank 0's single JSON line on its own stdout and exits with the children's
    status.  Under a launcher (RANK set, the driver's documented form) this function is not reached."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:           # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:                       # rank 0 prints exactly one JSON line; anything else goes to stderr
        t = out.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr)
        rc = 1
    sys.exit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args)                        # does not return
    global distributed, native, odometry, se3, synthetic
    import phovo_amd  # noqa: F401
    from phovo_amd import distributed, native, odometry, se3, synthetic
    wl = WORKLOADS[args.workload]
    if args.pairs is None:
        args.pairs = wl.get("pairs", 8192)
    if args.max_iterations is None and wl.get("max_iterations") and args.thresholds == "fixed":
        args.max_iterations = wl["max_iterations"]
    YML = os.path.join(ROOT, "config_files", wl["yml"])
    W, H = wl["size"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch = dist = device = None
    # PHOVO_BENCH_FORCE_DIST=1: initialise torch.distributed even for one rank (rehearses the RCCL code path --
    # init, barrier, all_reduce, all_gather -- on a one-GPU box; launch with torch.distributed.run --nproc-per-node 1)
    force_dist = os.environ.get("PHOVO_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    saved_stdout = None
    if world > 1 or args.gpus > 1 or force_dist:
        # RCCL prints its version banner on stdout when the first communicator is made; stdout carries ONE JSON line,
        # so everything the native libraries write there goes to stderr until that line is printed.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        # (also when a launcher other than launch_ranks() started this rank: the host driver only does dmabuf IPC)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        import torch
        import torch.distributed as dist
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
        # PHOVO_BENCH_BACKEND=gloo is a rehearsal aid for a one-GPU box (RCCL refuses two ranks on one
        # device); the driver's multi-GPU runs use the default, nccl (= RCCL on ROCm).
        backend = os.environ.get("PHOVO_BENCH_BACKEND", "nccl")
        n_dev = native.lib().phovo_device_count()
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            device = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", device_id=device)
        else:
            local_rank = local_rank % max(n_dev, 1)
            device = torch.device("cpu")
            dist.init_process_group(backend)
        print(f"bench.py: rank {rank}/{world} joined the {backend} group", file=sys.stderr, flush=True)

    use_dist = dist is not None
    if native.lib().phovo_device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the alignment path has no CPU fallback")

    # ---- synthetic inputs: one sequence per rank, `distinct` pairs, replicated to `pairs` slots ----
    distinct = max(1, min(args.distinct, args.pairs))
    seq = synthetic.make_sequence(seed=100 + rank, n_frames=distinct + 1, width=W, height=H, holes=0.01, scene=args.scene,
                                  workers=min(16, os.cpu_count() or 1))
    reps = (args.pairs + distinct - 1) // distinct
    cfg_ref = native.read_config_file(YML)
    nl = cfg_ref.num_levels
    cfg_fixed = native.read_config_file(YML)
    if args.max_iterations:
        override = [int(v) for v in args.max_iterations.split(",")]
        if len(override) != nl:
            raise SystemExit(f"--max-iterations needs {nl} values for {wl['yml']}")
        for l in range(nl):
            cfg_ref.max_num_iterations[l] = cfg_fixed.max_num_iterations[l] = override[l]
    max_iter = list(cfg_ref.max_num_iterations[:nl])
    shipped = args.thresholds == "shipped"
    if not shipped:
        for l in range(nl):
            cfg_fixed.min_gradient_norm[l] = 0.0
    min_grad_timed = [float(cfg_fixed.min_gradient_norm[l]) for l in range(nl)]

    eng = odometry.AlignmentEngine(local_rank)
    eng.set_batch_invariant(True)      # the same kernels whatever --pairs is (no latency forms for small batches)
    storage_code = {"f64": native.STORAGE_F64, "f32": native.STORAGE_F32, "f16": native.STORAGE_F16}[args.storage]
    # bytes per pixel of the five planes a pixel-iteration reads (I0, D0, I1, GX1, GY1) in this storage
    plane_bytes = {"f64": 40.0, "f32": 20.0, "f16": 12.0}[args.storage]
    if args.storage != "f64" or args.huber > 0 or args.bilinear:
        eng.set_extensions(native.make_extensions(
            plane_storage=storage_code, huber_delta=[args.huber] * nl,
            sampling=native.SAMPLING_BILINEAR if args.bilinear else native.SAMPLING_NEAREST_SCATTER,
            jacobian_corrected=args.bilinear))
    eng.set_config(cfg_fixed)
    eng.set_intrinsic_matrix(seq["K"])
    n_frames = reps * (distinct + 1)
    eng.reserve_frames(n_frames, W, H)
    src, tgt = [], []
    for r in range(reps):                       # every replica has its own copy of the planes in HBM
        base = r * (distinct + 1)
        eng.upload_frames(base, seq["gray"], seq["depth"])
        for t in range(distinct):
            src.append(base + t)
            tgt.append(base + t + 1)
    # (int32 arrays: what the C ABI takes; as Python lists the two conversions cost 0.3 ms of host time per step, which in a
    # one-enqueue-at-a-time loop is 1 % of a fixed-iteration step and 6 % of one with the shipped thresholds)
    src, tgt = np.array(src[:args.pairs], dtype=np.int32), np.array(tgt[:args.pairs], dtype=np.int32)
    n_local = len(src)
    n_global = n_local * world
    level_sizes = [eng.level_size(l)[0] * eng.level_size(l)[1] for l in range(nl)]

    def barrier():
        if use_dist:
            dist.barrier()
            if device.type == "cuda":
                torch.cuda.synchronize()
        eng.synchronize()

    # ---- timed region: fixed-iteration mode ---------------------------------------------------
    run_steps(eng, src, tgt, args.warmup, use_dist, device, n_global)
    if use_dist:
        if args.warmup == 0:
            run_steps(eng, src, tgt, 1, False, device, n_global)       # something to compare
        probe_zero_copy(eng, n_local, device)
    barrier()
    pipe_main = args.pipeline == "on" or (args.pipeline == "auto" and shipped)
    serial = None
    if pipe_main:        # the per-enqueue event spans of a pipelined run overlap: take them from a serial run of the same steps first
        wall_s, per_level_ms = run_steps(eng, src, tgt, args.steps, use_dist, device, n_global)
        barrier()
        serial = dict(value=n_local * world * args.steps / wall_s, ms_per_step=1e3 * wall_s / args.steps)
        wall, _ = run_steps(eng, src, tgt, args.steps, use_dist, device, n_global, pipelined=True)
    else:
        wall, per_level_ms = run_steps(eng, src, tgt, args.steps, use_dist, device, n_global)
    step_wall_ms = step_wall_summary()
    barrier()
    if use_dist:
        tmax = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())
    states, reports = eng.fetch_results(n_local, want_reports=True)
    timed_launches = eng.last_launches()         # what one timed step launched, as the engine recorded it
    iters = np.array([list(r.iterations[:nl]) for r in reports])
    nonfinite = int(sum(1 for r in reports if r.flags & native.PAIR_NONFINITE))
    window_fallback = int(sum(1 for r in reports if r.flags & native.PAIR_WINDOW_FALLBACK))
    value = n_global * args.steps / wall
    ms_per_step = 1e3 * wall / args.steps

    # per-launch roofline: algorithmic bytes of one launch / its average duration (HIP events on the enqueue's own
    # stream, one start/stop pair around each launch; a fused launch covers several levels)
    levels_out = launch_rows(timed_launches, per_level_ms, args.steps, iters, level_sizes, plane_bytes, max_iter, n_local,
                             {"f64": "double, double", "f32": "float, float", "f16": "__half, float"}[args.storage])
    dom = max(levels_out, key=lambda d: d["avg_launch_ms"])
    total_bytes = sum(d["algorithmic_bytes"] for d in levels_out)
    total_ms = per_level_ms[native.MAX_LEVELS] / args.steps       # device time of a whole enqueue (HIP events, first launch to last)
    roofline = dict(bound="hbm", kernel=dom["kernel"], levels_of_kernel=dom["levels"],
                    achieved=dom["achieved_GBs"], peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=dom["achieved_GBs"] / HBM_PEAK_GBS, traffic=None,
                    all_levels_achieved=total_bytes / (total_ms * 1e-3) / 1e9,
                    all_levels_frac=total_bytes / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    launches=levels_out)
    # What the PMC counters say about that kernel: collected in separate rocprofv3 --pmc passes of this same command
    # (profiles/README.md; they cannot be measured from inside the run) -- HBM-side traffic per launch (FETCH_SIZE + WRITE_SIZE,
    # calibrated) and how busy the vector units are (SQ_ACTIVE_INST_VALU).  `bound` stays "hbm", the roofline SURVEY.md 8(d)
    # prices the path against; which of the two limits the stored counters show nearer saturation is said in
    # from_stored_profile.nearer_saturated (this path's arithmetic is fp64 on the vector unit: with narrow plane storages or
    # few re-reads from HBM that unit, not the memory system, is what the kernel waits for).
    plain_mode = args.huber == 0.0 and not args.bilinear      # the counters were collected on the reference path
    ctx = profile_context(dom["kernel"], n_local if plain_mode or args.bilinear else -1)
    # Everything below comes from profiles/ -- another run, maybe another box -- and is kept apart under its own name;
    # `traffic` (the contract's field) is filled from it only when the profile was collected on THIS build of the library.
    stored = None
    if ctx:
        # the same BUILD: the library file the counters were collected on, or -- the file is not reproducible bit for bit across
        # build directories -- a library built from the same sources
        lib_sha, src_sha = library_sha256(), native.source_sha256()
        same_build = ctx.get("profile_library_sha256") == lib_sha or (ctx.get("profile_source_sha256") is not None and
                                                                      ctx.get("profile_source_sha256") == src_sha)
        stored = dict(note="read from committed rocprofv3 counter passes (profiles/), NOT measured in this run; the "
                           "*_with_this_runs_duration figures divide the stored bytes by this run's launch duration",
                      library_sha256_of_profile=ctx.get("profile_library_sha256"), source_sha256_of_profile=ctx.get("profile_source_sha256"),
                      commit_of_profile=ctx.get("profile_commit"),
                      library_sha256_loaded=lib_sha, source_sha256_of_this_tree=src_sha, same_build_as_loaded_library=bool(same_build))
        if ctx.get("traffic") is not None:
            stored["traffic"] = ctx["traffic"]
            stored["traffic_source"] = ctx["traffic_source"]
            stored["traffic_over_algorithmic"] = ctx["traffic"] / dom["algorithmic_bytes"]
            stored["traffic_GBs_with_this_runs_duration"] = ctx["traffic"] / (dom["avg_launch_ms"] * 1e-3) / 1e9
            if same_build:
                roofline["traffic"] = ctx["traffic"]
                roofline["traffic_source"] = ctx["traffic_source"] + " -- same library build as this run, other process"
        if ctx.get("valu") is not None:
            v = dict(ctx["valu"])
            chunk_iterations = sum(dom["pair_iterations_per_level"][str(l)] * ((level_sizes[l] + 63) // 64) for l in dom["levels"])
            if v.get("instructions_per_launch") and chunk_iterations > 0:
                v["wave_instructions_per_64_pixel_iteration"] = v["instructions_per_launch"] / chunk_iterations
            stored["valu"] = v
            mem_sat = stored.get("traffic_GBs_with_this_runs_duration", dom["achieved_GBs"]) / HBM_PEAK_GBS
            stored["nearer_saturated"] = "valu" if v["busy"] > mem_sat else "hbm"
        if ctx.get("hbm_stream_measured") is not None:
            stored["hbm_stream_measured"] = ctx["hbm_stream_measured"]
    roofline["from_stored_profile"] = stored
    roofline["step_wall_ms"] = step_wall_ms

    # ---- shipped thresholds (reference termination), same resident inputs ---------------------
    ref_term = None
    if not args.no_reference_termination and not shipped:
        eng.set_config(cfg_ref)
        run_steps(eng, src, tgt, 1, use_dist, device, n_global)
        barrier()
        k2 = max(2, args.steps)
        wall2, lv2 = run_steps(eng, src, tgt, k2, use_dist, device, n_global)           # one enqueue at a time: event spans
        steps_serial = step_wall_summary()
        barrier()
        wall2_serial = wall2
        steps_pipelined = None
        if args.pipeline != "off":
            wall2, _ = run_steps(eng, src, tgt, k2, use_dist, device, n_global, pipelined=True)
            steps_pipelined = step_wall_summary()
            barrier()
        if use_dist:
            tmax = torch.tensor([wall2, wall2_serial], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            wall2, wall2_serial = float(tmax[0].item()), float(tmax[1].item())
        _, reps2 = eng.fetch_results(n_local, want_reports=True)
        it2 = np.array([list(r.iterations[:nl]) for r in reps2])
        hist = {}
        for l in range(nl):
            if max_iter[l] > 0:                  # iterations executed -> number of pairs (this rank's pairs)
                vals, counts = np.unique(it2[:, l], return_counts=True)
                hist[f"level_{l}"] = {int(v): int(c) for v, c in zip(vals, counts)}
        ref_term = dict(value=n_global * k2 / wall2, unit="alignments/s", steps=k2,
                        pipelined=args.pipeline != "off", one_enqueue_at_a_time=n_global * k2 / wall2_serial,
                        distinct_pairs=distinct,
                        step_wall_ms=dict(one_enqueue_at_a_time=steps_serial, pipelined=steps_pipelined),
                        mean_iterations_per_level=[float(x) for x in it2.mean(axis=0)],
                        max_iterations_per_level=[int(x) for x in it2.max(axis=0)],
                        avg_launch_ms_per_level=[float(x) / k2 for x in lv2[:nl]],
                        avg_enqueue_ms=float(lv2[native.MAX_LEVELS]) / k2,
                        note="a fused launch covers several levels: its span is reported at the coarsest of them; "
                             "avg_enqueue_ms is first launch to last",
                        iteration_histogram=hist,
                        launches=eng.last_launches())

    # ---- PCIe-inclusive figure (never `value`): raw frames in host memory -> poses --------------------------
    # batched u16-depth upload + device pyramids + Optimize() with the shipped thresholds, one sequence of
    # `distinct + 1` frames per replica, as the VisualOdometry app's --batch mode drives the engine.
    e2e = None
    if rank == 0 and world == 1 and not args.no_reference_termination:
        d16 = np.rint(seq["depth"] * 5000.0).astype(np.uint16)
        gray_all = np.ascontiguousarray(seq["gray"])
        # the caller's buffers are page-locked once (phovo_host_register = hipHostRegister), as a capture pipeline that
        # reuses its frame buffers would do: uploads are then direct DMA
        pinned = []
        if os.environ.get("PHOVO_BENCH_PAGEABLE") != "1":
            for arr in (gray_all, d16):
                if native.lib().phovo_host_register(arr.ctypes.data, arr.nbytes) == 0:
                    pinned.append(arr)
        t0 = time.perf_counter()
        for r in range(reps):
            eng.upload_frames(r * (distinct + 1), gray_all, d16, depth_scale=1.0 / 5000.0)
        eng.enqueue_align(src, tgt)
        eng.synchronize()
        eng.fetch_results(n_local)
        t_e2e = time.perf_counter() - t0
        for arr in pinned:
            native.lib().phovo_host_unregister(arr.ctypes.data)
        e2e = dict(value=n_local / t_e2e, unit="alignments/s", host_buffers="page-locked" if pinned else "pageable",
                   note="host buffers in (u8 gray + u16 depth, 0.92 MB/frame over PCIe), device pyramids, Optimize() "
                        "with the shipped thresholds, poses out; bound by PCIe, not by the alignment kernels")

    # ---- one pair at a time through the class surface (BASELINE.json configs[1] read literally; never `value`) ----
    single = None
    if (rank == 0 and world == 1 and not args.no_reference_termination and plain_mode and args.storage == "f64"
            and args.workload == "cfg2"):
        single = {}
        with odometry.CPhotoconsistencyOdometryAnalytic(local_rank) as po:
            po.SetIntrinsicMatrix(seq["K"])
            for forms in ("default", "latency_forms"):
                po.SetLatencyForms(forms == "latency_forms")
                for name, cfg_one in (("shipped_thresholds", cfg_ref), ("fixed_iterations", cfg_fixed)):
                    po.SetConfiguration(cfg_one)
                    po.SetSourceFrame(seq["gray"][0], seq["depth"][0])
                    po.SetTargetFrame(seq["gray"][1], seq["depth"][1])
                    dev_ms, wall_ms = [], []
                    for _ in range(20):
                        po.SetInitialStateVector(np.zeros(6))
                        t0 = time.perf_counter()
                        po.Optimize()
                        wall_ms.append((time.perf_counter() - t0) * 1e3)
                        dev_ms.append(po.LastOptimizeMilliseconds())
                    single.setdefault(forms, {})[name] = dict(
                        device_ms=float(np.median(dev_ms)), host_wall_ms=float(np.median(wall_ms)),
                        iterations=[int(v) for v in po.GetReport().iterations[:nl]])
        single["note"] = (f"one {W}x{H} pair per Optimize() call (SetSourceFrame/SetTargetFrame outside the timer, as the "
                          "reference's FrameAlignment app times it): latency-bound.  default: the kernels and geometries a batch "
                          "takes (one arithmetic per pair: bit-identical to the same pair in a batch); latency_forms: the opt-in "
                          "forms that finish soonest for one pair (phovo_odometry_set_latency_forms; last bits may differ)")

    # ---- CPU baseline: the oracle, one thread, bounded sample of the same workload ------------
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        ocfg = oracle.make_config(num_levels=nl, max_iter=max_iter, min_grad=min_grad_timed)
        done, t_cpu, t = 0, 0.0, 0
        # The oracle's poses of this leg are kept and compared -- outside every timed region -- with the poses the
        # timed GPU steps left in `states` (slot t of replica 0 is the pair (frame t, frame t+1)): the checker
        # checking the thing measured, on the full 50 + 20 iterations no clipped test reaches.
        check_parity = plain_mode and args.storage == "f64"
        worst, its_equal, checked = 0.0, True, set()
        while t_cpu < args.cpu_seconds:
            i0p, d0p = oracle.build_source_pyramids(seq["gray"][t], seq["depth"][t], ocfg)
            i1p, gxp, gyp = oracle.build_target_pyramids(seq["gray"][t + 1], ocfg)
            c0 = time.perf_counter()
            es, eits = oracle.optimize(ocfg, seq["K"], i0p, d0p, i1p, gxp, gyp)   # Optimize() only, as the reference times it
            t_cpu += time.perf_counter() - c0
            done += 1
            if check_parity and t not in checked and t < n_local:
                checked.add(t)
                worst = max(worst, float(se3.state_distance(states[t], es)))
                its_equal = its_equal and [int(v) for v in iters[t]] == [int(v) for v in eits]
            t = (t + 1) % distinct
        if check_parity:
            # the bar of tests/test_gpu_parity.py (north_star's own is 1e-5); a run over it is not a result
            parity = dict(checked=len(checked), max_pose_distance=worst, iterations_equal=bool(its_equal),
                          bar=PARITY_BAR, ok=bool(worst < PARITY_BAR and its_equal),
                          note="poses of the timed steps vs the CPU oracle on the same pairs, "
                               "||log(T_gpu^-1 T_cpu)||; compared outside the timed region")
        cpu = dict(value=done / t_cpu, unit="alignments/s", cores=1, kind="port",
                   sample=f"{done} alignments over the same synthetic {W}x{H} pairs, fixed-iteration mode, Optimize() only "
                          f"({t_cpu:.1f} s, gcc -O3 -mtune=native, single thread as the reference builds)",
                   host_cpus=os.cpu_count())

    # the same oracle on the GPU box's CPU share (16 threads for one GPU), independent pairs per thread
    cpu_all = None
    if cpu is not None and args.cpu_threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle
        ocfg = oracle.make_config(num_levels=nl, max_iter=max_iter, min_grad=min_grad_timed)
        pyr = []
        for t in range(min(distinct, args.cpu_threads)):
            i0p, d0p = oracle.build_source_pyramids(seq["gray"][t], seq["depth"][t], ocfg)
            i1p, gxp, gyp = oracle.build_target_pyramids(seq["gray"][t + 1], ocfg)
            pyr.append((i0p, d0p, i1p, gxp, gyp))
        budget = args.cpu_seconds

        def worker(i):
            n_done, t_end = 0, time.perf_counter() + budget
            while time.perf_counter() < t_end:
                oracle.optimize(ocfg, seq["K"], *pyr[i % len(pyr)])       # ctypes releases the GIL
                n_done += 1
            return n_done
        c0 = time.perf_counter()
        with ThreadPoolExecutor(args.cpu_threads) as ex:
            total = sum(ex.map(worker, range(args.cpu_threads)))
        wall_cpu = time.perf_counter() - c0
        cpu_all = dict(value=total / wall_cpu, unit="alignments/s", cores=args.cpu_threads, kind="port",
                       sample=f"{total} alignments, {args.cpu_threads} threads x {budget:.0f} s, independent pairs per "
                              f"thread, fixed-iteration mode, Optimize() only")

    if rank == 0:
        out = {
            "metric": wl["metric"],
            "value": value,
            "unit": "alignments/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # arithmetic is fp64 in every mode; a narrow plane storage (an extension, never the default) is named too
            "dtype": "f64" if args.storage == "f64" else f"f64 arithmetic on {args.storage} planes",
            "data": "synthetic",
            "config": {
                "workload": f"{wl['yml']} on synthetic {W}x{H} RGB-D ({args.scene} scene), "
                            + ("SHIPPED thresholds in the timed region (diagnostic: data-dependent early stop; at most "
                               if shipped else "fixed-iteration mode (min_gradient_norm=0: ")
                            + " + ".join(f"{max_iter[l]} iterations at {eng.level_size(l)[0]}x{eng.level_size(l)[1]}"
                                         for l in range(nl - 1, -1, -1) if max_iter[l] > 0)
                            + " per pair), Optimize() only with pyramids resident in HBM",
                "thresholds": args.thresholds, "scene": args.scene,
                "pairs_per_gpu": n_local, "global_pairs_per_step": n_global,
                "distinct_pairs_per_gpu": distinct, "image": [W, H], "levels": nl,
                "max_num_iterations": max_iter, "max_num_iterations_overridden": bool(args.max_iterations),
                "parallelism": f"pairs sharded x{world}, "
                               + ("RCCL" if os.environ.get("PHOVO_BENCH_BACKEND", "nccl") == "nccl" else
                                  os.environ["PHOVO_BENCH_BACKEND"] + " (rehearsal backend, not RCCL)")
                               + " all_gather of states",
                "all_gather_from_device_buffer": ZERO_COPY["ok"],
            },
            "pipelined": bool(pipe_main),
            "one_enqueue_at_a_time": serial,      # pipelined runs only: the same steps, each waited for before the next
            "iterations_per_pair": [float(x) for x in iters.mean(axis=0)],
            "nonfinite_pairs": nonfinite,
            "window_fallback_pairs": window_fallback,
            "parity": parity,
            "algorithmic_MB_per_alignment": algorithmic_bytes(
                level_sizes, [m if max_iter[l] > 0 else 0.0 for l, m in enumerate(iters.mean(axis=0))]) / 1e6
            * plane_bytes / 40.0,
            "extensions": None if (args.storage == "f64" and args.huber <= 0 and not args.bilinear) else
            {"plane_storage": args.storage, "huber_delta": args.huber, "bilinear_corrected": args.bilinear,
             "note": "not in the reference; arithmetic stays fp64, planes are rounded once when stored"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "reference_termination": ref_term,
            # --thresholds shipped (diagnostic): every launch of the timed enqueue with its share of the work, host order
            "launches": timed_launches,
            "end_to_end_pcie_inclusive": e2e,
            "single_pair": single,
        }
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    if parity is not None and not parity["ok"]:
        raise SystemExit(f"bench.py: the timed poses differ from the CPU oracle's: {parity}")


if __name__ == "__main__":
    main()

"""Soak test of the work queue's pair hand-over: N launches of 2500 shuffled copies of three problems (5-level
configuration, 5 fixed iterations per active level); every copy of a problem must come out bit-identical in every launch.
    python tools/queue_soak.py [launches=500] [fixed|shipped|layered|pipelined|slide]
`shipped` keeps the yml's thresholds instead (data-dependent termination: 80x60 and 160x120 are ONE fused launch in which a
pair flows through both levels, gn_fused_kernel); `layered` is `shipped` on problems of the layered scene (most pairs run
long); `pipelined` is `shipped` with two enqueues in flight at a time (tickets, phovo_hip.h "Pipelining"); `slide` soaks the
sliding-window kernel (320x240, one level, 6 fixed iterations, a problem with a 0.3 rad in-plane rotation among them, i.e.
its hand-over to the exact kernel too).
Before the explicit LDS wait in front of the loop-head barrier about one launch in ten failed."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 500
mode = sys.argv[2] if len(sys.argv) > 2 else "fixed"
ncfg = native.read_config_file(os.path.join(os.path.dirname(__file__), "..", "config_files",
                                            "config_5_level_optimization_analytic.yml"))
nl = ncfg.num_levels
W, H = 640, 480
if mode == "fixed":
    ncfg = native.make_config(num_levels=nl, max_iter=[min(m, 5) for m in ncfg.max_num_iterations[:nl]], min_grad=[0.0] * nl)
elif mode == "slide":
    W, H, nl = 320, 240, 1
    ncfg = native.make_config(num_levels=1, max_iter=[6], min_grad=[0.0])
probs = [synthetic.make_pair(21, W, H, holes=0.02, trans=0.004, rot=0.002),
         synthetic.make_pair(22, W, H, holes=0.0, trans=0.03, rot=0.015),
         synthetic.make_pair(23, W, H, holes=0.05, trans=0.06, rot=0.03)]
if mode == "layered":               # shipped thresholds on the layered scene: two of three problems run long on every level
    probs = [synthetic.make_pair(31, W, H, scene="layered", trans=0.05, rot=0.012, invalid=0.25),
             synthetic.make_pair(22, W, H, holes=0.0, trans=0.03, rot=0.015),
             synthetic.make_pair(33, W, H, scene="layered", trans=0.08, rot=0.02, invalid=0.3)]
if mode == "slide":                 # one problem that leaves the window: rendered with a large in-plane rotation
    from phovo_amd import se3
    scene = synthetic.Scene(91)
    K = synthetic.intrinsics(W, H)
    g0, d0 = synthetic.render(scene, np.eye(4), W, H, K, 0.02, hole_seed=1)
    g1, d1 = synthetic.render(scene, se3.eigen_pose([0.01, -0.005, 0.004, 0.30, 0.002, -0.003]), W, H, K, 0.02, hole_seed=2)
    probs[2] = dict(gray0=g0, depth0=d0, gray1=g1, depth1=d1, K=K)
order = np.random.RandomState(5).randint(0, 3, size=2500)
bad = 0
with odometry.AlignmentEngine() as eng:
    eng.set_config(ncfg)
    eng.set_intrinsic_matrix(probs[0]["K"])
    eng.reserve_frames(6, W, H)
    for i, p in enumerate(probs):
        eng.upload_frame(2 * i, p["gray0"], p["depth0"], roles=native.ROLE_SOURCE)
        eng.upload_frame(2 * i + 1, p["gray1"], None, roles=native.ROLE_TARGET)
    src, tgt = [2 * int(i) for i in order], [2 * int(i) + 1 for i in order]
    init = None
    if mode == "slide":              # start the rotated problem near its rotation, so that it leaves the window at once
        init = np.zeros((len(src), 6))
        init[order == 2, 3] = 0.29
    ref = eng.align_pairs(src, tgt, init_states=init)
    for i in range(3):
        idx = np.where(order == i)[0]
        assert all(np.array_equal(ref[idx[0]], ref[k]) for k in idx), "first launch already inconsistent"
    pending = None
    for t in range(launches):
        if mode == "pipelined":          # launch t is issued before launch t - 1 is fetched
            eng.enqueue_align(src, tgt, init_states=init)
            ticket = eng.last_ticket()
            s = eng.fetch(pending, len(src)) if pending is not None else ref
            pending = ticket
        else:
            s = eng.align_pairs(src, tgt, init_states=init)
        if not np.array_equal(s, ref):
            bad += 1
            print(f"launch {t}: {int((s != ref).any(axis=1).sum())} pairs differ")
    if pending is not None and not np.array_equal(eng.fetch(pending, len(src)), ref):
        bad += 1
print(f"{mode}: {launches} launches of 2500 pairs, {bad} with deviations")
sys.exit(1 if bad else 0)

"""GPU tests of the fused level launch (run with -m gpu on an MI355X), through the C ABI against the CPU oracle.

The reference's Optimize() loops per pair over levels (...Analytic.h:502-563).  With a gradient threshold a pair leaves a
level after a data-dependent number of iterations (:376-392), so consecutive levels that fit the 512-thread scatter kernel
are ONE persistent launch in which a workgroup runs a pair through all of them (csrc/gn_kernels.hip, gn_fused_kernel;
phovo_engine_set_level_fusion).  Bars as in test_gpu_parity.py: identical iteration counts, poses within 1e-9 of the oracle
(the specification's bar is 1e-5); fused against the same geometry launched level by level: bit for bit.
"""
import os

import numpy as np
import pytest

import phovo_amd  # noqa: F401
from phovo_amd import native, odometry, se3, synthetic
from oracle import oracle

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_DIR = os.path.join(ROOT, "config_files")
POSE_TOL = 1e-9


def _problems():
    probs = [synthetic.make_pair(21, 640, 480, holes=0.02, trans=0.004, rot=0.002),
             synthetic.make_pair(22, 640, 480, holes=0.0, trans=0.03, rot=0.015),
             synthetic.make_pair(23, 640, 480, holes=0.05, trans=0.06, rot=0.03),
             synthetic.make_pair(24, 640, 480, holes=0.01, trans=0.09, rot=0.05),
             synthetic.make_pair(411, 640, 480, scene="layered", trans=0.07, rot=0.01)]
    # three pairs of bench.py's own sequence that stall above the shipped threshold: 12 and 20 iterations at 160x120, 18 at 80x60
    seq = synthetic.make_sequence(seed=100, n_frames=17, width=640, height=480, holes=0.01)
    for t in (11, 12, 15):
        probs.append(dict(K=seq["K"], gray0=seq["gray"][t], depth0=seq["depth"][t], gray1=seq["gray"][t + 1]))
    return probs


def _upload(eng, probs):
    eng.reserve_frames(2 * len(probs), 640, 480)
    for i, p in enumerate(probs):
        eng.upload_frame(2 * i, p["gray0"], p["depth0"], roles=native.ROLE_SOURCE)
        eng.upload_frame(2 * i + 1, p["gray1"], None, roles=native.ROLE_TARGET)


@pytest.mark.parametrize("yml", ["config_4_level_optimization_analytic.yml", "config_5_level_optimization_analytic.yml"])
def test_fused_levels_equal_level_by_level_launches_and_the_oracle(yml):
    """2500 mixed pairs -- problems that stop after 1-3 iterations next to ones that run to max_num_iterations -- with the
    shipped thresholds, three ways: consecutive levels fused into one launch (default), the same geometry launched level by
    level (PHOVO_FUSION_SPLIT) and one launch per level in each level's own geometry (PHOVO_FUSION_OFF).  Identical
    iteration counts and valid-pixel counts everywhere, every copy of a problem bit-identical within a run, poses equal to
    the oracle's; fused and split agree bit for bit (states, gradient norms): per level they run the same arithmetic in the
    same order, only the launch boundaries differ."""
    ncfg = native.read_config_file(os.path.join(CFG_DIR, yml))
    nl = ncfg.num_levels
    max_iter, min_grad = list(ncfg.max_num_iterations[:nl]), list(ncfg.min_gradient_norm[:nl])
    ocfg = oracle.make_config(num_levels=nl, max_iter=max_iter, min_grad=min_grad)
    probs = _problems()
    expect = [oracle.align_frames(ocfg, p["K"], p["gray0"], p["depth0"], p["gray1"]) for p in probs]
    its = [e[1] for e in expect]
    assert max(max(i) for i in its) >= 5 and min(max(i) for i in its) <= 3, its       # long and short pairs side by side
    order = np.random.RandomState(9).randint(0, len(probs), size=2500)
    src, tgt = [2 * int(i) for i in order], [2 * int(i) + 1 for i in order]
    out, launches = {}, {}
    with odometry.AlignmentEngine() as eng:
        eng.set_config(ncfg)
        eng.set_intrinsic_matrix(probs[0]["K"])
        _upload(eng, probs)
        for mode in (native.FUSION_AUTO, native.FUSION_SPLIT, native.FUSION_OFF):
            eng.set_level_fusion(mode)
            out[mode] = eng.align_pairs(src, tgt, want_reports=True)
            launches[mode] = eng.last_launches()
    # what was launched: 80x60 and 160x120 are one fused launch; 40x30 (5-level file) stays a launch of its own
    fused = [r for r in launches[native.FUSION_AUTO] if r["kind"] == "fused"]
    assert len(fused) == 1 and fused[0]["levels"] == [3, 2] and fused[0]["threads"] == 512, launches[native.FUSION_AUTO]
    assert all(r["kind"] == "persistent" for r in launches[native.FUSION_OFF])
    assert [r["levels"] for r in launches[native.FUSION_SPLIT] if r["threads"] == 512] == [[3], [2]]
    first = {int(i): pos for pos, i in reversed(list(enumerate(order)))}
    for mode, (sc, rc) in out.items():
        for pos, i in enumerate(order):
            es, eits = expect[int(i)]
            assert list(rc[pos].iterations[:nl]) == eits, (mode, pos)
            assert rc[pos].flags == 0
            assert np.array_equal(sc[pos], sc[first[int(i)]]), (mode, pos)            # position and history do not matter
            assert list(rc[pos].valid_pixels[:nl]) == list(rc[first[int(i)]].valid_pixels[:nl])
        for i, pos in first.items():
            assert se3.state_distance(sc[pos], expect[i][0]) < POSE_TOL, (mode, i)
    a, s = out[native.FUSION_AUTO], out[native.FUSION_SPLIT]
    assert np.array_equal(a[0], s[0])
    assert all(x.gradient_norm == y.gradient_norm and list(x.valid_pixels[:nl]) == list(y.valid_pixels[:nl])
               for x, y in zip(a[1], s[1]))


def test_fused_launch_with_extensions_and_initial_states():
    """The fused launch with what can be combined with it: fp16 planes + Huber weights (the extension copies of pass 2) and
    non-zero initial states, against the identically extended oracle; fused and split bit for bit."""
    nl, max_iter, min_grad = 4, [0, 0, 12, 30], [0.0, 0.0, 150.0, 150.0]
    ncfg = native.make_config(num_levels=nl, max_iter=max_iter, min_grad=min_grad)
    ocfg = oracle.make_config(num_levels=nl, max_iter=max_iter, min_grad=min_grad)
    probs = _problems()[:3]
    huber = [0.0, 0.0, 0.05, 0.05]
    rs = np.random.RandomState(3)
    inits = rs.uniform(-1, 1, size=(len(probs), 6)) * np.array([0.01, 0.01, 0.01, 0.004, 0.004, 0.004])
    src = [2 * (k % len(probs)) for k in range(600)]
    tgt = [s + 1 for s in src]
    init = np.stack([inits[k % len(probs)] for k in range(600)])
    out = {}
    with odometry.AlignmentEngine() as eng:
        eng.set_extensions(native.make_extensions(plane_storage=native.STORAGE_F16, huber_delta=huber))
        eng.set_config(ncfg)
        eng.set_intrinsic_matrix(probs[0]["K"])
        _upload(eng, probs)
        stored = []      # oracle inputs = exactly what the device holds (rounded once to fp16 / fp32); other levels: zeros
        for i in range(len(probs)):
            i0p, d0p, i1p, gxp, gyp = [], [], [], [], []
            for l in range(nl):
                if max_iter[l] > 0:
                    i0, d0, _, _ = eng.get_level_planes(2 * i, l)
                    i1, _, gx, gy = eng.get_level_planes(2 * i + 1, l)
                else:
                    lw, lh = oracle.level_size(640, 480, l)
                    i0 = d0 = i1 = gx = gy = np.zeros((lh, lw))
                i0p.append(i0); d0p.append(d0); i1p.append(i1); gxp.append(gx); gyp.append(gy)
            stored.append((i0p, d0p, i1p, gxp, gyp))
        for mode in (native.FUSION_AUTO, native.FUSION_SPLIT):
            eng.set_level_fusion(mode)
            out[mode] = eng.align_pairs(src, tgt, init_states=init, want_reports=True)
        assert [r["kind"] for r in eng.last_launches()] == ["persistent", "persistent"]
    assert np.array_equal(out[native.FUSION_AUTO][0], out[native.FUSION_SPLIT][0])
    for i, p in enumerate(probs):
        i0p, d0p, i1p, gxp, gyp = stored[i]
        es, eits = oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp, init_state=inits[i], huber_delta=huber)
        s, r = out[native.FUSION_AUTO][0][i], out[native.FUSION_AUTO][1][i]
        assert list(r.iterations[:nl]) == eits, (i, list(r.iterations[:nl]), eits)
        assert se3.state_distance(s, es) < POSE_TOL, (i, se3.state_distance(s, es))


def test_small_batches_and_thresholds_of_zero_take_one_launch_per_level():
    """Fusion is for data-dependent termination in a throughput batch: with min_gradient_norm = 0 (every pair runs
    max_num_iterations; the mode bench.py times) the levels stay separate launches, each in its own best geometry.  One pair
    takes the fused launch like a batch does (a pair's bits do not depend on its batch on levels with the owner map in LDS)
    unless the caller asks for the latency forms (phovo_engine_set_latency_forms), which phovo_engine_set_batch_invariant
    overrides."""
    p = synthetic.make_pair(5, 640, 480, holes=0.01)
    thr = native.make_config(num_levels=4, max_iter=[0, 0, 6, 9], min_grad=[0.0, 0.0, 200.0, 200.0])
    fixed = native.make_config(num_levels=4, max_iter=[0, 0, 6, 9], min_grad=[0.0] * 4)
    with odometry.AlignmentEngine() as eng:
        eng.set_intrinsic_matrix(p["K"])
        eng.set_config(thr)
        _upload(eng, [p])
        eng.align_pairs([0] * 64, [1] * 64)
        assert [r["kind"] for r in eng.last_launches()] == ["fused"]
        inv_one = eng.align_pairs([0], [1])
        assert [r["kind"] for r in eng.last_launches()] == ["fused"]
        eng.set_latency_forms(True)
        one = eng.align_pairs([0], [1])
        assert "fused" not in [r["kind"] for r in eng.last_launches()]
        eng.set_batch_invariant(True)
        assert np.array_equal(eng.align_pairs([0], [1]), inv_one)
        assert [r["kind"] for r in eng.last_launches()] == ["fused"]
        inv_many = eng.align_pairs([0] * 700, [1] * 700)
        assert np.array_equal(inv_many, np.tile(inv_one, (700, 1)))
        assert se3.state_distance(one[0], inv_one[0]) < POSE_TOL
        eng.set_config(fixed)
        eng.align_pairs([0] * 64, [1] * 64)
        ls = eng.last_launches()
        assert [r["kind"] for r in ls] == ["persistent", "persistent"] and [r["threads"] for r in ls] == [256, 1024], ls


def test_two_enqueues_in_flight_change_nothing_but_the_clock():
    """phovo_hip.h, "Pipelining": the engine keeps two enqueues in flight, each on its own stream with its own pair
    buffers, so that the next batch's kernels fill the CUs the last long pairs of a batch leave idle.  Three batches issued
    back to back and fetched one behind come out bit-identical to the same batches run one at a time; a ticket is refused
    once two later enqueues have been issued; an upload issued while a batch is in flight waits for it (the batch still
    sees the frames it was enqueued on)."""
    ncfg = native.read_config_file(os.path.join(CFG_DIR, "config_4_level_optimization_analytic.yml"))
    probs = _problems()
    rs = np.random.RandomState(4)
    batches = []
    for n in (1800, 2300, 1500):
        order = rs.randint(0, len(probs), size=n)
        batches.append(([2 * int(i) for i in order], [2 * int(i) + 1 for i in order]))
    with odometry.AlignmentEngine() as eng:
        eng.set_config(ncfg)
        eng.set_batch_invariant(True)
        eng.set_intrinsic_matrix(probs[0]["K"])
        _upload(eng, probs)
        serial = [eng.align_pairs(s, t, want_reports=True) for s, t in batches]
        tickets, got = [], {}
        for k, (s, t) in enumerate(batches):
            eng.enqueue_align(s, t)
            tickets.append(eng.last_ticket())
            if k > 0:
                got[k - 1] = eng.fetch(tickets[k - 1], len(batches[k - 1][0]), want_reports=True)
        assert tickets == [tickets[0], tickets[0] + 1, tickets[0] + 2]
        with pytest.raises(native.PhovoError) as err:                 # two later enqueues have taken its slot
            eng.fetch(tickets[0], len(batches[0][0]))
        assert err.value.status == native.E_INVALID_ARGUMENT
        # an upload while the last batch may still be running: it waits for the batch, which sees the old frame
        blank = np.zeros_like(probs[0]["gray1"])
        eng.upload_frame(1, blank, None, roles=native.ROLE_TARGET)
        got[2] = eng.fetch(tickets[2], len(batches[2][0]), want_reports=True)
        total_ms, _ = eng.align_ms(tickets[2])
        assert total_ms > 0.0
    for k in range(3):
        assert np.array_equal(got[k][0], serial[k][0]), k
        assert all(list(a.iterations[:4]) == list(b.iterations[:4]) and a.gradient_norm == b.gradient_norm and a.flags == b.flags
                   for a, b in zip(got[k][1], serial[k][1])), k


def test_fused_run_of_two_small_levels_at_another_image_size():
    """448x336: levels 3 and 2 are 56x42 and 112x84 pixels -- both would take the 256-thread geometry on their own, fused they
    take the 512-thread one.  Fused, the same geometry level by level (PHOVO_FUSION_SPLIT: bit-identical, also on the LAST
    level of the run, whose own plan would differ) and one launch per level in each level's own geometry all match the
    oracle."""
    p = synthetic.make_pair(77, 448, 336, holes=0.02, trans=0.02, rot=0.01)
    nl, max_iter, min_grad = 4, [0, 0, 7, 11], [0.0, 0.0, 120.0, 120.0]
    ncfg = native.make_config(num_levels=nl, max_iter=max_iter, min_grad=min_grad)
    ocfg = oracle.make_config(num_levels=nl, max_iter=max_iter, min_grad=min_grad)
    es, eits = oracle.align_frames(ocfg, p["K"], p["gray0"], p["depth0"], p["gray1"])
    out, launches = {}, {}
    with odometry.AlignmentEngine() as eng:
        eng.set_config(ncfg)
        eng.set_intrinsic_matrix(p["K"])
        eng.reserve_frames(2, 448, 336)
        eng.upload_frame(0, p["gray0"], p["depth0"], roles=native.ROLE_SOURCE)
        eng.upload_frame(1, p["gray1"], None, roles=native.ROLE_TARGET)
        assert eng.level_size(3) == (56, 42) and eng.level_size(2) == (112, 84)
        for mode in (native.FUSION_AUTO, native.FUSION_SPLIT, native.FUSION_OFF):
            eng.set_level_fusion(mode)
            out[mode] = eng.align_pairs([0] * 300, [1] * 300, want_reports=True)
            launches[mode] = [(r["kind"], r["threads"]) for r in eng.last_launches()]
    assert launches[native.FUSION_AUTO] == [("fused", 512)]
    assert launches[native.FUSION_SPLIT] == [("persistent", 512), ("persistent", 512)]
    assert launches[native.FUSION_OFF] == [("persistent", 256), ("persistent", 256)]
    for mode, (s, r) in out.items():
        assert list(r[0].iterations[:nl]) == eits, (mode, list(r[0].iterations[:nl]), eits)
        assert se3.state_distance(s[0], es) < POSE_TOL, (mode, se3.state_distance(s[0], es))
        assert all(np.array_equal(s[0], s[k]) for k in range(300))
    assert np.array_equal(out[native.FUSION_AUTO][0], out[native.FUSION_SPLIT][0])


def test_a_refused_enqueue_leaves_the_engine_as_it_was():
    """phovo_engine_enqueue_align validates before it takes a ticket and a slot: an enqueue that is refused (a frame index out
    of range; a level that cannot run: > 2 097 151 pixels with more pairs than the wide form takes) hands out no ticket,
    evicts nothing -- the results of BOTH enqueues in flight stay fetchable under their tickets, bit for bit -- and
    fetch_results keeps meaning the last enqueue that exists.  An empty enqueue is an enqueue: ticket, zero pairs, no device
    buffer; fetching another count under its ticket is refused."""
    ncfg = native.read_config_file(os.path.join(CFG_DIR, "config_4_level_optimization_analytic.yml"))
    probs = _problems()[:3]
    src, tgt = [0, 2, 4] * 50, [1, 3, 5] * 50
    with odometry.AlignmentEngine() as eng:
        eng.set_config(ncfg)
        eng.set_intrinsic_matrix(probs[0]["K"])
        _upload(eng, probs)
        first = eng.align_pairs(src, tgt)
        eng.enqueue_align(src[:100], tgt[:100])
        t1 = eng.last_ticket()
        eng.enqueue_align(src, tgt)
        t2 = eng.last_ticket()
        assert t2 == t1 + 1
        for bad_src in ([0, 2, 99], [-1]):
            with pytest.raises(native.PhovoError) as err:
                eng.enqueue_align(bad_src, [1] * len(bad_src))
            assert err.value.status == native.E_INVALID_ARGUMENT
        assert eng.last_ticket() == t2                                   # no ticket was handed out
        assert np.array_equal(eng.fetch(t1, 100), first[:100])           # nothing was evicted
        assert np.array_equal(eng.fetch(t2, len(src)), first)
        assert np.array_equal(eng.fetch_results(len(src)), first)
        eng.enqueue_align([], [])
        t3 = eng.last_ticket()
        assert t3 == t2 + 1 and eng.fetch(t3, 0).shape == (0, 6) and eng.device_states(t3) is None
        with pytest.raises(native.PhovoError) as err:
            eng.fetch(t3, 5)
        assert err.value.status == native.E_INVALID_ARGUMENT
        assert np.array_equal(eng.fetch(t2, len(src)), first)            # the other slot is untouched
    # a configuration whose only active level is too large for the persistent kernels, in a batch too large for the wide form
    big = native.make_config(num_levels=1, max_iter=[2], min_grad=[0.0])
    g = np.zeros((1200, 1800), dtype=np.uint8)
    d = np.ones((1200, 1800))
    with odometry.AlignmentEngine() as eng:
        eng.set_config(big)
        eng.set_intrinsic_matrix(synthetic.intrinsics(1800, 1200))
        eng.reserve_frames(2, 1800, 1200)
        eng.upload_frame(0, g, d)
        eng.upload_frame(1, g, d)
        ok = eng.align_pairs([0], [1])                                   # one pair: the wide form, no size limit
        t = eng.last_ticket()
        with pytest.raises(native.PhovoError) as err:
            eng.enqueue_align([0] * 40, [1] * 40)
        assert err.value.status == native.E_SHAPE
        assert eng.last_ticket() == t and np.array_equal(eng.fetch(t, 1), ok, equal_nan=True)

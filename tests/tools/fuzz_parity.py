"""Randomised device-vs-oracle sweep (run on the GPU box): odd image sizes, 1-3 levels, intrinsics that are not
half-integers, depth holes / NaN / out-of-range depth, large motions, non-zero initial states, every launch geometry
the sizes select, optionally narrow storage / Huber / bilinear.  Prints one line per failure and a summary; exit code 1
if any case misses the pose bar or an iteration count.  A case beyond the bar whose oracle result itself moves by more than a
quarter of that distance under one ulp of fx is reported as chaotic instead (a diverging alignment: round 5, seed 91 case 382).

The pose bar is 1e-9 x max(1, cond(J^T J) / 1e5), with cond the largest condition number of the normal equations over the
oracle's iterations (printed with every failure and, as a maximum, in the summary).  Device and oracle sum the same terms
in a different order, i.e. they solve normal equations that differ by ~1e-14 relative; the solve returns that times the
condition number, iteration after iteration where the iteration does not converge.  What the reference's configurations
produce on 640x480 pyramids has cond 1e2 ... 1e3 (plane and layered scenes, levels 2-4), two orders below where the
scaling starts: there the bar is the flat 1e-9 the GPU tests hold.  It matters for degenerate shapes only -- strips of a
dozen rows barely constrain the rotation about the image's long axis (cond 1e6 ... 1e8): round 3's sweeps met two such
cases at 1.2e-9 and 1.3e-9 (282x15, 309x15), round 4's strips sweep cases up to 5e-9 at cond 2e6.

    python tests/tools/fuzz_parity.py [cases=150] [seed=0] [ext] [big] [strips]

With `strips` every case is a one-level strip of 250-330 x 12-20 pixels (that shape class, on purpose).

With `ext` every case also draws a combination of the opt-in extensions (fp32 / fp16 plane storage, Huber weights,
bilinear sampling with or without the corrected Jacobian); the oracle is then fed the planes as the device stored them.
PHOVO_TOOLS_LIBRARY=<path> sweeps another build of the library (e.g. csrc/build_tuning/libphovo_hip_tuning.so with its
environment switches).  With `big` the images are 240x200 ... 700x500 with 1-2 levels, 40 or 300 pairs and in-plane rotations of up to 0.25 rad:
level 0 then exceeds what an owner map in LDS holds, i.e. the sliding-window kernel and, for the large rotations, its
hand-over to the exact kernel (PHOVO_PAIR_WINDOW_FALLBACK) are what is being swept.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tools"))
import _variant  # noqa: E402
_variant.use_from_environment()       # PHOVO_TOOLS_LIBRARY=<a diagnostic build>: sweep that build instead (tools/_variant.py)
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, se3, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
with_ext = "ext" in sys.argv[3:]
big = "big" in sys.argv[3:]
strips = "strips" in sys.argv[3:]
bad, worst, variants, fallbacks = 0, 0.0, {}, 0
worst_ratio, worst_cond, ill = 0.0, 0.0, 0
worst_ill, ill_only_by_scaling = 0.0, 0
chaotic = 0


def worst_condition(trace):
    c = 1.0
    for e in trace:
        h = e["hessian"]
        if np.all(np.isfinite(h)) and np.any(h != 0.0):
            c = max(c, float(np.linalg.cond(h)))
    return c


# FUZZ_ONLY=12,345: replay the random draws of every case but run only these (to look at a failure again)
only = {int(c) for c in os.environ["FUZZ_ONLY"].split(",")} if os.environ.get("FUZZ_ONLY") else None
for case in range(cases):
    active = only is None or case in only
    nl = int(rs.randint(1, 3 if big else 4))
    unit = 2 ** (nl - 1)
    w = int(rs.randint(240, 700) if big else rs.randint(16, 330)) // unit * unit + (unit if rs.rand() < 0.5 else 0)
    h = int(rs.randint(200, 500) if big else rs.randint(12, 250)) // unit * unit + (unit if rs.rand() < 0.5 else 0)
    w, h = max(w, 8 * unit), max(h, 8 * unit)
    if strips:
        nl, unit = 1, 1
        w, h = int(rs.randint(250, 331)), int(rs.randint(12, 21))
    kw = dict(holes=float(rs.choice([0.0, 0.02, 0.2])), trans=float(rs.choice([0.002, 0.02, 0.08])),
              rot=float(rs.choice([0.001, 0.01, 0.05, 0.25] if big else [0.001, 0.01, 0.05])))
    p = synthetic.make_pair(1000 + case, w, h, **kw) if active else dict(K=synthetic.intrinsics(w, h), depth0=np.zeros((h, w)))
    K = p["K"].copy()
    if rs.rand() < 0.6:                     # principal point / focal lengths that are not exactly representable
        K[0, 2] += rs.uniform(-3, 3)
        K[1, 2] += rs.uniform(-3, 3)
        K[0, 0] *= rs.uniform(0.9, 1.1)
        K[1, 1] *= rs.uniform(0.9, 1.1)
    d0 = p["depth0"].copy()
    if rs.rand() < 0.3:
        d0[rs.rand(h, w) < 0.01] = np.nan
        d0[rs.rand(h, w) < 0.01] = 7.5      # beyond max depth
        d0[rs.rand(h, w) < 0.01] = -1.0
    fixed = rs.rand() < 0.5
    max_iter = [int(rs.randint(0, 7)) for _ in range(nl)]
    if sum(max_iter) == 0:
        max_iter[-1] = 3
    min_grad = [0.0] * nl if fixed else [float(rs.choice([1.0, 30.0, 300.0])) for _ in range(nl)]
    lam = [float(rs.choice([1.0, 0.7])) for _ in range(nl)]
    ncfg = native.make_config(num_levels=nl, max_iter=max_iter, min_grad=min_grad, lam=lam)
    ocfg = oracle.make_config(num_levels=nl, max_iter=max_iter, min_grad=min_grad, lam=lam)
    init = None if rs.rand() < 0.5 else rs.uniform(-1, 1, 6) * np.array([0.02, 0.02, 0.02, 0.01, 0.01, 0.01])
    storage, huber, bilinear, corrected = native.STORAGE_F64, None, False, False
    if not active:                          # the remaining draws of this case, in order, and on to the next one
        if with_ext:
            rs.randint(0, 3)
            if not rs.rand() < 0.4:
                [rs.choice([0.0, 0.02, 0.1]) for _ in range(nl)]
            if rs.rand() < 0.5:
                rs.rand()
        if int(rs.choice([40, 300] if big else [1, 3, 40])) <= 8:
            rs.rand()                       # (the latency-forms draw of a small batch)
        continue
    if with_ext:
        storage = [native.STORAGE_F64, native.STORAGE_F32, native.STORAGE_F16][int(rs.randint(0, 3))]
        huber = None if rs.rand() < 0.4 else [float(rs.choice([0.0, 0.02, 0.1])) for _ in range(nl)]
        bilinear = rs.rand() < 0.5
        corrected = bool(bilinear and rs.rand() < 0.5)
    else:
        es, eits, otrace = oracle.align_frames(ocfg, K, p["gray0"], d0, p["gray1"], init_state=init, want_trace=True)
    with odometry.AlignmentEngine() as eng:
        eng.set_config(ncfg)
        if with_ext:
            eng.set_extensions(native.make_extensions(
                plane_storage=storage, huber_delta=huber,
                sampling=native.SAMPLING_BILINEAR if bilinear else native.SAMPLING_NEAREST_SCATTER,
                jacobian_corrected=corrected))
        eng.set_intrinsic_matrix(K)
        eng.reserve_frames(2, w, h)
        eng.upload_frame(0, p["gray0"], d0, roles=native.ROLE_SOURCE)
        eng.upload_frame(1, p["gray1"], None, roles=native.ROLE_TARGET)
        if with_ext:                       # oracle inputs = the planes as stored (rounded once to the storage type)
            planes = [[], [], [], [], []]
            for l in range(nl):
                if max_iter[l] > 0:
                    i0, dd, _, _ = eng.get_level_planes(0, l)
                    i1, _, gx, gy = eng.get_level_planes(1, l)
                else:
                    lw, lh = oracle.level_size(w, h, l)
                    i0 = dd = i1 = gx = gy = np.zeros((lh, lw))
                for lst, v in zip(planes, (i0, dd, i1, gx, gy)):
                    lst.append(v)
            es, eits, otrace = oracle.optimize(ocfg, K, *planes, init_state=init, huber_delta=huber, bilinear=bilinear,
                                               corrected=corrected, want_trace=True)
        n_pairs = int(rs.choice([40, 300] if big else [1, 3, 40]))          # 40 > 32: never the wide form
        if n_pairs <= 8 and rs.rand() < 0.5:
            eng.set_latency_forms(True)         # half of the small batches: the forms that finish soonest (off by default)
        inits = None if init is None else np.tile(init, (n_pairs, 1))
        s, reps = eng.align_pairs([0] * n_pairs, [1] * n_pairs, init_states=inits, want_reports=True)
        for l in range(nl):
            if max_iter[l] > 0:
                info = eng.level_launch_info(l)
                key = (info["threads"], info["owner_in_lds"], info["source_in_lds"], bool(eng.level_uses_wide(l, n_pairs)))
                variants[key] = variants.get(key, 0) + 1
    fallbacks += int(bool(reps[0].flags & native.PAIR_WINDOW_FALLBACK))
    its = list(reps[0].iterations[:nl])
    finite = np.all(np.isfinite(es))
    cond = worst_condition(otrace)
    # flat 1e-9 up to cond(J^T J) = 1e5, scaled with the condition number above, and NEVER looser than the specification's
    # own 1e-5 (north_star); iteration counts must agree whatever the conditioning
    bar = min(1e-5, 1e-9 * max(1.0, cond / 1e5))
    if finite:
        d = se3.state_distance(s[0], es)
        ok = its == eits and d < bar and all(np.array_equal(s[0], s[i]) for i in range(n_pairs))
        worst_ratio, worst_cond = max(worst_ratio, d / bar), max(worst_cond, cond)
        ill += int(cond > 1e5)
        if cond > 1e5:
            worst_ill = max(worst_ill, d)
            ill_only_by_scaling += int(ok and d >= 1e-9)
    else:                                   # the oracle ran into NaN (no valid pixel / singular H): flagged, not hidden
        d = 0.0
        ok = bool(reps[0].flags & native.PAIR_NONFINITE) and not np.all(np.isfinite(s[0]))
    worst = max(worst, d)
    if not ok and finite and its == eits and d < 1e-5 and all(np.array_equal(s[0], s[i]) for i in range(n_pairs)):
        # Missed the scaled bar with equal iteration counts: before calling it a failure, ask the ORACLE how far its own result
        # moves when one focal length changes by one ulp.  An alignment that is diverging (a strip with a 0.05 rad rotation
        # ends at x = 5.7 m, pitch -3.2 rad) amplifies any rounding by more than cond(J^T J) says, and no comparison can be
        # tighter than that; such a case is counted as chaotic (and printed), not as a failure, when the device is within 4 x
        # that one-ulp sensitivity.
        K1 = K.copy()
        K1[0, 0] = np.nextafter(K1[0, 0], 10.0 * K1[0, 0])
        if with_ext:
            es1, _ = oracle.optimize(ocfg, K1, *planes, init_state=init, huber_delta=huber, bilinear=bilinear, corrected=corrected)
        else:
            es1, _ = oracle.align_frames(ocfg, K1, p["gray0"], d0, p["gray1"], init_state=init)
        sens = se3.state_distance(es, es1) if np.all(np.isfinite(es1)) else float("inf")
        if d <= 4.0 * sens:
            chaotic += 1
            ok = True
            print(f"chaotic case {case}: {w}x{h} levels {nl} max_iter {max_iter} distance {d:.3e} bar {bar:.1e} cond {cond:.2e}; "
                  f"one ulp in fx moves the oracle's own result by {sens:.3e}")
    if not ok:
        bad += 1
        print(f"FAIL case {case}: {w}x{h} levels {nl} max_iter {max_iter} min_grad {min_grad} pairs {n_pairs} "
              f"iterations gpu {its} cpu {eits} distance {d:.3e} cond {cond:.2e} bar {bar:.1e} flags {reps[0].flags} "
              f"ext(storage {storage}, huber {huber}, bilinear {bilinear}, corrected {corrected})")
    if (case + 1) % 500 == 0:                 # a long sweep must not look hung to whoever is watching its output
        print(f"... {case + 1} cases so far, {bad} failures, worst {worst:.3e}", flush=True)
print(f"{cases} cases, {bad} failures, worst pose distance {worst:.3e}")
print(f"largest cond(J^T J) {worst_cond:.2e}; {ill} cases above 1e5 (bar scaled, capped at 1e-5): worst absolute distance among them "
      f"{worst_ill:.3e}, {ill_only_by_scaling} of them passed only because of the scaling (distance >= 1e-9); "
      f"worst distance / bar {worst_ratio:.3f}")
print(f"chaotic cases (beyond the scaled bar, within 4 x the oracle's own sensitivity to one ulp of fx; not failures): {chaotic}")
print(f"pairs finished by the exact kernel after leaving the sliding window: {fallbacks} cases")
print("launch geometries exercised (threads, owner in LDS, source in LDS, wide form): ", variants)
sys.exit(1 if bad else 0)

"""Generates the committed golden fixtures tests/golden/case_*.npz.

Each fixture holds the INPUTS of one alignment problem (u8 intensities, fp64 depths,
intrinsics, per-level parameters, initial state) and the EXPECTED per-iteration
normal equations and states, computed by the independent numpy restatement
oracle/numpy_twin.py -- not by the C oracle and not by the HIP path, both of which
are tested against these files.  The reference itself holds no fixtures (it has no
tests) and cannot be built here, so these vectors are manufactured, not recorded
from the reference: parity against the reference stays "unpinned".

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import phovo_amd  # noqa: E402,F401
from phovo_amd import synthetic  # noqa: E402
from oracle import numpy_twin as twin  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, seed, W, H, holes, levels, max_iter, min_grad, lambda, init_state, min/max depth
    dict(name="case_a", seed=0, w=96, h=72, holes=0.0, num_levels=4,
         max_iter=[0, 4, 8, 12], min_grad=[1.0, 150.0, 9.0, 4.5], lam=[1, 1, 1, 1]),
    dict(name="case_b", seed=1, w=96, h=72, holes=0.05, num_levels=3,
         max_iter=[3, 5, 10], min_grad=[0.0, 0.0, 0.0], lam=[1, 0.8, 1]),
    dict(name="case_c", seed=2, w=128, h=96, holes=0.02, num_levels=4,
         max_iter=[0, 0, 6, 10], min_grad=[0.5, 0.5, 25.0, 14.0], lam=[1, 1, 1, 1],
         init=[0.004, -0.003, 0.002, 0.002, -0.001, 0.0015], min_depth=0.5, max_depth=2.2),
    # the layered desk-like scene (synthetic.LayeredScene through sensor_like): depth discontinuities, occlusion (sources
    # of different layers on one target), 20 % invalid REGIONS, Kinect-style depth noise; a larger motion than the others
    dict(name="case_d", seed=11, w=160, h=120, scene="layered", trans=0.06, rot=0.02, num_levels=3,
         max_iter=[4, 8, 12], min_grad=[0.0, 6.0, 3.0], lam=[1, 1, 1]),
]


def main():
    only = sys.argv[1:]                  # python tests/golden/make_golden.py [case_x ...]: regenerate just those
    for c in CASES:
        if only and c["name"] not in only:
            continue
        if c.get("scene", "plane") == "layered":
            p = synthetic.make_pair(c["seed"], c["w"], c["h"], scene="layered", trans=c["trans"], rot=c["rot"])
        else:
            p = synthetic.make_pair(c["seed"], c["w"], c["h"], holes=c["holes"])
        nl = c["num_levels"]
        gs = [0.0625] * nl
        pyr = twin.build_pyramids(p["gray0"], p["depth0"], p["gray1"], nl, gs)
        cfg = dict(num_levels=nl, lam=[float(v) for v in c["lam"]], max_iter=c["max_iter"],
                   min_grad=c["min_grad"], min_depth=c.get("min_depth", 0.3),
                   max_depth=c.get("max_depth", 5.0))
        init = np.array(c.get("init", [0.0] * 6), dtype=np.float64)
        state, iters, trace = twin.optimize(pyr, p["K"], cfg, init)
        out = dict(
            gray0=p["gray0"], depth0=p["depth0"], gray1=p["gray1"], depth1=p["depth1"],
            K=p["K"], motion=p["motion"],
            num_levels=np.int32(nl), grad_scale=np.array(gs), lam=np.array(cfg["lam"]),
            max_iter=np.array(c["max_iter"], dtype=np.int32), min_grad=np.array(c["min_grad"], dtype=np.float64),
            min_depth=np.float64(cfg["min_depth"]), max_depth=np.float64(cfg["max_depth"]),
            init_state=init,
            exp_state=state, exp_iters=np.array(iters, dtype=np.int32),
            exp_trace_level=np.array([t["level"] for t in trace], dtype=np.int32),
            exp_trace_iteration=np.array([t["iteration"] for t in trace], dtype=np.int32),
            exp_trace_gradient=np.array([t["gradient"] for t in trace]),
            exp_trace_hessian=np.array([t["hessian"] for t in trace]),
            exp_trace_state=np.array([t["state"] for t in trace]),
        )
        # coarsest-level planes, to pin the pyramid producers too
        for l in (nl - 1, 1):
            a, d, b, gx, gy = pyr[l]
            out[f"exp_L{l}_i0"], out[f"exp_L{l}_d0"] = a, d
            out[f"exp_L{l}_gx1"], out[f"exp_L{l}_gy1"] = gx, gy
        path = os.path.join(HERE, c["name"] + ".npz")
        np.savez_compressed(path, **out)
        print(c["name"], "iters", iters, "executed", len(trace), "state", state,
              "bytes", os.path.getsize(path))
        for l in range(nl - 1, -1, -1):
            if c["max_iter"][l] > 0:
                print("   level", l, twin.scatter_statistics(pyr[l], l, p["K"], state, cfg["min_depth"], cfg["max_depth"]))


if __name__ == "__main__":
    main()

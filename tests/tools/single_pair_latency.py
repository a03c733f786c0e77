"""Device time of ONE Optimize() call (one frame pair) for the shipped analytic configurations -- the case the
reference's FrameAlignment app times (apps/PhotoconsistencyFrameAlignment/PhotoconsistencyFrameAlignment.cpp:99-102)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import phovo_amd  # noqa: E402,F401
from phovo_amd import native, odometry, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402

CFG = os.path.join(os.path.dirname(__file__), "..", "..", "config_files")
for yml, size in (("config_4_level_optimization_analytic.yml", (640, 480)),
                  ("config_5_level_optimization_analytic.yml", (640, 480)),
                  ("config_6_level_optimization_analytic.yml", (1280, 960)),
                  ("config_only_level_0_analytic.yml", (640, 480))):
    p = synthetic.make_pair(1, size[0], size[1], holes=0.01, trans=0.01, rot=0.005)
    n = native.read_config_file(os.path.join(CFG, yml))
    nl = n.num_levels
    for mode in ("shipped", "fixed"):
        mi = list(n.max_num_iterations[:nl])
        if yml.startswith("config_only") and mode == "fixed":
            mi = [20]                      # 5000 fixed iterations of 307200 px is not a useful probe
        mg = list(n.min_gradient_norm[:nl]) if mode == "shipped" else [0.0] * nl
        ncfg = native.make_config(num_levels=nl, max_iter=mi, min_grad=mg)
        ocfg = oracle.make_config(num_levels=nl, max_iter=mi, min_grad=mg)
        with odometry.CPhotoconsistencyOdometryAnalytic() as po:
            po.SetConfiguration(ncfg)
            po.SetIntrinsicMatrix(p["K"])
            po.SetSourceFrame(p["gray0"], p["depth0"])
            po.SetTargetFrame(p["gray1"], p["depth1"])
            best = 1e9
            for _ in range(3):
                po.SetInitialStateVector(np.zeros(6))
                t0 = time.perf_counter()
                po.Optimize()
                wall = (time.perf_counter() - t0) * 1e3
                best = min(best, po.LastOptimizeMilliseconds())
            its = list(po.GetReport().iterations[:nl])
        i0p, d0p = oracle.build_source_pyramids(p["gray0"], p["depth0"], ocfg)
        i1p, gxp, gyp = oracle.build_target_pyramids(p["gray1"], ocfg)
        t0 = time.perf_counter()
        oracle.optimize(ocfg, p["K"], i0p, d0p, i1p, gxp, gyp)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        print(f"{yml:45s} {size[0]}x{size[1]} {mode:8s} iterations {its}: device {best:8.3f} ms, "
              f"wall {wall:8.3f} ms, CPU oracle {cpu_ms:9.2f} ms")

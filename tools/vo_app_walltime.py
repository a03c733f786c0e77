"""Wall time of the VisualOdometry front ends on a synthetic TUM-layout sequence (run on the GPU box):

    python tools/vo_app_walltime.py [frames=601]

pair-by-pair loop (the reference's structure), --batch (parallel decode, one batched upload, one batched alignment)
and the sharded driver with 1 and 2 gloo ranks on the one GPU.  The dataset generator is the test suite's."""
import importlib.util
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("seqtest", os.path.join(ROOT, "tests", "test_sequence_sharded.py"))
seqtest = importlib.util.module_from_spec(spec)
spec.loader.exec_module(seqtest)

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 601
cfg = os.path.join(ROOT, "config_files", "config_5_level_optimization_analytic.yml")
app = os.path.join(ROOT, "apps", "bin", "PhotoconsistencyVisualOdometry")
sharded = os.path.join(ROOT, "apps", "PhotoconsistencyVisualOdometrySharded.py")
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "apps")])
with tempfile.TemporaryDirectory() as tmp:
    seq = os.path.join(tmp, "seq")
    t0 = time.time()
    seqtest.write_tum_sequence(seq, frames, distinct=40)
    print(f"dataset: {frames} frames written in {time.time() - t0:.1f} s", flush=True)
    runs = [("loop", [app, cfg, seq, os.path.join(tmp, "loop.txt")]),
            ("batch", [app, cfg, seq, os.path.join(tmp, "batch.txt"), "--batch"]),
            ("sharded x1", [sys.executable, sharded, cfg, seq, os.path.join(tmp, "s1.txt")]),
            ("sharded x2 (gloo, one GPU)", [sys.executable, sharded, cfg, seq, os.path.join(tmp, "s2.txt"), "--ranks", "2",
                                            "--backend", "gloo"])]
    outs = {}
    for name, cmd in runs:
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, env=seqtest._clean_env())
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = open(cmd[4 if cmd[0] == sys.executable else 3], "rb").read()
        inner = [l for l in r.stdout.splitlines() if l.startswith("Time =")]
        print(f"{name:28s} {dt:7.2f} s wall  ({(frames - 1) / dt:8.1f} pairs/s)  {inner[-1] if inner and name == 'batch' else ''}", flush=True)
    same = all(v == outs["batch"] for k, v in outs.items() if k != "loop")
    print("batch and sharded trajectory files byte-identical:", same)
    # the loop aligns one pair per call (latency geometry): same poses to the parity bar, not to the last bit
    a = [[float(x) for x in l.split()] for l in outs["loop"].decode().splitlines()[2:]]
    b = [[float(x) for x in l.split()] for l in outs["batch"].decode().splitlines()[2:]]
    print("loop vs batch, largest difference of any printed number:", max(abs(x - y) for r, q in zip(a, b) for x, y in zip(r, q)))

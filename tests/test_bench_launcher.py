"""`python bench.py --gpus N` must start its own ranks when no launcher wraps it (the driver's SCALE runs may call it
either way), and the parent must do so before anything touches the GPU.  These tests drive that logic on the CPU with
PHOVO_BENCH_BACKEND=gloo up to the device check, where the ranks stop ("no CPU fallback")."""
import ast
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, extra_env=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_parent_starts_the_ranks_itself_and_relays_their_status():
    """Two gloo ranks on the CPU: both join the group (world size 2), both stop at the device check, the parent exits
    non-zero with their message and prints no result line."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--pairs", "4", "--distinct", "2",
              "--no-cpu-baseline", "--no-reference-termination"], {"PHOVO_BENCH_BACKEND": "gloo"})
    assert "starting 2 ranks" in r.stderr
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr
    assert "rank 0/2 joined the gloo group" in r.stderr and "rank 1/2 joined the gloo group" in r.stderr
    import phovo_amd  # noqa: F401
    from phovo_amd import native
    if native.lib().phovo_device_count() < 1:
        assert r.returncode != 0
        assert "needs an MI355X" in r.stderr
        assert r.stdout.strip() == ""
    else:                                           # on a GPU box this is a real two-rank run on the shared card
        assert r.returncode == 0, r.stderr[-2000:]
        import json
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2


def test_under_a_launcher_the_parent_path_is_not_taken():
    """RANK set = already one of the ranks: no second launcher is started (the world-size check speaks instead)."""
    r = _run(["--gpus", "2", "--no-cpu-baseline"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                                                      "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999",
                                                      "PHOVO_BENCH_BACKEND": "gloo"})
    assert "starting 2 ranks" not in r.stderr
    assert r.returncode != 0 and "needs WORLD_SIZE=2" in r.stderr


def test_parent_touches_nothing_gpu_side_before_launching():
    """Structure check: at module level bench.py imports neither torch nor the package (whose import loads
    libphovo_hip.so and with it the HIP runtime), and main() calls launch_ranks() before its first such import."""
    tree = ast.parse(open(BENCH).read())
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    names = {a.name.split(".")[0] for n in top if isinstance(n, ast.Import) for a in n.names}
    names |= {n.module.split(".")[0] for n in top if isinstance(n, ast.ImportFrom) and n.module}
    assert not names & {"torch", "phovo_amd", "oracle"}, names
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    first_import = min(n.lineno for n in ast.walk(main) if isinstance(n, (ast.Import, ast.ImportFrom)))
    launch = min(n.lineno for n in ast.walk(main)
                 if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "launch_ranks")
    assert launch < first_import
    lr = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "launch_ranks")
    src = ast.get_source_segment(open(BENCH).read(), lr)
    assert "os.exec" not in src and "execv" not in src          # children, never a replaced process
    assert "Popen" in src


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_gloo_ranks_share_the_gpu_and_print_one_contract_line():
    """`python bench.py --gpus 2` with no launcher around it, on a one-GPU box (gloo instead of RCCL, which refuses two
    ranks on one device): the parent starts both ranks, each aligns its own shard on the shared card, one all_gather per
    step, rank 0 prints ONE JSON line with the driver's contract keys; value = pairs of both ranks / slowest rank's time."""
    import json
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "256", "--distinct", "8", "--no-cpu-baseline"],
             {"PHOVO_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert d["config"]["pairs_per_gpu"] == 256 and d["config"]["global_pairs_per_step"] == 512
    assert d["value"] > 0 and d["nonfinite_pairs"] == 0 and d["cpu_baseline"] is None
    assert abs(d["value"] - 512 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert "rank 0/2 joined the gloo group" in r.stderr and "rank 1/2 joined the gloo group" in r.stderr

"""N > 1 host path on CPU: two processes (gloo) shard a pair list, gather the per-pair states with one
all_gather and chain the trajectory; the result must equal the single-process one bit for bit.
The states are stand-ins (seeded random small motions): the GPU alignment itself is covered by the -m gpu
tests, what is exercised here is sharding, padding of uneven shards, the collective and pose chaining."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import phovo_amd  # noqa: E402,F401
from phovo_amd import distributed, se3  # noqa: E402


def _states(n):
    rs = np.random.RandomState(42)
    return np.concatenate([rs.uniform(-0.03, 0.03, (n, 3)), rs.uniform(-0.015, 0.015, (n, 3))], axis=1)


def test_shard_ranges_cover_and_balance():
    for n in (0, 1, 7, 8, 599, 600, 4096):
        for world in (1, 2, 3, 8):
            rng = [distributed.shard_range(n, world, r) for r in range(world)]
            assert rng[0][0] == 0 and rng[-1][1] == n
            assert all(rng[i][1] == rng[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rng]
            assert max(sizes) - min(sizes) <= 1
    assert distributed.frames_needed(10, 20) == (10, 21)      # one frame of overlap per shard


def _worker(rank, world, port, n_pairs, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = distributed.shard_range(n_pairs, world, rank)
        local = _states(n_pairs)[a:b]                # what this rank's GPU would have produced
        full = distributed.gather_states(local, n_pairs)
        traj = distributed.trajectory_from_states(full)
        np.save(os.path.join(out_dir, f"full_{rank}.npy"), full)
        np.save(os.path.join(out_dir, f"traj_{rank}.npy"), traj)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,n_pairs", [(2, 9), (2, 16),      # uneven and even shards
                                            (8, 601),            # the node: 601 pairs -> one shard of 76, seven of 75
                                            (8, 5)])             # fewer pairs than ranks: three ranks hold nothing
def test_ranks_gather_equals_single_process(tmp_path, world, n_pairs):
    """World sizes 2 and 8 (the node the driver's scaling run uses): shard arithmetic, the padded all_gather_into_tensor of
    uneven shards -- empty ones included -- and the pose chain, on every rank, against the single-process result."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_pairs, str(tmp_path)), nprocs=world, join=True)
    ref = _states(n_pairs)
    ref_traj = distributed.trajectory_from_states(ref)
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"full_{r}.npy"), ref)
        np.testing.assert_array_equal(np.load(tmp_path / f"traj_{r}.npy"), ref_traj)
    sizes = [distributed.shard_range(n_pairs, world, r) for r in range(world)]
    assert sum(b - a for a, b in sizes) == n_pairs
    if (world, n_pairs) == (8, 601):
        assert [b - a for a, b in sizes] == [76] + [75] * 7
        assert distributed.frames_needed(*sizes[7]) == (526, 602)      # the last rank needs frames 526 .. 601


def test_trajectory_chain_and_tum_format():
    st = _states(5)
    traj = distributed.trajectory_from_states(st)
    pose = np.eye(4)
    for k in range(5):                                # pose *= Rt^-1  (...VisualOdometry.cpp:233-234)
        pose = pose @ np.linalg.inv(se3.eigen_pose(st[k]))
        np.testing.assert_allclose(traj[k], pose, atol=1e-15)
    txt = distributed.format_trajectory([1305031102.175304 + k for k in range(5)], traj)
    lines = txt.strip().split("\n")
    assert lines[0] == "# estimated trajectory" and lines[1] == "# timestamp tx ty tz qx qy qz qw"
    f = [float(x) for x in lines[2].split()]
    assert len(f) == 8 and abs(np.linalg.norm(f[4:]) - 1.0) < 1e-12
    assert lines[2].split()[0] == "1305031102.175304"   # 16 significant digits keep the TUM timestamp
    q = se3.rotation_to_quaternion(traj[0][:3, :3])
    np.testing.assert_allclose(f[4:], q, atol=1e-15)


def test_se3_log_round_trip():
    rs = np.random.RandomState(1)
    for _ in range(10):
        s = np.concatenate([rs.uniform(-0.1, 0.1, 3), rs.uniform(-0.3, 0.3, 3)])
        T = se3.eigen_pose(s)
        assert se3.pose_distance(T, T) < 1e-15
        d = se3.pose_distance(np.eye(4), T)
        assert 0 < d < 1.0
        assert abs(se3.pose_distance(T, np.eye(4)) - d) < 1e-12

"""Sharding of independent frame pairs across the GPUs of one node.

The reference is single-process and single-threaded; what makes the path shard is in its VO app:
the initial state is reset to zero for every pair and pyramids are rebuilt per pair
(apps/PhotoconsistencyVisualOdometry/PhotoconsistencyVisualOdometry.cpp:175,222-224), so pair t
depends on nothing but frames t-1 and t.  Only the pose chain `pose *= Rt^-1` (:233-234) is sequential,
and it is a 4x4 product per pair on the host.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests):
  * rank r aligns the contiguous pair range shard_range(P, world, r) -- no data-path collective,
  * ONE all_gather of the per-pair state vectors (6 fp64 each, padded to the largest shard) moves the
    result to every rank; rank 0 chains the trajectory.
"""
import numpy as np

from . import se3


def shard_range(n_pairs, world_size, rank):
    """Contiguous, balanced split: the first (n_pairs % world_size) ranks get one extra pair."""
    base, extra = divmod(int(n_pairs), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def frames_needed(pair_start, pair_stop):
    """Pair t aligns frame t (source) with frame t+1 (target): a shard needs one frame of overlap."""
    if pair_stop <= pair_start:
        return pair_start, pair_start
    return pair_start, pair_stop + 1


class _DeviceBuffer:
    """A [n, 6] fp64 array that already lives on the GPU (the engine's result buffer), described the way
    torch.as_tensor understands (__cuda_array_interface__), so that the collective can start from it without a trip
    through host memory."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n), 6), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_states_tensor(ptr, n, device):
    """Zero-copy torch view of the engine's device-resident states (phovo_engine_results_device_ptr).  The engine's
    stream must have been synchronised; the view is valid until the next enqueue."""
    import torch
    return torch.as_tensor(_DeviceBuffer(ptr, n), device=device)


def gather_states(local_states, n_pairs_total, device=None, group=None):
    """all_gather of the per-rank [p_r, 6] state blocks -> [n_pairs_total, 6] on every rank.

    local_states: numpy [p_r, 6] (or a torch tensor already on `device`).  Shards may be uneven:
    blocks are padded to the largest shard and trimmed after the collective."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if isinstance(local_states, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(local_states, dtype=np.float64))
        if device is not None:
            t = t.to(device)
    else:
        t = local_states
    if not dist.is_initialized():
        return t.detach().cpu().numpy().reshape(-1, 6)[:n_pairs_total]
    sizes = [shard_range(n_pairs_total, world, r) for r in range(world)]
    max_p = max(b - a for a, b in sizes)
    pad = torch.zeros((max_p, 6), dtype=torch.float64, device=t.device)
    a, b = sizes[rank]
    if t.shape[0] != b - a:
        raise ValueError(f"rank {rank} holds {t.shape[0]} pairs, its shard has {b - a}")
    pad[: b - a] = t.reshape(-1, 6)
    # one collective into one tensor and ONE copy back to the host (eight small device-to-host copies per step would
    # cost more than the all_gather itself)
    big = torch.empty((world, max_p, 6), dtype=torch.float64, device=t.device)
    try:
        dist.all_gather_into_tensor(big, pad, group=group)
    except (RuntimeError, NotImplementedError, AttributeError):      # a backend without the tensor form
        out = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(out, pad, group=group)
        big = torch.stack(out)
    host = big.cpu().numpy()
    return np.concatenate([host[r, : (sb - sa)] for r, (sa, sb) in enumerate(sizes)], axis=0)


def trajectory_from_states(states):
    """Global poses from per-pair states, as the VO app accumulates them (:233-234)."""
    rts = np.stack([se3.eigen_pose(s) for s in states]) if len(states) else np.zeros((0, 4, 4))
    return se3.chain_trajectory(rts)


def format_trajectory(timestamps, poses):
    """TUM trajectory lines `timestamp tx ty tz qx qy qz qw` with 16 significant digits
    (...VisualOdometry.cpp:187-188,240-243: setprecision(digits10 + 1))."""
    lines = ["# estimated trajectory", "# timestamp tx ty tz qx qy qz qw"]
    for ts, T in zip(timestamps, poses):
        q = se3.rotation_to_quaternion(T[:3, :3])
        vals = [ts, T[0, 3], T[1, 3], T[2, 3], q[0], q[1], q[2], q[3]]
        lines.append(" ".join(f"{v:.16g}" for v in vals))
    return "\n".join(lines) + "\n"

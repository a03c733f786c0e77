#!/usr/bin/env python3
"""PhotoconsistencyVisualOdometry over the GPUs of one node: the frame pairs of ONE sequence are sharded across the
ranks, one RCCL all_gather brings the 6-vector results together, rank 0 writes the trajectory
(photoconsistency-visual-odometry_amd/sequence.py; reference loop:
apps/PhotoconsistencyVisualOdometry/PhotoconsistencyVisualOdometry.cpp:175,208-243).

    PhotoconsistencyVisualOdometrySharded.py <config_file.yml> <rgbd_dataset_directory> <output_trajectory_file>
                                             [--ranks N] [--backend nccl|gloo]

Without --ranks (or under a launcher that has set RANK / WORLD_SIZE) this process is one rank.  With --ranks N > 1
and no launcher it starts N fresh rank processes -- before importing anything that could initialise the GPU -- and
exits with their status.  The output is byte-identical to `PhotoconsistencyVisualOdometry ... --batch` on one GPU."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser(usage="%(prog)s <config_file.yml> <rgbd_dataset_directory> <output_trajectory_file> "
                                       "[--ranks N] [--backend nccl|gloo]")
    ap.add_argument("config_file")
    ap.add_argument("dataset_dir")
    ap.add_argument("trajectory_file")
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None,
                    help="nccl = RCCL, one GPU per rank (default); gloo = rehearsal, ranks share the visible GPUs")
    args = ap.parse_args()
    for path, what in ((args.config_file, "Input config file"), (args.dataset_dir, "Input RGBD dataset directory"),
                       (os.path.join(args.dataset_dir, "rgb.txt"), "Input RGB data file"),
                       (os.path.join(args.dataset_dir, "depth.txt"), "Input depth data file")):
        if not os.path.exists(path):
            print(f"{what} {path} does not exist", file=sys.stderr)
            return 1
    if args.ranks > 1 and "RANK" not in os.environ:
        import socket
        import subprocess
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.ranks}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"sequence: starting {args.ranks} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
        return subprocess.call(cmd, env=env)           # children, never an exec; this parent never touches the GPU
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import phovo_amd  # noqa: F401
    from phovo_amd import sequence
    return sequence.run(args.config_file, args.dataset_dir, args.trajectory_file, backend=args.backend)


if __name__ == "__main__":
    sys.exit(main())

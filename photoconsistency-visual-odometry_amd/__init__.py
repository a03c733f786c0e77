"""MI355X-native analytic Gauss-Newton RGB-D frame alignment (phovo hot path).

The directory name carries a hyphen (it mirrors the upstream repository name), so it
is imported through the `phovo_amd` alias module at the repository root.
The compute path is the HIP library csrc/ -> libphovo_hip.so, reached through the
C ABI declared in include/phovo_hip.h; this package is the thin Python host mirror
of the reference's class surface used by tests and bench.
"""
from . import se3, synthetic  # noqa: F401

__all__ = ["se3", "synthetic", "odometry", "native"]


def __getattr__(name):
    if name in ("odometry", "native", "distributed", "sequence"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)

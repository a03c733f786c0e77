"""Opt-in to a diagnostic build of the library for tools/ entry points (never for tests, bench.py or the apps).

    import _variant; _variant.use("timeline")      # BEFORE the package is imported

builds csrc/build_<name>/libphovo_hip_<name>.so (csrc/Makefile, `make <name>`) if it is missing and makes native.py load it:
native.py ignores PHOVO_HIP_LIBRARY unless PHOVO_TOOLS_LIBRARY_OPT_IN=tools is set, which only this module does.  A path
to any other build of the library (an A/B against an earlier commit) goes through use_path()."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "photoconsistency-visual-odometry_amd", "csrc")


def use_path(path):
    if "phovo_amd" in sys.modules:
        raise RuntimeError("choose the library before importing the package")
    os.environ["PHOVO_TOOLS_LIBRARY_OPT_IN"] = "tools"
    os.environ["PHOVO_HIP_LIBRARY"] = os.path.abspath(path)
    return os.environ["PHOVO_HIP_LIBRARY"]


def use(name):
    so = os.path.join(CSRC, f"build_{name}", f"libphovo_hip_{name}.so")
    subprocess.check_call(["make", "-s", "-C", CSRC, name])       # (up to date: a no-op)
    return use_path(so)


def use_from_environment():
    """For the A/B shell scripts: PHOVO_TOOLS_LIBRARY=<path> (set by tools/ab_*.sh) selects library B."""
    path = os.environ.get("PHOVO_TOOLS_LIBRARY")
    return use_path(path) if path else None

#!/usr/bin/env python3
"""bench.py on a named diagnostic build or on another build of the library (A/B runs):
    python3 tools/bench_with.py tuning|timeline|stamps|<path to a libphovo_hip.so> [bench.py arguments]
bench.py itself always loads the product library; this entry point is the opt-in (tools/_variant.py)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: E402

which = sys.argv[1]
sys.argv = [os.path.join(_variant.ROOT, "bench.py")] + sys.argv[2:]
if os.sep in which or which.endswith(".so"):
    _variant.use_path(which)
else:
    _variant.use(which)
runpy.run_path(sys.argv[0], run_name="__main__")

"""Import alias: `import phovo_amd` loads the package in
`photoconsistency-visual-odometry_amd/` (a hyphen cannot appear in an import statement)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "photoconsistency-visual-odometry_amd")
_spec = importlib.util.spec_from_file_location(
    "phovo_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["phovo_amd"] = _mod
_spec.loader.exec_module(_mod)

"""The C-ABI library loads here (no GPU) and exports every symbol include/phovo_hip.h declares;
host-only entry points (configuration, eigenPose, error reporting) behave like the reference's.
No compute is called: the product path has no CPU fallback, which is asserted too."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import phovo_amd  # noqa: F401
from phovo_amd import native, se3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_DIR = os.path.join(ROOT, "config_files")


def test_header_and_binding_declare_the_same_symbols():
    hdr = open(os.path.join(ROOT, "include", "phovo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(phovo_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(native.SYMBOLS), declared ^ set(native.SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = native.lib()                       # builds with hipcc if missing; raises on failure
    for name in native.SYMBOLS:
        assert hasattr(L, name), name
    assert b"gfx950" in L.phovo_version()


def test_defaults_match_reference_constructor():
    cfg = native.make_config()             # ...Analytic.h:430-443
    assert cfg.num_levels == 5
    assert list(cfg.max_num_iterations[:5]) == [0, 0, 5, 20, 50]
    assert list(cfg.blur_filter_size[:5]) == [0] * 5
    assert list(cfg.image_gradients_scaling_factor[:5]) == [0.0625] * 5
    assert list(cfg.lambda_optimization_step[:5]) == [1.0] * 5
    assert list(cfg.min_gradient_norm[:5]) == [300.0] * 5
    assert cfg.visualize_iterations == 0


@pytest.mark.parametrize("name,levels,max_iter,min_grad,vis", [
    ("config_4_level_optimization_analytic.yml", 4, [0, 0, 20, 50], [300] * 4, 0),
    ("config_5_level_optimization_analytic.yml", 5, [0, 0, 5, 20, 50], [300] * 5, 0),
    ("config_6_level_optimization_analytic.yml", 6, [0, 0, 5, 20, 50, 50], [100] * 5 + [10], 0),
    ("config_only_level_0_analytic.yml", 1, [5000], [300], 1),
])
def test_reference_yml_files_parse_unchanged(name, levels, max_iter, min_grad, vis):
    cfg = native.read_config_file(os.path.join(CFG_DIR, name))
    assert cfg.num_levels == levels
    assert list(cfg.max_num_iterations[:levels]) == max_iter
    assert list(cfg.min_gradient_norm[:levels]) == [float(v) for v in min_grad]
    assert list(cfg.image_gradients_scaling_factor[:levels]) == [0.0625] * levels
    assert list(cfg.lambda_optimization_step[:levels]) == [1.0] * levels
    assert list(cfg.blur_filter_size[:levels]) == [0] * levels
    assert cfg.visualize_iterations == vis


def test_yml_errors_are_reported_not_swallowed(tmp_path):
    with pytest.raises(native.PhovoError) as ei:
        native.read_config_file(str(tmp_path / "missing.yml"))
    assert ei.value.status == 6            # PHOVO_E_IO
    p = tmp_path / "short.yml"
    p.write_text("%YAML:1.0\nnumOptimizationLevels: 3\n"
                 "blurFilterSize (at each level): [0, 0]\n"
                 "imageGradientsScalingFactor (at each level): [0.0625, 0.0625, 0.0625]\n"
                 "lambda_optimization_step (at each level): [1,1,1]\n"
                 "max_num_iterations (at each level): [0, 5, 10]\n"
                 "min_gradient_norm (at each level): [300,300,300]\nvisualizeIterations: 0\n")
    with pytest.raises(native.PhovoError) as ei:
        native.read_config_file(str(p))
    assert ei.value.status == 2 and "fewer entries" in str(ei.value)
    p2 = tmp_path / "nokey.yml"
    p2.write_text("%YAML:1.0\nnumOptimizationLevels: 1\n")
    with pytest.raises(native.PhovoError) as ei:
        native.read_config_file(str(p2))
    assert ei.value.status == 2 and "missing" in str(ei.value)


def test_yml_multiline_sequence_and_real_valued_ints(tmp_path):
    p = tmp_path / "ml.yml"
    p.write_text("%YAML:1.0\nnumOptimizationLevels: 2\n"
                 "blurFilterSize (at each level): [0,\n   3]\n"
                 "imageGradientsScalingFactor (at each level): [0.0625, 0.125]\n"
                 "lambda_optimization_step (at each level): [1, 0.5]\n"
                 "max_num_iterations (at each level): [ 7., 9 ]\n"
                 "min_gradient_norm (at each level): [1e2, 3.5e1]\nvisualizeIterations: 1\n")
    cfg = native.read_config_file(str(p))
    assert list(cfg.blur_filter_size[:2]) == [0, 3]
    assert list(cfg.max_num_iterations[:2]) == [7, 9]
    assert list(cfg.min_gradient_norm[:2]) == [100.0, 35.0]
    assert list(cfg.lambda_optimization_step[:2]) == [1.0, 0.5]


def test_eigen_pose_matches_restatement():
    L = native.lib()
    rs = np.random.RandomState(0)
    for _ in range(5):
        s = rs.uniform(-1, 1, 6)
        rt = np.zeros(16)
        dp = C.POINTER(C.c_double)
        assert L.phovo_eigen_pose(s.ctypes.data_as(dp), rt.ctypes.data_as(dp)) == 0
        np.testing.assert_allclose(rt.reshape(4, 4), se3.eigen_pose(s), rtol=0, atol=1e-15)
        R = rt.reshape(4, 4)[:3, :3]
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-14)


def test_no_cpu_fallback_without_a_gpu():
    L = native.lib()
    if L.phovo_device_count() > 0:
        pytest.skip("a HIP device is present")
    h = C.c_void_p()
    st = L.phovo_engine_create(0, C.byref(h))
    assert st == 4 and not h.value         # PHOVO_E_HIP, loudly
    assert b"no CPU path" in L.phovo_last_error()
    from phovo_amd import odometry
    with pytest.raises(native.PhovoError):
        odometry.CPhotoconsistencyOdometryAnalytic()


def test_null_arguments_are_rejected():
    L = native.lib()
    assert L.phovo_config_default(None) == 1
    assert L.phovo_eigen_pose(None, None) == 1
    assert L.phovo_engine_destroy(None) == 0
    assert L.phovo_odometry_optimize(None) == 1

"""Native (non-Python) clients of the boundary, CPU only:
  * a C11 program compiled with gcc against include/phovo_hip.h and linked to libphovo_hip.so,
  * AddressSanitizer / UBSan builds of the host-side parsers (yml reader, PNG codec) on good and hostile input
    (GPU sanitizers are not available on the pool; the CPU build is what can be sanitised)."""
import os
import struct
import subprocess
import zlib

import pytest

import phovo_amd  # noqa: F401
from phovo_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "config_files")
PKG = os.path.dirname(native.library_path())


def test_header_is_valid_c_and_c_client_runs(tmp_path):
    native.lib()
    exe = tmp_path / "cabi_c_client"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "cabi_c_client.c"), "-o", str(exe),
                           "-L", PKG, "-lphovo_hip", "-lm", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe), os.path.join(CFG, "config_4_level_optimization_analytic.yml")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cabi_c_client ok" in r.stdout


@pytest.mark.gpu
def test_c_client_aligns_a_golden_pair_on_the_gpu(tmp_path):
    """A program written in plain C drives the whole path on the device -- phovo_odometry_create ... set_source_frame,
    set_target_frame, optimize, get_* in the reference's call order (...FrameAlignment.cpp:92-105) -- on the inputs of
    tests/golden/case_a.npz and lands within 1e-9 of the fixture's expected state (the numpy twin's), iteration counts
    equal."""
    import numpy as np
    native.lib()
    exe = tmp_path / "cabi_c_client"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "cabi_c_client.c"), "-o", str(exe),
                           "-L", PKG, "-lphovo_hip", "-lm", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib"])
    d = np.load(os.path.join(ROOT, "tests", "golden", "case_a.npz"))
    nl = int(d["num_levels"])
    h, w = d["gray0"].shape
    blob = [np.array([w, h, nl], dtype=np.int32).tobytes(), np.ascontiguousarray(d["K"], dtype=np.float64).tobytes(),
            np.array([float(d["min_depth"]), float(d["max_depth"])]).tobytes()]
    for l in range(nl):
        blob.append(np.array([int(d["max_iter"][l])], dtype=np.int32).tobytes())
        blob.append(np.array([d["min_grad"][l], d["lam"][l], d["grad_scale"][l]], dtype=np.float64).tobytes())
    blob += [np.ascontiguousarray(d["gray0"], dtype=np.uint8).tobytes(), np.ascontiguousarray(d["depth0"], dtype=np.float64).tobytes(),
             np.ascontiguousarray(d["gray1"], dtype=np.uint8).tobytes(),
             np.ascontiguousarray(d["init_state"], dtype=np.float64).tobytes(),
             np.ascontiguousarray(d["exp_state"], dtype=np.float64).tobytes(),
             np.ascontiguousarray(d["exp_iters"][:nl], dtype=np.int32).tobytes()]
    problem = tmp_path / "case_a.bin"
    problem.write_bytes(b"".join(blob))
    r = subprocess.run([str(exe), os.path.join(CFG, "config_4_level_optimization_analytic.yml"), str(problem)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gpu alignment from C" in r.stdout and "cabi_c_client ok" in r.stdout
    print(r.stdout)


def test_yml_reader_under_asan_ubsan(tmp_path):
    exe = tmp_path / "yml_asan"
    csrc = os.path.join(ROOT, "photoconsistency-visual-odometry_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", csrc, "-I", "/opt/rocm/include",
                           os.path.join(csrc, "yml_config.cpp"), os.path.join(ROOT, "tests", "native", "yml_asan_driver.cpp"),
                           "-o", str(exe)])
    good = [os.path.join(CFG, f) for f in sorted(os.listdir(CFG))]
    r = subprocess.run([str(exe)] + good, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    hostile = {
        "empty.yml": "",
        "nolevels.yml": "%YAML:1.0\nblurFilterSize (at each level): [0]\n",
        "huge.yml": "%YAML:1.0\nnumOptimizationLevels: 999999999999\n",
        "unterminated.yml": "%YAML:1.0\nnumOptimizationLevels: 2\nblurFilterSize (at each level): [0, 0\n",
        "garbage.yml": "%YAML:1.0\nnumOptimizationLevels: two\n: : :\n[[[[\n" + "x" * 100000 + "\n",
        "long_seq.yml": "%YAML:1.0\nnumOptimizationLevels: 2\nblurFilterSize (at each level): [" + ",".join(["1"] * 5000) + "]\n"
                        "imageGradientsScalingFactor (at each level): [1,1]\nlambda_optimization_step (at each level): [1,1]\n"
                        "max_num_iterations (at each level): [1,1]\nmin_gradient_norm (at each level): [1,1]\nvisualizeIterations: 0\n",
        "nan.yml": "%YAML:1.0\nnumOptimizationLevels: nan\n",
    }
    paths = []
    for name, text in hostile.items():
        (tmp_path / name).write_text(text)
        paths.append(str(tmp_path / name))
    r = subprocess.run([str(exe)] + paths, capture_output=True, text=True, timeout=120)
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
    # every hostile file is rejected with a status, except the over-long sequence, which is legal
    lines = [l for l in r.stdout.splitlines() if "->" in l]
    status = {os.path.basename(l.split(" -> ")[0]): int(l.split(" -> ")[1].split()[0]) for l in lines}
    assert status["long_seq.yml"] == 0
    assert all(v != 0 for k, v in status.items() if k != "long_seq.yml"), status


def test_png_codec_under_asan_ubsan(tmp_path):
    exe = tmp_path / "png_asan"
    apps = os.path.join(ROOT, "apps")
    subprocess.check_call(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", apps, os.path.join(apps, "tools", "png_probe.cpp"), os.path.join(apps, "io", "png_io.cpp"),
                           "-o", str(exe), "-lz"])

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    sig = b"\x89PNG\r\n\x1a\n"
    ihdr = lambda w, h, bd, ct, il=0: chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bd, ct, 0, 0, il))
    good_raw = b"".join(b"\x00" + bytes(range(8)) for _ in range(4))
    files = {
        "ok.png": sig + ihdr(8, 4, 8, 0) + chunk(b"IDAT", zlib.compress(good_raw)) + chunk(b"IEND", b""),
        "truncated_idat.png": sig + ihdr(8, 4, 8, 0) + chunk(b"IDAT", zlib.compress(good_raw)[:5]) + chunk(b"IEND", b""),
        "short_data.png": sig + ihdr(64, 64, 8, 0) + chunk(b"IDAT", zlib.compress(good_raw)) + chunk(b"IEND", b""),
        "huge_dims.png": sig + ihdr(0x7fffffff, 2, 16, 6) + chunk(b"IDAT", zlib.compress(b"\x00")) + chunk(b"IEND", b""),
        "bad_filter.png": sig + ihdr(8, 1, 8, 0) + chunk(b"IDAT", zlib.compress(b"\x09" + bytes(8))) + chunk(b"IEND", b""),
        "interlaced.png": sig + ihdr(8, 4, 8, 0, 1) + chunk(b"IDAT", zlib.compress(good_raw)) + chunk(b"IEND", b""),
        "chunk_overrun.png": sig + struct.pack(">I", 0xfffffff0) + b"IDAT" + b"\x00" * 16,
        "palette_oob.png": sig + ihdr(4, 1, 8, 3) + chunk(b"PLTE", b"\x01\x02\x03") + chunk(b"IDAT", zlib.compress(b"\x00\x00\x01\x02\x03")) + chunk(b"IEND", b""),
        "no_ihdr.png": sig + chunk(b"IDAT", zlib.compress(good_raw)) + chunk(b"IEND", b""),
    }
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
        for mode in ("gray8", "raw16"):
            r = subprocess.run([str(exe), mode, str(tmp_path / name), str(tmp_path / "out.raw")],
                               capture_output=True, text=True, timeout=60)
            assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (name, mode, r.stderr[-2000:])
            if name == "ok.png":
                assert r.returncode == 0, r.stderr
            else:
                assert r.returncode == 1, (name, mode, r.returncode, r.stderr)


def test_shards_vote_before_the_collective(tmp_path):
    """apps/rccl/shard_vote.h, the vote the shard threads of PhotoconsistencyVisualOdometry --batch --gpus N --rccl take before
    the all_gather: with a blocking barrier standing in for the collective, a failing shard must keep EVERY shard out of it
    (nobody waits for a rank that never comes) and every thread must return -- the program ends, within the timeout."""
    exe = tmp_path / "shard_vote_test"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread", "-I", os.path.join(ROOT, "apps"),
                           os.path.join(ROOT, "tests", "native", "shard_vote_test.cpp"), "-o", str(exe)])
    for shards, failing, expect in ((8, -1, "entered=8 told_ok=8 failed=0"), (8, 3, "entered=0 told_ok=0 failed=1"),
                                    (2, 0, "entered=0 told_ok=0 failed=1"), (1, -1, "entered=1 told_ok=1 failed=0")):
        r = subprocess.run([str(exe), str(shards), str(failing)], capture_output=True, text=True, timeout=20)
        assert r.returncode == 0 and r.stdout.strip() == expect, (shards, failing, r.stdout, r.stderr)

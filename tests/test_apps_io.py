"""apps/io PNG codec (what cv::imread / cv::imwrite do for the reference's apps) against PIL and against
hand-encoded PNGs that exercise all five scanline filters.  CPU only."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "apps", "bin", "png_probe")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "apps"), os.path.join(ROOT, "apps", "bin", "png_probe")])


def _probe(mode, src, dst):
    subprocess.check_call([PROBE, mode, str(src), str(dst)])


def _read_raw(path, dtype):
    with open(path, "rb") as f:
        w, h = [int(v) for v in f.readline().split()]
        return np.frombuffer(f.read(), dtype=dtype).reshape(h, w)


def _encode_png(rows_be, width, height, bit_depth, color_type, filters):
    """Minimal PNG encoder with an explicit filter type per scanline (PNG spec section 9)."""
    bpp = max(1, {0: 1, 2: 3, 4: 2, 6: 4}[color_type] * bit_depth // 8)
    raw = bytearray()
    prev = bytes(len(rows_be[0]))
    for y, row in enumerate(rows_be):
        ft = filters[y % len(filters)]
        out = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = b
            elif ft == 3:
                pred = (a + b) // 2
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (v - pred) & 0xFF
        raw.append(ft)
        raw += out
        prev = row

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    comp = zlib.compress(bytes(raw), 9)
    half = len(comp) // 2          # two IDAT chunks: the decoder must concatenate them
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, bit_depth, color_type, 0, 0, 0))
            + chunk(b"tEXt", b"k\x00v") + chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b""))


def test_gray16_depth_all_filters(tmp_path):
    rs = np.random.RandomState(0)
    d = rs.randint(0, 65536, (37, 53)).astype(np.uint16)
    rows = [row.astype(">u2").tobytes() for row in d]
    p = tmp_path / "d.png"
    p.write_bytes(_encode_png(rows, 53, 37, 16, 0, [0, 1, 2, 3, 4]))
    np.testing.assert_array_equal(np.array(Image.open(p)), d)          # the hand encoder itself is right
    _probe("raw16", p, tmp_path / "d.raw")
    np.testing.assert_array_equal(_read_raw(tmp_path / "d.raw", np.uint16), d)


def test_rgb8_to_gray_matches_libpng_coefficients(tmp_path):
    rs = np.random.RandomState(1)
    rgb = rs.randint(0, 256, (29, 41, 3)).astype(np.uint8)
    rows = [row.tobytes() for row in rgb]
    p = tmp_path / "c.png"
    p.write_bytes(_encode_png(rows, 41, 29, 8, 2, [4, 3, 1, 2, 0]))
    np.testing.assert_array_equal(np.array(Image.open(p)), rgb)
    _probe("gray8", p, tmp_path / "c.raw")
    r, g, b = [rgb[..., i].astype(np.int64) for i in range(3)]
    expect = ((9797 * r + 19234 * g + 3737 * b + 16384) >> 15).astype(np.uint8)
    np.testing.assert_array_equal(_read_raw(tmp_path / "c.raw", np.uint8), expect)
    assert np.max(np.abs(expect.astype(int) - (0.299 * r + 0.587 * g + 0.114 * b))) <= 1.0


def test_pil_written_files_and_round_trip(tmp_path):
    rs = np.random.RandomState(2)
    g8 = rs.randint(0, 256, (48, 64)).astype(np.uint8)
    d16 = rs.randint(0, 65536, (48, 64)).astype(np.uint16)
    Image.fromarray(g8).save(tmp_path / "g.png")
    Image.fromarray(d16).save(tmp_path / "d.png")
    Image.fromarray(np.stack([g8, g8, g8, 255 - g8], axis=-1), "RGBA").save(tmp_path / "rgba.png")
    _probe("gray8", tmp_path / "g.png", tmp_path / "g.raw")
    np.testing.assert_array_equal(_read_raw(tmp_path / "g.raw", np.uint8), g8)
    _probe("raw16", tmp_path / "d.png", tmp_path / "d.raw")
    np.testing.assert_array_equal(_read_raw(tmp_path / "d.raw", np.uint16), d16)
    _probe("gray8", tmp_path / "rgba.png", tmp_path / "a.raw")
    np.testing.assert_array_equal(_read_raw(tmp_path / "a.raw", np.uint8), g8)       # R=G=B -> unchanged
    _probe("gray8", tmp_path / "d.png", tmp_path / "d8.raw")                       # imread(.,0) of 16-bit: high byte
    np.testing.assert_array_equal(_read_raw(tmp_path / "d8.raw", np.uint8), (d16 >> 8).astype(np.uint8))
    # our writer -> PIL
    _probe("copy8", tmp_path / "g.png", tmp_path / "g2.png")
    _probe("copy16", tmp_path / "d.png", tmp_path / "d2.png")
    np.testing.assert_array_equal(np.array(Image.open(tmp_path / "g2.png")), g8)
    np.testing.assert_array_equal(np.array(Image.open(tmp_path / "d2.png")), d16)


def test_bad_files_are_reported(tmp_path):
    (tmp_path / "x.png").write_bytes(b"not a png at all")
    r = subprocess.run([PROBE, "gray8", str(tmp_path / "x.png"), str(tmp_path / "o")], capture_output=True)
    assert r.returncode == 1 and b"not a PNG" in r.stderr
    r = subprocess.run([PROBE, "raw16", str(tmp_path / "missing.png"), str(tmp_path / "o")], capture_output=True)
    assert r.returncode == 1 and b"cannot open" in r.stderr

"""CPU suite: the headless host's image loader (path-tracer_amd/host/kl_image_impl.cpp, kl_jpeg_impl.cpp), the role stb_image
plays behind Image::f_load in the reference (source/I-O/File.cpp:13-29). Texels feed the shader through a NEAREST lookup, so the
loader has to give the bytes stb_image gives: checked on the reference's own JPEG texture ("resources/Texture.png") and on a
set of small JPEG files, each against the bytes the reference's vendored stb_image decoded from it (tests/golden/make_golden.py
jpeg). Runs the compiled loader through `kl_bench --decode`, which touches no GPU.

Not covered by a known answer: 1x2 sampling (4:4:0), which this Pillow cannot write."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

BENCH = os.path.join(ROOT, "path-tracer_amd", "bin", "kl_bench")
CASES = np.load(os.path.join(GOLDEN, "jpeg_stb.npz"))
NAMES = sorted(k[:-6] for k in CASES.files if k.endswith("__jpeg"))
PNG_CASES = np.load(os.path.join(GOLDEN, "png_stb.npz"))
PNG_NAMES = sorted(k[:-5] for k in PNG_CASES.files if k.endswith("__png"))


def decode(path, tmp_path):
    out = str(tmp_path / "pixels.raw")
    res = subprocess.run([BENCH, "--decode", str(path), out], capture_output=True, text=True, timeout=60)
    if res.returncode != 0:
        assert res.returncode == 1 and "cannot decode" in res.stderr, (res.returncode, res.stderr)    # refused, not crashed
        return None
    w, h, c = map(int, res.stdout.split())
    return np.fromfile(out, dtype=np.uint8).reshape(h, w, c)


def test_reference_texture_decodes_to_the_bytes_stb_image_gives(tmp_path):
    want = np.load(os.path.join(GOLDEN, "assets", "Texture_stb_rgb8.npz"))["rgb"]
    got = decode(os.path.join(GOLDEN, "assets", "Texture.png"), tmp_path)          # JPEG data under a .png name
    assert got is not None and got.shape == (1080, 1920, 3)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name", NAMES)
def test_jpeg_cases_match_stb_image(name, tmp_path):
    jpg = tmp_path / (name + ".jpg")
    jpg.write_bytes(CASES[name + "__jpeg"].tobytes())
    got = decode(jpg, tmp_path)
    assert got is not None
    assert np.array_equal(got, CASES[name + "__stb"])


@pytest.mark.parametrize("name", PNG_NAMES)
def test_png_kinds_match_stb_image(name, tmp_path):
    """grey 1 / 8 / 16 bit, grey + alpha, RGB, RGBA, palettes of 1 / 2 / 4 / 8 bits, palette and colour-key transparency: the
    channel count and the bytes stb_image reduces each file to"""
    png = tmp_path / (name + ".png")
    png.write_bytes(PNG_CASES[name + "__png"].tobytes())
    got = decode(png, tmp_path)
    assert got is not None
    assert got.shape == PNG_CASES[name + "__stb"].shape and np.array_equal(got, PNG_CASES[name + "__stb"])


def test_png_still_goes_through_the_png_reader(tmp_path):
    from PIL import Image
    got = decode(os.path.join(GOLDEN, "assets", "Background.png"), tmp_path)
    assert np.array_equal(got, np.array(Image.open(os.path.join(GOLDEN, "assets", "Background.png"))))


def test_unsupported_and_damaged_files_are_refused_cleanly(tmp_path):
    import io
    from PIL import Image
    rng = np.random.default_rng(3)
    img = Image.fromarray(rng.integers(0, 256, (24, 40, 3), dtype=np.uint8))
    buf = io.BytesIO()
    img.save(buf, "JPEG", progressive=True)                      # progressive frames are not handled
    p = tmp_path / "progressive.jpg"
    p.write_bytes(buf.getvalue())
    assert decode(p, tmp_path) is None
    buf = io.BytesIO()
    img.convert("CMYK").save(buf, "JPEG")                        # nor four components
    p.write_bytes(buf.getvalue())
    assert decode(p, tmp_path) is None
    good = CASES["s420_45x31__jpeg"].tobytes()
    for cut in (0, 2, 3, 20, 200, len(good) // 2, len(good) - 2):
        p.write_bytes(good[:cut])
        got = decode(p, tmp_path)                                # a cut inside the entropy-coded data still gives an image, as stb does
        assert got is None or got.shape == (31, 45, 3)
    # a frame header that announces 65535 x 65535 pixels in a 90-byte file
    p.write_bytes(bytes.fromhex("ffd8ffc0000b08ffffffff01011100ffda0008010100003f00") + bytes(64))
    assert decode(p, tmp_path) is None
    p.write_bytes(b"not an image at all")
    assert decode(p, tmp_path) is None

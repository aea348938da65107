"""GPU suite: the compiled headless host (path-tracer_amd/host: GLSL_Renderer-shaped class, PNG codec,
kl_bench CLI) end to end against the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

import kloracle as ko
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
BENCH = os.path.join(ROOT, "path-tracer_amd", "bin", "kl_bench")


def test_kl_bench_renders_the_builtin_scene(tmp_path, oracle_builtin, textures):
    assert os.path.exists(BENCH), "build the host tool first (make tools)"
    out = str(tmp_path / "frame.png")
    res = subprocess.run([BENCH, "--scene", "builtin", "--mode", "pt", "--w", "64", "--h", "36", "--frames", "3", "--warmup", "0",
                          "--hdri", os.path.join(GOLDEN, "assets", "Background.png"),
                          "--albedo", os.path.join(GOLDEN, "assets", "Texture.png"), "--out", out],     # JPEG data, as the reference ships it
                         check=True, capture_output=True, text=True, timeout=120)
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["frames"] == 3 and info["samples_per_s"] > 0 and 3 < info["rays_per_sample"] < 9
    from PIL import Image
    got = np.array(Image.open(out))
    p = ko.default_params(width=64, height=36, mode=ko.MODE_PT, max_frames=0)
    acc = ko.render_mean(oracle_builtin, ko.default_camera(), p, 3)
    want = ko.resolve(acc)[::-1]                 # PNG rows run top to bottom, buffer rows bottom to top
    assert np.array_equal(got, want)             # exercises the C++ PNG and JPEG readers (textures) and the PNG writer


def test_host_session_follows_the_reference_reset_rules():
    res = subprocess.run([BENCH, "--interactive-demo", "--w", "48", "--h", "27", "--frames", "5"],
                         check=True, capture_output=True, text=True, timeout=120)
    first, second = [json.loads(l) for l in res.stdout.strip().splitlines()]
    assert first == {"iFrame": 5, "mode": 1}
    # V switches PATHTRACED -> ZBUFFER and zeroes iFrame; a frame with D held moves the camera by the
    # move sensitivity (0.15) and restarts again (that frame is iFrame 0); two more frames follow
    assert second["mode"] == 2 and second["iFrame"] == 3 and abs(second["camera_x"] - 0.15) < 1e-9


def test_kl_bench_reproduces_the_reference_screenshot(tmp_path):
    """the whole host side in one line of kl_bench: its JPEG and PNG readers, Camera::f_compile from position / yaw / pitch, the
    shipped accumulation (SAMPLES freeze, 8-bit attachments), kl_resolve and the PNG writer, against the upstream author's
    path-traced screenshot (the picture shows the image's upper 1974 rows)"""
    from PIL import Image
    out = str(tmp_path / "s4.png")
    subprocess.run([BENCH, "--scene", "builtin", "--mode", "pt", "--w", "3840", "--h", "2019", "--frames", "40", "--warmup", "0", "--shipped",
                    "--pos", "10.58809", "4.05715", "29.47826", "--yaw", "-112.42518", "--pitch", "-0.82758",
                    "--hdri", os.path.join(GOLDEN, "assets", "Background.png"), "--albedo", os.path.join(GOLDEN, "assets", "Texture.png"),
                    "--out", out], check=True, capture_output=True, text=True, timeout=180)
    got = np.array(Image.open(out).convert("RGB"))[:1974].astype(np.int32)               # PNG rows run top to bottom
    shot = np.array(Image.open(os.path.join(GOLDEN, "assets", "S4.png")).convert("RGB"))[45:].astype(np.int32)
    d = np.abs(got - shot).max(axis=2)
    assert (d == 0).mean() >= 0.86 and (d <= 1).mean() >= 0.97 and (d <= 8).mean() >= 0.996 and np.abs(got - shot).mean() <= 0.16


def test_kl_bench_drives_a_group_of_gpus(tmp_path):
    """the headless tool with --gpus: one process, kl_group_*, tiles over RCCL with the one GPU of the box, and three members
    sharing it through the copy exchange; same image mean as the single-context run"""
    import json
    import subprocess
    exe = os.path.join(ROOT, "path-tracer_amd", "bin", "kl_bench")
    base = [exe, "--scene", "builtin", "--w", "96", "--h", "50", "--frames", "5", "--warmup", "0"]

    def run(extra):
        out = subprocess.run(base + extra, check=True, stdout=subprocess.PIPE, text=True, timeout=120).stdout
        return json.loads(out.strip().splitlines()[-1])

    tiles = run(["--gpus", "1", "--partition", "tiles"])
    copy3 = run(["--gpus", "3", "--devices", "0,0,0", "--partition", "tiles", "--exchange", "copy"])
    samp3 = run(["--gpus", "3", "--devices", "0,0,0", "--partition", "samples", "--exchange", "copy"])
    assert tiles["gpus"] == 1 and copy3["gpus"] == 3
    assert tiles["image_mean"] == copy3["image_mean"]                      # tiles: bit-identical images
    assert abs(samp3["image_mean"] - tiles["image_mean"]) <= 1e-5 * tiles["image_mean"]

import os
import sys

import numpy as np
import pytest

# Load order: PyTorch ships its own copy of the HIP runtime. If libklrender.so (which names the system's /opt/rocm copy) is
# loaded first, PyTorch later binds to that already-loaded copy and cannot initialise ("No HIP GPUs are available"); loaded
# the other way round both work. Tests that hand device memory to the library through torch tensors therefore need torch
# in the process before the library, whatever subset of test files is collected.
try:
    import torch  # noqa: F401
except Exception:          # the C-ABI tests themselves do not need it
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "path-tracer_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def textures():
    """The two images the reference host loads (GLSL_Renderer.cpp:237,240) as 8-bit arrays with rows
    top-to-bottom, exactly as stb_image hands them over: the environment is a lossless PNG; the albedo
    ("Texture.png" is JPEG data) is kept as the bytes stb_image decodes (tests/golden/make_golden.py)."""
    from PIL import Image
    bg = np.array(Image.open(os.path.join(GOLDEN, "assets", "Background.png")))
    tx = np.load(os.path.join(GOLDEN, "assets", "Texture_stb_rgb8.npz"))["rgb"]
    assert bg.shape == (2048, 4096, 4) and tx.shape == (1080, 1920, 3)
    return {"hdri": bg, "albedo": tx}


@pytest.fixture(scope="session")
def oracle_builtin(textures):
    import kloracle as ko
    sc = ko.Scene(builtin=True)
    sc.set_texture(ko.TEX_HDRI, textures["hdri"])
    sc.set_texture(ko.TEX_ALBEDO, textures["albedo"])
    return sc


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def frac_within_rel(a, ref, tol=1e-4, floor=1e-6):
    """fraction of pixels whose worst channel is within tol relative error"""
    d = np.abs(a.astype(np.float64) - ref.astype(np.float64))
    rel = d / np.maximum(np.abs(ref.astype(np.float64)), floor)
    if rel.ndim == 3:
        rel = rel.max(-1)
    return float(np.mean(rel <= tol))


def mae_clamped(a, ref):
    """mean absolute error of what reaches the screen: PostProcess.glsl:38 writes the
    accumulation texel to a UNORM backbuffer, i.e. clamp(x, 0, 1)"""
    return float(np.mean(np.abs(np.clip(a, 0, 1).astype(np.float64) - np.clip(ref, 0, 1).astype(np.float64))))


def free_port():
    """a TCP port nobody listens on right now (the kernel picks it), for torch.distributed rendezvous on 127.0.0.1"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

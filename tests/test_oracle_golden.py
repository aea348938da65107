"""CPU suite: the oracle (oracle/kl_oracle.c) against golden vectors produced by
executing the reference shaders on llvmpipe (tests/golden/make_golden.py).

Tolerances. ZDepth and AO: per-pixel relative error <= 1e-4 (north-star bar); the
fraction of pixels allowed outside it is an outlier budget for silhouette pixels where
llvmpipe's polynomial sin/cos send the second AO ray to another surface. Path traced:
mean absolute error of clamp(x,0,1) (what PostProcess.glsl:38 puts on screen), because
llvmpipe's atan/acos approximations move ~0.3 % of environment lookups by one texel.
"""
import numpy as np
import pytest

import kloracle as ko
from conftest import frac_within_rel, golden, mae_clamped

# converged-image bounds (4096 spp), shared with the GPU test of the same golden: mean absolute error of
# clamp(x,0,1), largest absolute difference, relative tolerance and the fraction of pixels that must meet it.
# Measured, oracle vs llvmpipe: 3.0e-6, 2.0e-3, 99.73 % (82.8 % within 1e-4, 43 % of values bit-identical).
CONVERGED_MAE, CONVERGED_MAX_ABS, CONVERGED_REL, CONVERGED_FRAC = 1e-5, 5e-3, 1e-3, 0.99

MODES = {"z": ko.MODE_Z, "ao": ko.MODE_AO, "pt": ko.MODE_PT, "ptwhite": ko.MODE_PT}


@pytest.fixture(scope="module")
def white_scene():
    return ko.Scene(builtin=True)


@pytest.mark.parametrize("size", [(64, 36), (160, 90)])
def test_builtin_frames(size, oracle_builtin, white_scene):
    w, h = size
    g = golden("builtin_%dx%d.npz" % (w, h))
    cam = ko.default_camera()
    for key in sorted(g.files):
        name, f = key.split("_f")
        p = ko.default_params(width=w, height=h, mode=MODES[name], max_frames=0)
        scene = white_scene if name == "ptwhite" else oracle_builtin
        out = scene.render_raw(cam, p, int(f))
        assert np.all(out[..., 3] == 1.0)
        out, ref = out[..., :3], g[key]
        bitexact = np.mean(out.view(np.uint32) == ref.view(np.uint32))
        if name == "z":
            assert bitexact == 1.0, key                       # no transcendental on this path
        elif name == "ao":
            assert frac_within_rel(out, ref) >= 0.999, key
            assert bitexact >= 0.97, key
        elif name == "ptwhite":
            assert frac_within_rel(out, ref) >= 0.999, key
        else:
            assert frac_within_rel(out, ref) >= 0.99, key
            assert mae_clamped(out, ref) <= 5e-4, key


def test_config1_16spp_mean(oracle_builtin):
    """BASELINE config 1: bundled scene, 640x360, 16 spp, path traced, 32-bounce cap."""
    ref = golden("config1_pt_640x360_16spp.npz")["acc"]
    p = ko.default_params(width=640, height=360, mode=ko.MODE_PT, max_frames=30)
    cn = ko.Counters()
    acc = ko.render_mean(oracle_builtin, ko.default_camera(), p, 16, counters=cn)[..., :3]
    assert mae_clamped(acc, ref) <= 1e-4          # measured 3e-6
    assert frac_within_rel(acc, ref) >= 0.94      # measured 0.962: any of 16 frames may flip a texel
    assert 5.7 < cn.rays / cn.paths < 6.1         # ~5.9 rays per path on this scene


def test_converged_4096spp(oracle_builtin):
    """The converged image: the reference's running mean over 4096 frames (SAMPLES patched to 4096) at
    160x90, 32-bounce cap. A texel flipped by llvmpipe's atan/acos in one frame moves a pixel by at most
    1/4096 of a texel difference, so the tolerance is far tighter than for a single frame."""
    ref = golden("pt_160x90_4096spp.npz")["acc"]
    p = ko.default_params(width=160, height=90, mode=ko.MODE_PT, max_frames=0)
    acc = ko.render_mean(oracle_builtin, ko.default_camera(), p, 4096)[..., :3]
    assert mae_clamped(acc, ref) <= CONVERGED_MAE
    assert np.abs(acc - ref).max() <= CONVERGED_MAX_ABS
    assert frac_within_rel(acc, ref, tol=CONVERGED_REL) >= CONVERGED_FRAC


def test_bounce_cap_parameter(oracle_builtin):
    ref = golden("pt_160x90_4spp_8bounce.npz")["acc"]
    p = ko.default_params(width=160, height=90, mode=ko.MODE_PT, max_frames=30, max_bounces=8)
    acc = ko.render_mean(oracle_builtin, ko.default_camera(), p, 4)[..., :3]
    assert mae_clamped(acc, ref) <= 1e-4 and frac_within_rel(acc, ref) >= 0.98


def test_samples_freeze(oracle_builtin):
    """iFrame >= SAMPLES returns the accumulation texel (RawFrame.glsl:426,461-463)."""
    ref = golden("pt_64x36_freeze4_6frames.npz")["acc"]
    p = ko.default_params(width=64, height=36, mode=ko.MODE_PT, max_frames=4)
    acc = ko.render_mean(oracle_builtin, ko.default_camera(), p, 6)[..., :3]
    assert mae_clamped(acc, ref) <= 1e-4 and frac_within_rel(acc, ref) >= 0.98


@pytest.mark.parametrize("name,mode,budget", [("z", ko.MODE_Z, 0.9999), ("ao", ko.MODE_AO, 0.99)])
def test_config2_1080p(name, mode, budget, oracle_builtin):
    """BASELINE config 2: 1920x1080, frame 0, ZDepth and AO (col zero-initialised)."""
    g = golden("config2_1920x1080_f0.npz")
    p = ko.default_params(width=1920, height=1080, mode=mode, max_frames=0)
    out = oracle_builtin.render_raw(ko.default_camera(), p, 0)
    assert np.array_equal(out[..., 0], out[..., 1]) and np.array_equal(out[..., 0], out[..., 2])
    rows = out[3::8, :, 0]
    assert frac_within_rel(rows, g[name + "_rows"]) >= budget
    assert abs(out[..., 0].astype(np.float64).mean() - float(g[name + "_mean"])) <= 1e-6
    assert abs(np.count_nonzero(out[..., 0]) - int(g[name + "_nonzero"])) <= 200


def test_rng_known_answers():
    """rand4() for seed (x, y, frame, x+y); values observed from the shader on llvmpipe."""
    kat = {
        (3, 2, 0): ([4241451707, 1087484907, 2807052885, 2628532002], [0.98753995, 0.25319982, 0.6535679, 0.6120028]),
        (3, 2, 5): ([3260634414, 2699197596, 3298188390, 3720524794], [0.7591756, 0.62845594, 0.7679193, 0.8662522]),
    }
    for (x, y, f), (h, r) in kat.items():
        assert ko.hash4([x, y, f, x + y]).tolist() == h
        assert np.array_equal(ko.rand4(x, y, f), np.array(r, dtype=np.float32))
    more = {(0, 0, 0): [0.0586390607, 0.177101582, 0.198009744, 0.824787676],
            (32, 18, 0): [0.172015533, 0.645216823, 0.0416498035, 0.712452352],
            (63, 35, 0): [0.981784999, 0.271676093, 0.534377098, 0.566799879]}
    for (x, y, f), r in more.items():
        assert np.array_equal(ko.rand4(x, y, f), np.array(r, dtype=np.float32))
    # the float conversion can reach exactly 1.0: float(h)/2^32 rounds up for h near 2^32
    assert np.float32(0xFFFFFFFF) / np.float32(4294967296.0) == 1.0


def test_accumulate_formula():
    rng = np.random.default_rng(0)
    acc = rng.random((5, 7, 4), dtype=np.float32)
    raw = rng.random((5, 7, 4), dtype=np.float32)
    for frame, change in ((0, True), (1, False), (7, False)):
        a = acc.copy()
        ko.accumulate(a, raw, frame, change)
        w = np.float32(frame)
        want = (acc[..., :3] * w + raw[..., :3]) / (w + np.float32(1))
        assert np.array_equal(a[..., :3], want) and np.all(a[..., 3] == 1)
    a = acc.copy()
    ko.accumulate(a, raw, 5, True)     # the unreachable else-branch of Accumulation.glsl:21-23
    assert np.array_equal(a, raw)


def test_math_accuracy():
    x = np.linspace(0, 2 * np.pi, 20001, dtype=np.float32)
    s = np.array([ko.lib().ko_sinf(float(v)) for v in x[::10]])
    c = np.array([ko.lib().ko_cosf(float(v)) for v in x[::10]])
    assert np.abs(s - np.sin(x[::10].astype(np.float64))).max() < 3e-7
    assert np.abs(c - np.cos(x[::10].astype(np.float64))).max() < 3e-7
    v = np.linspace(-1, 1, 2001, dtype=np.float32)
    a = np.array([ko.lib().ko_acosf(float(t)) for t in v])
    assert np.abs(a - np.arccos(v.astype(np.float64))).max() < 1e-6
    ang = np.linspace(-np.pi, np.pi, 2001)
    t = np.array([ko.lib().ko_atan2f(float(np.float32(np.sin(q))), float(np.float32(np.cos(q)))) for q in ang])
    want = np.arctan2(np.float32(np.sin(ang)).astype(np.float64), np.float32(np.cos(ang)).astype(np.float64))
    assert np.abs(t - want).max() < 1e-6

"""CPU suite: the C-ABI library loads, exports every symbol the header declares, and its
host-only entry points (defaults, synthetic-scene generator, argument checks) behave.
No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import klrender as kl
from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "Math", "kl_render.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(r"KL_API\s+[\w\s\*]+?\b(kl_\w+)\s*\(", text)


def test_header_symbols_exported():
    names = declared_symbols()
    assert len(names) >= 30
    L = kl.lib()
    for n in names:
        assert hasattr(L, n), "libklrender.so does not export %s" % n
        assert n in kl.SIGNATURES, "python binding lacks %s" % n
    assert set(kl.SIGNATURES) == set(names)
    assert L.kl_abi_version() == 2


def test_forwarding_header():
    text = open(os.path.join(ROOT, "include", "kl_render.h")).read()
    assert '#include "Math/kl_render.h"' in text


def test_struct_layouts_match_c():
    # sizes the C side assumes (kl_render.h); a mismatch would corrupt uploads silently
    assert kl.material_dtype.itemsize == 32 and kl.sphere_dtype.itemsize == 20
    assert kl.quad_dtype.itemsize == 52 and kl.tri_dtype.itemsize == 64
    assert C.sizeof(kl.Camera) == 44 and C.sizeof(kl.Params) == 64
    assert C.sizeof(kl.Counters) == 72 and C.sizeof(kl.SynthParams) == 16


def test_defaults_match_reference():
    p = kl.default_params()
    # RawFrame.glsl:32-40
    assert (p.max_bounces, p.max_frames) == (32, 30)
    assert p.eps == np.float32(0.001) and p.max_dist == 5000.0 and p.ao_length == 1.0 and p.hdri_strength == 1.0
    c = kl.default_camera()
    # source/Object/Camera.cpp:3-14
    assert list(c.pos) == [0, 1, 5] and list(c.front) == [0, 0, -1] and list(c.up) == [0, 1, 0]
    assert c.focal == np.float32(0.05) and c.sensor == np.float32(0.036)


def test_synthetic_scene_is_deterministic_and_sized():
    s, q, t, m = kl.synth_scene(blobs=5, tris_per_blob=1000, analytic_spheres=3, seed=7)
    assert len(q) == 7 and len(t) == 5000 and len(s) == 3 and len(m) == 7 + 5 + 3
    s2, q2, t2, m2 = kl.synth_scene(blobs=5, tris_per_blob=1000, analytic_spheres=3, seed=7)
    assert s.tobytes() == s2.tobytes() and t.tobytes() == t2.tobytes() and m.tobytes() == m2.tobytes()
    _, _, t3, _ = kl.synth_scene(blobs=5, tris_per_blob=1000, analytic_spheres=3, seed=8)
    assert t.tobytes() != t3.tobytes()
    # inside the room, radii log-uniform in [0.02, 0.12] (+5 % jitter)
    v = np.concatenate([t["v0"], t["v1"], t["v2"]])
    assert v[:, 0].min() > -2.66 and v[:, 0].max() < 2.66 and v[:, 1].min() > -0.1 and v[:, 1].max() < 3.0
    assert np.all(s["radius"] >= 0.02) and np.all(s["radius"] <= 0.12 * 1.0001)
    assert t["material"].min() >= 7 and t["material"].max() < len(m)
    # no degenerate triangles
    e1, e2 = t["v1"] - t["v0"], t["v2"] - t["v0"]
    assert np.all(np.linalg.norm(np.cross(e1, e2), axis=1) > 0)
    # the room is the reference's seven quads (RawFrame.glsl:185-191)
    assert np.allclose(q["v0"][0], [-2.66, 0, -15]) and m["type"][3] == kl.TEXTURED and m["emissive"][5] == 2.5


def test_argument_checks_without_gpu():
    L = kl.lib()
    assert L.kl_default_params(None) == -1
    h = C.c_void_p()
    assert L.kl_create(C.byref(h), 0, 0, 10) == -1           # zero width
    assert L.kl_create(None, 0, 8, 8) == -1
    assert L.kl_destroy(None) == -1
    assert L.kl_render_frames(None, 1, 0) == -1
    sp = kl.SynthParams(1, 4, 0, 1)                            # tris_per_blob too small
    n = [C.c_uint32() for _ in range(4)]
    assert L.kl_synth_scene_counts(C.byref(sp), *[C.byref(x) for x in n]) == -1
    assert L.kl_status_string(-4) == b"invalid state"
    # new entry points refuse null arguments before touching a device
    assert L.kl_get_stream(None, None) == -1 and L.kl_scene_blob_bytes(None, None) == -1
    assert L.kl_scene_export(None, None, 0) == -1 and L.kl_scene_import(None, None, 0) == -1


def test_group_create_failure_keeps_its_reason():
    """kl_group_create has no group to hold an error text when it fails: kl_group_last_error(NULL) gives the reason of the
    last failed create on the calling thread"""
    L = kl.lib()
    g = C.c_void_p()
    devs = (C.c_int32 * 1)(0)
    assert L.kl_group_create(C.byref(g), devs, 0, kl.SPLIT_TILES, 64, 36, 8) == -1 and not g.value
    assert L.kl_group_last_error(None) == b"bad argument"
    # a member that cannot be created (no such device, or no GPU at all here): the member and its reason are named
    bad = (C.c_int32 * 1)(63)
    rc = L.kl_group_create(C.byref(g), bad, 1, kl.SPLIT_TILES | kl.EXCHANGE_COPY, 64, 36, 8)
    assert rc < 0 and not g.value
    assert L.kl_group_last_error(None).startswith(b"member 0 (device 63)")


def test_product_does_not_import_oracle():
    """the oracle is a checker only: nothing under path-tracer_amd/ or include/ may reference it"""
    for base in ("path-tracer_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                    text = open(os.path.join(dirpath, fn)).read()
                    assert "kloracle" not in text and "kl_oracle" not in text and "oracle/" not in text, fn

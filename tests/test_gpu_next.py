"""GPU suite for the rows beside the hot path: float environment maps, accumulation checkpoints, mesh files."""
import numpy as np
import pytest

import klrender as kl
import kloracle as ko
from test_gpu_parity import assert_bitexact

pytestmark = pytest.mark.gpu


def test_float_environment_matches_the_oracle_and_the_8bit_path(textures):
    w, h = 96, 54
    hdri = textures["hdri"]
    rng = np.random.default_rng(4)
    env = (rng.random((64, 128, 3), dtype=np.float32) * 4.0).astype(np.float32)     # values above 1: what 8 bits cannot hold
    with kl.Renderer(w, h) as r:
        r.load_builtin_scene()
        r.upload_texture(kl.TEX_HDRI, hdri); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
        r.update_params(max_bounces=8, max_frames=0)
        r.render_frames(3)
        assert r.frame == 3
        byte_img = r.read_accum()
        # the 8-bit texture restated as floats (byte * (1/255), the sampler's own conversion) gives the same image
        as_float = hdri[..., :3].astype(np.float32) * np.float32(1.0 / 255.0)
        r.upload_env_float(as_float)
        assert r.frame == 0                       # an input change resets the frame counter
        r.render_frames(3)
        assert_bitexact(r.read_accum(), byte_img, "float copy of the 8-bit environment")
        r.upload_env_float(env)
        r.render_frames(3)
        got = r.read_accum()
        r.upload_env_float(None)
        r.render_frames(3)
        assert_bitexact(r.read_accum(), byte_img, "back to the 8-bit environment")
    sc = ko.Scene(builtin=True)
    sc.set_texture(ko.TEX_HDRI, hdri); sc.set_texture(ko.TEX_ALBEDO, textures["albedo"])
    sc.set_env_float(env)
    p = ko.default_params(width=w, height=h, mode=ko.MODE_PT, max_frames=0, max_bounces=8)
    assert_bitexact(got, ko.render_mean(sc, ko.default_camera(), p, 3), "float environment")
    assert got[..., :3].max() > 1.0


def test_checkpoint_resume_is_bit_exact(tmp_path, textures):
    s, q, t, m = kl.synth_scene(blobs=12, tris_per_blob=160, analytic_spheres=10, seed=5)
    w, h = 80, 45
    path = str(tmp_path / "session.klck")

    def setup(r):
        r.upload_scene(s, q, t, m)
        r.upload_texture(kl.TEX_HDRI, textures["hdri"]); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])

    cam = kl.default_camera()
    cam.pos[0], cam.pos[1], cam.pos[2] = 0.4, 1.2, 4.0
    with kl.Renderer(w, h) as r:
        setup(r)
        r.update_params(max_bounces=6, max_frames=0, frames_in_flight=2)
        r.set_camera(cam)
        r.render_frames(5)
        r.save_checkpoint(path)
        r.render_frames(6)
        straight = r.read_accum(); straight_raw = r.read_raw()
    with kl.Renderer(w, h) as r:
        setup(r)                                  # scene and textures are the caller's to restore
        r.load_checkpoint(path)
        assert r.frame == 5 and r.get_params().max_bounces == 6
        r.render_frames(6)
        assert_bitexact(r.read_accum(), straight, "resumed accumulation")
        assert_bitexact(r.read_raw(), straight_raw, "resumed last raw frame")
    with kl.Renderer(w + 1, h) as r:
        setup(r)
        with pytest.raises(kl.KlError, match="another image size"):
            r.load_checkpoint(path)
    bad = tmp_path / "bad.klck"
    bad.write_bytes(b"not a checkpoint")
    with kl.Renderer(w, h) as r:
        setup(r)
        with pytest.raises(kl.KlError, match="not a checkpoint"):
            r.load_checkpoint(str(bad))


def test_mesh_file_scene_renders_like_the_oracle(tmp_path, textures):
    """an OBJ written from the synthetic scene's triangles, read back with kl_load_mesh, rendered on both sides"""
    s, q, t, m = kl.synth_scene(blobs=6, tris_per_blob=100, seed=2)
    lines = []
    for k, tri in enumerate(t):
        for v, uv in ((tri["v0"], tri["uv0"]), (tri["v1"], tri["uv1"]), (tri["v2"], tri["uv2"])):
            lines.append("v %.9g %.9g %.9g" % tuple(v)); lines.append("vt %.9g %.9g" % tuple(uv))
        lines.append("f %d/%d %d/%d %d/%d" % (3 * k + 1, 3 * k + 1, 3 * k + 2, 3 * k + 2, 3 * k + 3, 3 * k + 3))
    p = tmp_path / "blobs.obj"
    p.write_text("\n".join(lines) + "\n")
    tris = kl.load_mesh(str(p), material=6, n_materials=len(m))          # everything TEXTURED-or-whatever slot 6 is
    assert len(tris) == len(t)
    for f in ("v0", "v1", "v2", "uv0", "uv1", "uv2"):
        assert np.array_equal(tris[f], t[f])          # %.9g round-trips fp32
    w, h = 96, 54
    with kl.Renderer(w, h) as r:
        r.upload_scene(s, q, tris, m)
        r.upload_texture(kl.TEX_HDRI, textures["hdri"]); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
        r.update_params(max_bounces=6, max_frames=0, accel=kl.ACCEL_BVH)
        r.render_frames(2)
        got = r.read_accum()
    sc = ko.Scene(s, q, tris, m)
    sc.set_texture(ko.TEX_HDRI, textures["hdri"]); sc.set_texture(ko.TEX_ALBEDO, textures["albedo"])
    pr = ko.default_params(width=w, height=h, mode=ko.MODE_PT, max_frames=0, max_bounces=6, use_bvh=1)
    assert_bitexact(got, ko.render_mean(sc, ko.default_camera(), pr, 2), "scene from a mesh file")

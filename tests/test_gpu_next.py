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


# ---------------------------------------------------------------- GPU-side BVH build
def _trace_both(r, o, d):
    r.update_params(accel=kl.ACCEL_BVH)
    tb = r.trace_rays(o, d)
    r.update_params(accel=kl.ACCEL_BRUTE)
    tf = r.trace_rays(o, d)
    return tb, tf


@pytest.mark.parametrize("radius", [None, "0", "3"])
def test_gpu_built_tree_finds_the_same_hits(radius, monkeypatch):
    """the tree built on the GPU (clustering with the default search radius, the radix tree over the codes instead
    = radius 0, a small radius) against the flat loop and against the host SAH tree: same primitive, same t and uv
    bit for bit (the closest hit does not depend on the tree), degenerate inputs included"""
    from test_gpu_scene import random_rays
    if radius is not None:
        monkeypatch.setenv("KL_PLOC_RADIUS", radius)
    s, q, t, m = kl.synth_scene(blobs=24, tris_per_blob=160, analytic_spheres=40, seed=3)
    o, d = random_rays(3000, 13)
    o[:50] = t["v0"][:50]
    d[50:60] = np.eye(3, dtype=np.float32)[np.arange(10) % 3]
    with kl.Renderer(8, 8) as r:
        r.update_params(bvh_builder=kl.BUILD_GPU_LBVH)
        r.upload_scene(s, q, t, m)
        st = r.bvh_stats()
        assert st.builder == kl.BUILD_GPU_LBVH and st.max_leaf_size <= 2
        assert (st.spheres, st.quads, st.tris) == (len(s), len(q), len(t))
        assert st.flat_quads == 7 and st.leaves == len(s) + len(q) + len(t) - 7       # the room's walls stay out of the tree
        assert 1 <= st.nodes <= st.leaves - 1 and st.max_depth <= 32 and st.node_bytes == st.nodes * 64
        (tb, pb, ub), (tf, pf, uf) = _trace_both(r, o, d)
        r.update_params(bvh_builder=kl.BUILD_HOST_SAH, accel=kl.ACCEL_BVH)
        r.upload_scene(s, q, t, m)
        assert r.bvh_stats().builder == kl.BUILD_HOST_SAH
        th, ph, uh = r.trace_rays(o, d)
    assert np.array_equal(pb, pf) and np.array_equal(ph, pf) and (pf != kl.MISS).mean() > 0.9
    assert_bitexact(tb, tf, "t: GPU-built tree vs flat loop"); assert_bitexact(ub, uf, "uv: GPU-built tree vs flat loop")
    assert_bitexact(th, tf, "t: host tree vs flat loop")
    # many primitives in one Morton cell (exact duplicates) and a degenerate zero-area triangle
    dup = np.zeros(3000, dtype=kl.tri_dtype)
    dup["v0"] = (0.0, 1.0, -3.0); dup["v1"] = (0.5, 1.0, -3.0); dup["v2"] = (0.0, 1.5, -3.0)
    dup[-1]["v1"] = dup[-1]["v0"]
    dup["material"] = 0
    mats = np.zeros(1, dtype=kl.material_dtype); mats["type"] = kl.DIFFUSE; mats["color"] = 0.5; mats["roughness"] = 1.0
    o2 = np.tile(np.array([[0.1, 1.1, 2.0]], dtype=np.float32), (64, 1)); d2 = np.tile(np.array([[0, 0, -1]], dtype=np.float32), (64, 1))
    with kl.Renderer(8, 8) as r:
        r.update_params(bvh_builder=kl.BUILD_GPU_LBVH)
        r.upload_scene(None, None, dup, mats)
        assert r.bvh_stats().builder == kl.BUILD_GPU_LBVH
        (tb, pb, ub), (tf, pf, uf) = _trace_both(r, o2, d2)
    assert np.all(pb == 0) and np.array_equal(pb, pf)              # exact ties: the lowest declaration-order id wins
    assert_bitexact(tb, tf, "duplicates: t")


def test_gpu_built_tree_renders_like_the_oracle(textures):
    s, q, t, m = kl.synth_scene(blobs=24, tris_per_blob=160, analytic_spheres=40, seed=3)
    sc = ko.Scene(s, q, t, m)
    sc.set_texture(ko.TEX_HDRI, textures["hdri"]); sc.set_texture(ko.TEX_ALBEDO, textures["albedo"])
    w, h = 128, 72
    with kl.Renderer(w, h) as r:
        r.update_params(bvh_builder=kl.BUILD_GPU_LBVH, max_bounces=8, max_frames=0, accel=kl.ACCEL_BVH)
        r.upload_scene(s, q, t, m)
        r.upload_texture(kl.TEX_HDRI, textures["hdri"]); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
        r.render_frames(3)
        acc = r.read_accum()
    p = ko.default_params(width=w, height=h, mode=ko.MODE_PT, max_frames=0, max_bounces=8, use_bvh=1)
    assert_bitexact(acc, ko.render_mean(sc, ko.default_camera(), p, 3), "GPU-built tree, path traced")


def test_records_packed_on_the_gpu_match_the_host_packing(textures, monkeypatch):
    """The GPU builder writes the scene's records on the device; packed on the host from the same tree
    (KL_PACK_HOST=1) every pixel of every mode must come out the same, flat quads and textured triangles included."""
    s, q, t, m = kl.synth_scene(blobs=30, tris_per_blob=200, analytic_spheres=60, seed=11)
    w, h = 160, 90
    out = {}
    for where in ("gpu", "host"):
        if where == "host":
            monkeypatch.setenv("KL_PACK_HOST", "1")
        with kl.Renderer(w, h) as r:
            r.update_params(bvh_builder=kl.BUILD_GPU_LBVH, max_bounces=8, max_frames=0, accel=kl.ACCEL_BVH)
            r.upload_scene(s, q, t, m)
            st = r.bvh_stats()
            assert st.builder == kl.BUILD_GPU_LBVH
            r.upload_texture(kl.TEX_HDRI, textures["hdri"]); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
            imgs = []
            for mode in (kl.MODE_PT, kl.MODE_Z, kl.MODE_AO):
                r.update_params(mode=mode); r.reset()
                r.render_frames(4)
                imgs.append(r.read_accum())
            out[where] = (imgs, (st.nodes, st.flat_spheres, st.flat_quads, st.flat_tris))
    assert out["gpu"][1] == out["host"][1]
    assert out["gpu"][1][2] > 0, "the scene is meant to have flat quads"
    for a, b, name in zip(out["gpu"][0], out["host"][0], ("path traced", "depth", "ao")):
        assert_bitexact(a, b, "records packed on the GPU vs on the host: " + name)


def test_gpu_build_of_a_million_triangles():
    """BASELINE config 3's scene: the GPU build against the host build on the same rays"""
    from test_gpu_scene import random_rays
    s, q, t, m = kl.synth_scene(blobs=1000, tris_per_blob=1000, seed=1)
    o, d = random_rays(20000, 7)
    with kl.Renderer(8, 8) as r:
        r.update_params(bvh_builder=kl.BUILD_GPU_LBVH, accel=kl.ACCEL_BVH)
        r.upload_scene(s, q, t, m)
        st = r.bvh_stats()
        assert st.builder == kl.BUILD_GPU_LBVH and st.tris == 1_000_000 and st.max_depth <= 32
        tg, pg, ug = r.trace_rays(o, d)
        gpu_seconds = st.build_seconds
        r.update_params(bvh_builder=kl.BUILD_HOST_SAH)
        r.upload_scene(s, q, t, m)
        th, ph, uh = r.trace_rays(o, d)
        host_seconds = r.bvh_stats().build_seconds
    assert np.array_equal(pg, ph)
    assert_bitexact(tg, th, "1M triangles: GPU-built vs host-built tree, t"); assert_bitexact(ug, uh, "uv")
    print("build of 1M triangles: GPU %.4f s, host %.4f s" % (gpu_seconds, host_seconds))
    assert gpu_seconds < host_seconds


# ---------------------------------------------------------------- refit
def _moved(s, q, t, seed):
    """the same primitives somewhere else: an affine map of everything (quads stay planar), every blob vertex
    jittered on top, spheres resized"""
    rng = np.random.default_rng(seed)
    A = (np.eye(3) + 0.06 * rng.standard_normal((3, 3))).astype(np.float32)
    b = np.array([0.05, -0.03, 0.1], dtype=np.float32)
    s2, q2, t2 = s.copy(), q.copy(), t.copy()
    s2["center"] = s["center"] @ A.T + b
    s2["radius"] = s["radius"] * rng.uniform(0.7, 1.4, len(s)).astype(np.float32)
    for v in ("v0", "v1", "v2", "v3"):
        q2[v] = q[v] @ A.T + b
    for v in ("v0", "v1", "v2"):
        t2[v] = t[v] @ A.T + b + rng.uniform(-0.01, 0.01, (len(t), 3)).astype(np.float32)
    for v in ("uv0", "uv1", "uv2"):
        t2[v] = (t[v] * 0.9 + 0.05).astype(np.float32)
    return s2, q2, t2


@pytest.mark.parametrize("builder", [kl.BUILD_HOST_SAH, kl.BUILD_GPU_LBVH])
def test_refit_finds_the_hits_of_a_fresh_upload(builder):
    """kl_refit_scene keeps the tree and recomputes its boxes: every ray then finds what it finds in a tree built
    for the moved scene and what the flat loop finds, bit for bit"""
    from test_gpu_scene import random_rays
    s, q, t, m = kl.synth_scene(blobs=24, tris_per_blob=160, analytic_spheres=40, seed=3)
    s2, q2, t2 = _moved(s, q, t, 5)
    o, d = random_rays(4000, 17)
    o[:60] = t2["v0"][:60]
    with kl.Renderer(8, 8) as r:
        r.update_params(bvh_builder=builder)
        r.upload_scene(s, q, t, m)
        before = r.bvh_stats()
        assert before.refit_seconds == 0.0
        r.refit_scene(s2, q2, t2)
        after = r.bvh_stats()
        assert after.refit_seconds > 0.0 and (after.nodes, after.leaves, after.flat_quads) == (before.nodes, before.leaves, before.flat_quads)
        (tb, pb, ub), (tf, pf, uf) = _trace_both(r, o, d)
        r.update_params(accel=kl.ACCEL_BVH)
        r.upload_scene(s2, q2, t2, m)
        tn, pn, un = r.trace_rays(o, d)
        assert r.bvh_stats().refit_seconds == 0.0
        # and back again: the refit must not depend on what the boxes held before
        r.refit_scene(s, q, t)
        t0, p0, u0 = r.trace_rays(o, d)
        r.upload_scene(s, q, t, m)
        t1, p1, u1 = r.trace_rays(o, d)
    assert (pf != kl.MISS).mean() > 0.9
    assert np.array_equal(pb, pf) and np.array_equal(pb, pn)
    assert_bitexact(tb, tf, "refitted tree vs flat loop: t"); assert_bitexact(ub, uf, "refitted tree vs flat loop: uv")
    assert_bitexact(tb, tn, "refitted tree vs fresh tree: t"); assert_bitexact(ub, un, "refitted tree vs fresh tree: uv")
    assert np.array_equal(p0, p1)
    assert_bitexact(t0, t1, "refitted back vs fresh: t"); assert_bitexact(u0, u1, "refitted back vs fresh: uv")


def test_refit_renders_like_the_oracle(textures):
    s, q, t, m = kl.synth_scene(blobs=24, tris_per_blob=160, analytic_spheres=40, seed=3)
    s2, q2, t2 = _moved(s, q, t, 9)
    w, h = 128, 72
    with kl.Renderer(w, h) as r:
        r.update_params(max_bounces=8, max_frames=0, accel=kl.ACCEL_BVH)
        r.upload_scene(s, q, t, m)
        r.upload_texture(kl.TEX_HDRI, textures["hdri"]); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
        r.render_frames(2)
        r.refit_scene(s2, q2, t2)
        assert r.frame == 0                       # a scene change restarts the accumulation
        r.render_frames(3)
        acc = r.read_accum()
    sc = ko.Scene(s2, q2, t2, m)
    sc.set_texture(ko.TEX_HDRI, textures["hdri"]); sc.set_texture(ko.TEX_ALBEDO, textures["albedo"])
    p = ko.default_params(width=w, height=h, mode=ko.MODE_PT, max_frames=0, max_bounces=8, use_bvh=1)
    assert_bitexact(acc, ko.render_mean(sc, ko.default_camera(), p, 3), "refitted scene, path traced")


def test_refit_rejects_what_it_cannot_do():
    s, q, t, m = kl.synth_scene(blobs=4, tris_per_blob=160, analytic_spheres=4, seed=3)
    with kl.Renderer(8, 8) as r:
        with pytest.raises(kl.KlError):
            r.refit_scene(s, q, t)                # nothing uploaded yet
        r.upload_scene(s, q, t, m)
        with pytest.raises(kl.KlError):
            r.refit_scene(s, q, t[:-1])           # another primitive count
        r.refit_scene(s, q, t)


def test_refit_of_a_million_triangles():
    from test_gpu_scene import random_rays
    s, q, t, m = kl.synth_scene(blobs=1000, tris_per_blob=1000, seed=1)
    s2, q2, t2 = _moved(s, q, t, 2)
    o, d = random_rays(20000, 7)
    with kl.Renderer(8, 8) as r:
        r.update_params(accel=kl.ACCEL_BVH)
        r.upload_scene(s, q, t, m)
        build = r.bvh_stats().build_seconds
        r.refit_scene(s2, q2, t2)
        r.refit_scene(s2, q2, t2)
        refit = r.bvh_stats().refit_seconds
        ta, pa, ua = r.trace_rays(o, d)
        r.upload_scene(s2, q2, t2, m)
        tb, pb, ub = r.trace_rays(o, d)
    assert np.array_equal(pa, pb)
    assert_bitexact(ta, tb, "1M triangles refitted vs rebuilt: t"); assert_bitexact(ua, ub, "uv")
    print("1M triangles: host SAH build %.4f s, refit %.4f s" % (build, refit))
    assert refit < build


def test_packed_scene_block_import_equals_a_fresh_upload(textures):
    """kl_scene_export / kl_scene_import: the packed scene (records, tree, shading records, flat-quad table) leaves one
    context as a block of device memory and becomes another context's scene: same hits and same pixels, bit for bit, as the
    context that built the tree; the imported scene can be refitted; damaged blocks are refused before anything is copied."""
    import torch
    from test_gpu_scene import random_rays
    s, q, t, m = kl.synth_scene(blobs=24, tris_per_blob=160, analytic_spheres=40, seed=3)
    o, d = random_rays(20000, 41)
    w, h = 96, 54
    dev = torch.device("cuda", 0)
    with kl.Renderer(w, h) as a, kl.Renderer(w, h) as b:
        with pytest.raises(kl.KlError, match="invalid state"):
            a.scene_blob_bytes()
        a.update_params(max_bounces=8, max_frames=0, accel=kl.ACCEL_BVH)
        b.update_params(max_bounces=8, max_frames=0, accel=kl.ACCEL_BVH)
        a.upload_scene(s, q, t, m)
        n = a.scene_blob_bytes()
        assert n > 64 * (len(s) + len(q) + len(t)) * 2
        block = torch.zeros(n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        with pytest.raises(kl.KlError):
            a.scene_export(block.data_ptr(), n - 256)
        a.scene_export(block.data_ptr(), n)
        # refused: truncated, a foreign header, a header whose sizes do not add up to the block
        with pytest.raises(kl.KlError):
            b.scene_import(block.data_ptr(), 128)
        with pytest.raises(kl.KlError):
            b.scene_import(block.data_ptr(), n - 256)
        bad = block.clone(); bad[0] ^= 0xFF
        torch.cuda.synchronize(dev)
        with pytest.raises(kl.KlError, match="not a packed scene"):
            b.scene_import(bad.data_ptr(), n)
        bad = block.clone(); bad[16:24] = torch.tensor([0xFF] * 7 + [0x00], dtype=torch.uint8, device=dev)     # n_rec_f4: huge
        torch.cuda.synchronize(dev)
        with pytest.raises(kl.KlError):
            b.scene_import(bad.data_ptr(), n)
        with pytest.raises(kl.KlError, match="invalid state"):
            b.render_frames(1)                                # none of the refused blocks became a scene
        b.scene_import(block.data_ptr(), n)
        del block, bad
        sa, sb = a.bvh_stats(), b.bvh_stats()
        assert (sa.nodes, sa.leaves, sa.max_depth, sa.flat_quads, sa.tris, sa.spheres) == (sb.nodes, sb.leaves, sb.max_depth, sb.flat_quads, sb.tris, sb.spheres)
        ta, pa, ua = a.trace_rays(o, d)
        tb, pb, ub = b.trace_rays(o, d)
        assert (pa != kl.MISS).mean() > 0.9 and np.array_equal(pa, pb)
        assert_bitexact(ta, tb, "imported scene: t"); assert_bitexact(ua, ub, "imported scene: uv")
        for r in (a, b):
            r.upload_texture(kl.TEX_HDRI, textures["hdri"]); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
            r.render_frames(3)
        assert_bitexact(a.read_accum(), b.read_accum(), "imported scene, path traced")
        # the imported scene carries what a refit needs (flat-quad indices, tree root)
        s2, q2, t2 = _moved(s, q, t, 5)
        a.refit_scene(s2, q2, t2); b.refit_scene(s2, q2, t2)
        ta, pa, ua = a.trace_rays(o, d)
        tb, pb, ub = b.trace_rays(o, d)
        assert np.array_equal(pa, pb)
        assert_bitexact(ta, tb, "imported scene, refitted: t"); assert_bitexact(ua, ub, "imported scene, refitted: uv")

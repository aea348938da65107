"""GPU suite: uploaded scenes (spheres + quads + triangles), the SAH BVH, partitioning.

Triangles, the BVH and scene upload have no counterpart in the reference shader, so they are
pinned by self-consistency: BVH traversal == flat loop == the CPU oracle (brute force and its
own, differently shaped, median-split tree), bit for bit.
"""
import numpy as np
import pytest

import klrender as kl
import kloracle as ko
from test_gpu_parity import assert_bitexact, KO_MODE

pytestmark = pytest.mark.gpu


def random_rays(n, seed, box=((-2.6, 0.0, -15.0), (2.6, 3.0, 5.0))):
    rng = np.random.default_rng(seed)
    lo, hi = np.array(box[0]), np.array(box[1])
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


def oracle_trace(scene, o, d, use_bvh):
    p = ko.default_params(use_bvh=use_bvh)
    t = np.empty(len(o), dtype=np.float32)
    prim = np.empty(len(o), dtype=np.int64)
    uv = np.empty((len(o), 2), dtype=np.float32)
    for i in range(len(o)):
        prim[i], t[i], uv[i] = scene.trace(p, o[i], d[i])
    return t, prim, uv


@pytest.fixture(scope="module")
def small_scene():
    return kl.synth_scene(blobs=24, tris_per_blob=160, analytic_spheres=40, seed=3)


def test_bvh_guarantees(small_scene):
    s, q, t, m = small_scene
    with kl.Renderer(8, 8) as r:
        r.upload_scene(s, q, t, m)
        st = r.bvh_stats()
    assert (st.spheres, st.quads, st.tris) == (len(s), len(q), len(t))
    assert st.max_depth <= 32 and 1 <= st.max_leaf_size <= 8
    assert st.leaves <= len(s) + len(q) + len(t) and 1 <= st.nodes <= st.leaves - 1     # binary: ==, 4-wide: fewer
    assert st.node_bytes == st.nodes * 64


def test_bvh_equals_flat_loop_equals_oracle(small_scene):
    s, q, t, m = small_scene
    o, d = random_rays(3000, 11)
    # rays that start on geometry and axis-parallel rays stress the slab test
    o[:50] = t["v0"][:50]
    d[50:60] = np.eye(3, dtype=np.float32)[np.arange(10) % 3]
    d[60:70] = -np.eye(3, dtype=np.float32)[np.arange(10) % 3]
    d[70] = (np.nan, 0, 1)                                   # a NaN ray hits nothing
    with kl.Renderer(8, 8) as r:
        r.upload_scene(s, q, t, m)
        r.update_params(accel=kl.ACCEL_BVH)
        tb, pb, ub = r.trace_rays(o, d)
        r.update_params(accel=kl.ACCEL_BRUTE)
        tf, pf, uf = r.trace_rays(o, d)
        r.update_params(accel=kl.ACCEL_BVH, collect_stats=1)
        ts, ps, us = r.trace_rays(o, d)
        cn = r.counters()
    assert np.array_equal(pb, pf) and np.array_equal(ps, pf)
    assert_bitexact(tb, tf, "t bvh vs flat")
    assert_bitexact(ub, uf, "uv bvh vs flat")
    assert_bitexact(ts, tf, "t bvh(stats) vs flat")
    assert pf[70] == kl.MISS and (pf != kl.MISS).mean() > 0.9      # closed room: nearly every ray hits
    assert cn.nodes > 0 and cn.tri_tests > 0 and 0 < cn.lane_steps_active <= cn.lane_steps_total
    sc = ko.Scene(s, q, t, m)
    for use_bvh in (0, 1):
        to, po, uo = oracle_trace(sc, o[:600], d[:600], use_bvh)
        assert np.array_equal(np.where(po < 0, kl.MISS, po).astype(np.uint32), pf[:600])
        assert_bitexact(to, tf[:600], "t vs oracle")
        assert_bitexact(uo, uf[:600], "uv vs oracle")


@pytest.mark.parametrize("mode", [kl.MODE_Z, kl.MODE_AO, kl.MODE_PT])
def test_synthetic_scene_render_bitexact(mode, small_scene, textures):
    s, q, t, m = small_scene
    sc = ko.Scene(s, q, t, m)
    sc.set_texture(ko.TEX_HDRI, textures["hdri"]); sc.set_texture(ko.TEX_ALBEDO, textures["albedo"])
    w, h = 128, 72
    with kl.Renderer(w, h) as r:
        r.upload_scene(s, q, t, m)
        r.upload_texture(kl.TEX_HDRI, textures["hdri"]); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
        r.update_params(max_bounces=8, max_frames=0, accel=kl.ACCEL_BVH)
        r.set_mode(mode)
        r.render_frames(3)
        acc = r.read_accum()
    p = ko.default_params(width=w, height=h, mode=KO_MODE[mode], max_frames=0, max_bounces=8, use_bvh=1)
    want = ko.render_mean(sc, ko.default_camera(), p, 3)
    assert_bitexact(acc, want, "synthetic scene mode %d" % mode)
    assert np.isfinite(acc).all() and acc[..., :3].max() > 0


def test_million_triangle_scene_low_res(textures):
    """BASELINE config 3's scene (1000 blobs x 1000 triangles) at a size the oracle finishes in seconds"""
    s, q, t, m = kl.synth_scene(blobs=1000, tris_per_blob=1000, seed=1)
    assert len(t) == 1_000_000
    w, h = 96, 54
    with kl.Renderer(w, h) as r:
        r.upload_scene(s, q, t, m)
        st = r.bvh_stats()
        assert st.max_depth <= 32 and st.tris == 1_000_000
        r.upload_texture(kl.TEX_HDRI, textures["hdri"]); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
        r.update_params(max_bounces=8, max_frames=0)
        r.render_frames(2)
        acc = r.read_accum()
        o, d = random_rays(2048, 5)
        tb, pb, ub = r.trace_rays(o, d)
        r.update_params(accel=kl.ACCEL_BRUTE)                 # 2048 rays x 1M triangles through the flat loop
        tf, pf, uf = r.trace_rays(o, d)
    assert np.array_equal(pb, pf)
    assert_bitexact(tb, tf, "1M tris: bvh vs flat t")
    sc = ko.Scene(s, q, t, m)
    sc.set_texture(ko.TEX_HDRI, textures["hdri"]); sc.set_texture(ko.TEX_ALBEDO, textures["albedo"])
    p = ko.default_params(width=w, height=h, mode=ko.MODE_PT, max_frames=0, max_bounces=8, use_bvh=1)
    assert_bitexact(acc, ko.render_mean(sc, ko.default_camera(), p, 2), "1M-triangle scene")


@pytest.mark.parametrize("world,band_rows,size", [(2, 8, (96, 54)), (3, 5, (50, 33)), (8, 4, (64, 36)), (4, 16, (40, 10))])
def test_row_band_shards_stitch_to_the_full_image(world, band_rows, size, small_scene, textures):
    """image-tile partition: every rank renders its row bands with true global pixel coordinates;
    stitched together the result is bit-identical to the one-GPU render (ragged last band included)"""
    s, q, t, m = small_scene
    w, h = size

    def setup(r):
        r.upload_scene(s, q, t, m)
        r.upload_texture(kl.TEX_HDRI, textures["hdri"])
        r.update_params(max_bounces=6, frames_in_flight=2)

    with kl.Renderer(w, h) as r:
        setup(r)
        r.render_frames(3)
        full = r.read_accum()
    out = np.zeros_like(full)
    seen = np.zeros(h, dtype=int)
    for rank in range(world):
        with kl.Renderer(w, h) as r:
            setup(r)
            r.set_shard(rank, world, band_rows)
            rows = [y for y in range(h) if (y // band_rows) % world == rank]
            assert r.local_rows == len(rows)
            if not rows:
                r.render_frames(3)        # a rank without rows is a no-op, not an error
                continue
            r.render_frames(3)
            out[rows] = r.read_accum()
            seen[rows] += 1
    assert np.all(seen == 1)
    assert_bitexact(out, full, "stitched shards")


def test_sample_range_split_sums_to_the_mean(small_scene, textures):
    """sample-range partition: disjoint frame ranges accumulated as sums, then added and divided"""
    s, q, t, m = small_scene
    w, h, n = 80, 45, 12

    def setup(r):
        r.upload_scene(s, q, t, m)
        r.upload_texture(kl.TEX_HDRI, textures["hdri"])
        r.update_params(max_bounces=6, max_frames=0)

    with kl.Renderer(w, h) as r:
        setup(r)
        r.render_frames(n)
        mean = r.read_accum()
    total = np.zeros_like(mean, dtype=np.float64)
    for g in range(3):
        with kl.Renderer(w, h) as r:
            setup(r)
            r.set_accum_mode(kl.ACCUM_SUM)
            r.render_frames(n // 3, first_frame=g * (n // 3))
            total += r.read_accum()
    got = (total[..., :3] / n).astype(np.float32)
    # only fp32 re-association separates the two orders of summation
    assert np.allclose(got, mean[..., :3], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("lanes", [1, 2, 3, 4])
def test_batch_lanes_do_not_change_the_image(lanes, small_scene, textures):
    """consecutive batches run on separate HIP streams but accumulate in frame order: the running mean is
    bit-identical to the oracle's whatever the number of streams, also across calls and with a ragged last batch"""
    s, q, t, m = small_scene
    sc = ko.Scene(s, q, t, m)
    sc.set_texture(ko.TEX_HDRI, textures["hdri"]); sc.set_texture(ko.TEX_ALBEDO, textures["albedo"])
    w, h = 64, 36
    with kl.Renderer(w, h) as r:
        r.upload_scene(s, q, t, m)
        r.upload_texture(kl.TEX_HDRI, textures["hdri"]); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
        r.update_params(max_bounces=6, max_frames=0, frames_in_flight=2, batch_lanes=lanes, accel=kl.ACCEL_BVH)
        r.render_frames(7)            # batches of 2, 2, 2, 1 frames
        r.render_frames(4)            # continues at frame 7
        acc = r.read_accum()
        raw = r.read_raw()
    p = ko.default_params(width=w, height=h, mode=ko.MODE_PT, max_frames=0, max_bounces=6, use_bvh=1)
    want = ko.render_mean(sc, ko.default_camera(), p, 11)
    assert_bitexact(acc, want, "running mean with %d lanes" % lanes)
    assert np.isfinite(raw).all()


@pytest.mark.parametrize("mode", [kl.MODE_Z, kl.MODE_AO, kl.MODE_PT])
def test_shared_camera_ray_hits_do_not_change_the_image(mode, small_scene, textures):
    """the camera ray of a pixel is the same in every frame; tracing it once per batch or once per frame gives the
    same bits, in every view mode, with ragged sizes and with shards"""
    s, q, t, m = small_scene
    w, h = 77, 41

    def run(per_frame, shard=None):
        with kl.Renderer(w, h) as r:
            r.upload_scene(s, q, t, m)
            r.upload_texture(kl.TEX_HDRI, textures["hdri"]); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
            r.update_params(max_bounces=6, max_frames=0, frames_in_flight=4, trace_primary_per_frame=per_frame, accel=kl.ACCEL_BVH)
            r.set_mode(mode)
            if shard:
                r.set_shard(*shard)
            r.render_frames(9)              # batches of 4, 4 and 1 frames
            return r.read_accum(), r.counters().rays

    a, rays_a = run(1)
    b, rays_b = run(0)
    assert_bitexact(b, a, "shared camera-ray hits, mode %d" % mode)
    assert rays_b < rays_a
    c, _ = run(0, shard=(1, 3, 5))
    rows = [y for y in range(h) if (y // 5) % 3 == 1]
    assert_bitexact(c, a[rows], "shared camera-ray hits on a shard")


def _soup(rng, ns, nq, nt, nm=5):
    """random primitives with no structure at all: tiny and huge, slivers, overlapping, one degenerate"""
    s = np.zeros(ns, dtype=kl.sphere_dtype); q = np.zeros(nq, dtype=kl.quad_dtype); t = np.zeros(nt, dtype=kl.tri_dtype)
    m = np.zeros(nm, dtype=kl.material_dtype)
    m["type"] = rng.integers(0, 4, nm); m["color"] = rng.random((nm, 3)); m["roughness"] = rng.random(nm); m["ior"] = 1.3
    span = np.array([4.0, 3.0, 8.0], dtype=np.float32)
    s["center"] = (rng.random((ns, 3)) - 0.5) * span; s["radius"] = np.exp(rng.uniform(np.log(0.01), np.log(0.8), ns))
    s["material"] = rng.integers(0, nm, ns)
    c = (rng.random((nq, 3)) - 0.5) * span                            # parallelograms: planar by construction
    a = rng.normal(size=(nq, 3)) * np.exp(rng.uniform(np.log(0.02), np.log(2.0), (nq, 1)))
    b = rng.normal(size=(nq, 3)) * np.exp(rng.uniform(np.log(0.02), np.log(2.0), (nq, 1)))
    q["v0"] = c; q["v1"] = c + a; q["v2"] = c + a + b; q["v3"] = c + b; q["material"] = rng.integers(0, nm, nq)
    c = (rng.random((nt, 3)) - 0.5) * span
    size = np.exp(rng.uniform(np.log(0.005), np.log(3.0), (nt, 1)))
    t["v0"] = c; t["v1"] = c + rng.normal(size=(nt, 3)) * size; t["v2"] = c + rng.normal(size=(nt, 3)) * size
    t["uv0"] = rng.random((nt, 2)); t["uv1"] = rng.random((nt, 2)); t["uv2"] = rng.random((nt, 2))
    t["material"] = rng.integers(0, nm, nt)
    if nt > 3:
        t[3]["v2"] = t[3]["v1"]                                      # zero area
    return s, q, t, m


@pytest.mark.parametrize("counts", [(1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 2), (3, 2, 4), (0, 0, 33), (5, 5, 40), (40, 25, 700), (300, 60, 5000)])
def test_random_soups_tree_equals_flat_loop_equals_oracle(counts):
    """scenes without any structure, from one primitive up: host tree, GPU-built tree and a tree refitted from another
    soup of the same counts all find the flat loop's hits, and the oracle's, bit for bit"""
    rng = np.random.default_rng(1000 + sum(counts))
    s, q, t, m = _soup(rng, *counts)
    s0, q0, t0, _ = _soup(rng, *counts)
    o, d = random_rays(1500, 23, box=((-3.0, -2.5, -5.0), (3.0, 2.5, 5.0)))
    got = {}
    with kl.Renderer(8, 8) as r:
        for name, builder in (("host", kl.BUILD_HOST_SAH), ("gpu", kl.BUILD_GPU_LBVH)):
            r.update_params(bvh_builder=builder, accel=kl.ACCEL_BVH)
            r.upload_scene(s, q, t, m)
            got[name] = r.trace_rays(o, d)
            r.upload_scene(s0, q0, t0, m)                            # a tree built for other geometry ...
            r.refit_scene(s, q, t)                                   # ... bent onto this one
            got[name + " refit"] = r.trace_rays(o, d)
        r.update_params(accel=kl.ACCEL_BRUTE)
        tf, pf, uf = r.trace_rays(o, d)
    for name, (tb, pb, ub) in got.items():
        assert np.array_equal(pb, pf), name
        assert_bitexact(tb, tf, name + ": t"); assert_bitexact(ub, uf, name + ": uv")
    sc = ko.Scene(s, q, t, m)
    to, po, uo = oracle_trace(sc, o[:300], d[:300], 1)
    assert np.array_equal(np.where(po < 0, kl.MISS, po).astype(np.uint32), pf[:300])
    assert_bitexact(to, tf[:300], "t vs oracle"); assert_bitexact(uo, uf[:300], "uv vs oracle")


def test_sum_accumulation_past_the_samples_freeze(small_scene, textures):
    """KL_ACCUM_SUM with SAMPLES = 4: frames 4.. repeat the accumulation (RawFrame.glsl:426,461-463), so they add
    nothing to the sum; sum / 4 equals the frozen running mean up to fp32 re-association, also for a call that starts
    past the freeze (a rank whose sample range lies entirely behind it contributes zeros)."""
    s, q, t, m = small_scene
    w, h = 64, 36
    with kl.Renderer(w, h) as r:
        r.upload_scene(s, q, t, m)
        r.upload_texture(kl.TEX_HDRI, textures["hdri"])
        r.update_params(max_bounces=6, max_frames=4, frames_in_flight=3)
        r.render_frames(9)
        mean = r.read_accum()
        r.set_accum_mode(kl.ACCUM_SUM)
        r.render_frames(9, first_frame=0)          # batches 3 + 1 traced, then five frozen frames
        total = r.read_accum()
        r.set_accum_mode(kl.ACCUM_SUM)
        r.render_frames(3, first_frame=6)          # wholly behind the freeze
        nothing = r.read_accum()
    sc = ko.Scene(s, q, t, m)
    sc.set_texture(ko.TEX_HDRI, textures["hdri"])
    want = ko.render_mean(sc, ko.default_camera(), ko.default_params(width=w, height=h, mode=ko.MODE_PT, max_frames=4,
                                                                     max_bounces=6, use_bvh=1), 9)
    assert_bitexact(mean, want, "frozen running mean vs oracle")
    assert np.allclose(total[..., :3] / 4.0, want[..., :3], rtol=2e-6, atol=1e-7)
    assert not nothing[..., :3].any()


def test_trace_rays_tie_between_a_flat_quad_and_a_tree_quad_keeps_the_uv():
    """A wall-sized quad (kept out of the tree, found by the flat pass) and a small coplanar quad declared after it (in
    the tree): equal t, the lower declaration-order id wins as in the shader's in-order loop (RawFrame.glsl:316), and
    the winner's (u, v) must survive the traversal that met the tie."""
    quads = np.zeros(2, dtype=kl.quad_dtype)
    quads[0]["v0"], quads[0]["v1"], quads[0]["v2"], quads[0]["v3"] = (-8, -8, 0), (8, -8, 0), (8, 8, 0), (-8, 8, 0)
    quads[1]["v0"], quads[1]["v1"], quads[1]["v2"], quads[1]["v3"] = (-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)
    rng = np.random.default_rng(5)
    # enough small triangles elsewhere that a tree is built and the big quad counts as scene-spanning
    tris = np.zeros(200, dtype=kl.tri_dtype)
    c = rng.uniform(-6, 6, (200, 3)).astype(np.float32); c[:, 2] = rng.uniform(-6, -1, 200)
    tris["v0"] = c; tris["v1"] = c + np.float32([0.1, 0, 0]); tris["v2"] = c + np.float32([0, 0.1, 0])
    mats = np.zeros(1, dtype=kl.material_dtype); mats["color"] = 1
    o = np.zeros((256, 3), np.float32); o[:, 0] = rng.uniform(-0.9, 0.9, 256); o[:, 1] = rng.uniform(-0.9, 0.9, 256); o[:, 2] = 3
    d = np.tile(np.float32([0, 0, -1]), (256, 1))
    with kl.Renderer(16, 16) as r:
        r.update_params(accel=kl.ACCEL_BVH)
        r.upload_scene(None, quads, tris, mats)
        st = r.bvh_stats()
        assert st.flat_quads == 1
        t_bvh, p_bvh, uv_bvh = r.trace_rays(o, d)
        r.update_params(accel=kl.ACCEL_BRUTE)
        t_ref, p_ref, uv_ref = r.trace_rays(o, d)
    assert (p_ref == 0).all() and np.array_equal(p_bvh, p_ref) and np.array_equal(t_bvh, t_ref)
    assert np.array_equal(uv_bvh, uv_ref) and uv_ref.any()


@pytest.mark.parametrize("exchange", ["rccl", "copy"])
def test_group_partitions_through_the_c_abi(exchange, small_scene, textures):
    """kl_group_*: image tiles stitch to the one-GPU image bit for bit, sample ranges add up to the mean (fp32
    re-association only; bit for bit the sum of the members' sums over the frame count). `rccl`: the real collectives
    (ncclAllGather / ncclAllReduce) with the one member a one-GPU box allows; `copy`: three members sharing the GPU
    through the copy exchange, ragged image height, so the tile and frame-range arithmetic is exercised with N > 1."""
    s, q, t, m = small_scene
    w, h, n, band = 72, 50, 7, 8
    members = [0] if exchange == "rccl" else [0, 0, 0]
    flag = 0 if exchange == "rccl" else kl.EXCHANGE_COPY
    params = kl.default_params(max_bounces=6, max_frames=0)

    def setup(x):
        x.upload_scene(s, q, t, m)
        x.upload_texture(kl.TEX_HDRI, textures["hdri"]); x.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
        x.set_params(params)

    with kl.Renderer(w, h) as r:
        setup(r)
        r.render_frames(n)
        mean = r.read_accum()
        sums = []
        for k in range(len(members)):
            lo, hi = n * k // len(members), n * (k + 1) // len(members)
            r.set_accum_mode(kl.ACCUM_SUM)
            if hi > lo:
                r.render_frames(hi - lo, first_frame=lo)
                sums.append(r.read_accum())
    with kl.Group(members, kl.SPLIT_TILES | flag, w, h, band) as g:
        setup(g)
        g.render_frames(4, first_frame=0)
        g.render_frames(n - 4)                      # continues at frame 4
        assert_bitexact(g.gather(), mean, "tiles over %d member(s), %s exchange" % (len(members), exchange))
        g.reset()
        g.render_frames(n, first_frame=0)
        assert_bitexact(g.gather(), mean, "tiles again after a reset")
    with kl.Group(members, kl.SPLIT_SAMPLES | flag, w, h, band) as g:
        setup(g)
        g.render_frames(n, first_frame=0)
        got = g.gather()
    total = sums[0][..., :3].copy()
    for x in sums[1:]:
        total += x[..., :3]
    assert_bitexact(got[..., :3], total / np.float32(n), "sample ranges: sum of the members' sums / frames")
    assert np.all(got[..., 3] == 1.0)
    assert np.allclose(got[..., :3], mean[..., :3], rtol=2e-5, atol=1e-6)

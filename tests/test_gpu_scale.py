"""GPU suite at the benchmark's full sizes: properties that do not need the oracle to finish a whole image.

1080p / one million triangles (BASELINE config 3) and 4K tiles (config 4): determinism, shards == whole, batch
lanes == one stream, sample ranges == mean, plus the oracle on a few full-width rows. Config 5's ten-million-
primitive mixed scene: upload, tree guarantees, tree == flat loop on sampled rays, a 4K frame renders finite.
"""
import os
import zlib

import numpy as np
import pytest

import klrender as kl
import kloracle as ko
from conftest import ROOT, free_port
from test_gpu_parity import assert_bitexact
from test_gpu_scene import random_rays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def million():
    return kl.synth_scene(blobs=1000, tris_per_blob=1000, seed=1)


def _setup(r, scene, textures, **params):
    r.upload_scene(*scene)
    r.upload_texture(kl.TEX_HDRI, textures["hdri"]); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
    r.update_params(max_bounces=8, max_frames=0, **params)


def test_1080p_million_triangles_properties(million, textures):
    w, h, n = 1920, 1080, 6
    with kl.Renderer(w, h) as r:
        _setup(r, million, textures)
        r.render_frames(n)
        full = r.read_accum()
        cn = r.counters()
        r.reset(); r.render_frames(n)
        again = r.read_accum()
        # small batches on one stream and on three: same image as one 6-frame batch
        r.update_params(frames_in_flight=2, batch_lanes=1); r.reset(); r.render_frames(n)
        one_lane = r.read_accum()
        r.update_params(frames_in_flight=1, batch_lanes=3); r.reset(); r.render_frames(n)
        three_lanes = r.read_accum()
    assert cn.paths == w * h * n and cn.rays > 3 * cn.paths
    assert np.isfinite(full).all() and (full[..., 3] == 1.0).all()
    assert_bitexact(again, full, "second render of the same frames")
    assert_bitexact(one_lane, full, "batches of 2 frames, one stream")
    assert_bitexact(three_lanes, full, "batches of 1 frame, three streams")
    # checksum of the image: printed so that two boxes can be compared by eye in the log
    print("crc32 of the 1080p 6-frame mean: %08x" % zlib.crc32(full.tobytes()))

    # two row-band shards stitch to the whole (config 4's partition at config 3's size)
    out = np.zeros_like(full)
    for rank in range(2):
        with kl.Renderer(w, h) as r:
            _setup(r, million, textures)
            r.set_shard(rank, 2, 8)
            r.render_frames(n)
            rows = [y for y in range(h) if (y // 8) % 2 == rank]
            out[rows] = r.read_accum()
    assert_bitexact(out, full, "two row-band shards")

    # sample ranges: three disjoint 2-frame sums add up to the 6-frame mean (fp32 re-association only)
    total = np.zeros(full.shape, dtype=np.float64)
    for g in range(3):
        with kl.Renderer(w, h) as r:
            _setup(r, million, textures)
            r.set_accum_mode(kl.ACCUM_SUM)
            r.render_frames(2, first_frame=2 * g)
            total += r.read_accum()
    assert np.allclose((total[..., :3] / n).astype(np.float32), full[..., :3], rtol=2e-5, atol=1e-6)


def test_1080p_rows_against_the_oracle(million, textures):
    """frame 0 of the full-size workload, three full-width row bands, bit for bit against the CPU oracle"""
    w, h = 1920, 1080
    with kl.Renderer(w, h) as r:
        _setup(r, million, textures)
        r.render_frames(1)
        raw = r.read_raw()
    sc = ko.Scene(*million)
    sc.set_texture(ko.TEX_HDRI, textures["hdri"]); sc.set_texture(ko.TEX_ALBEDO, textures["albedo"])
    p = ko.default_params(width=w, height=h, mode=ko.MODE_PT, max_frames=0, max_bounces=8, use_bvh=1)
    for y0 in (0, 537, 1076):
        want = sc.render_raw(ko.default_camera(), p, 0, rows=(y0, y0 + 4))
        assert_bitexact(raw[y0:y0 + 4], want[y0:y0 + 4], "rows %d..%d" % (y0, y0 + 3))


def test_4k_tiles_over_eight_ranks(million, textures):
    """config 4's partition: 3840x2160 in 8-row bands over 8 ranks; every rank's rows equal the whole image's"""
    w, h, n = 3840, 2160, 2
    with kl.Renderer(w, h) as r:
        _setup(r, million, textures)
        r.render_frames(n)
        full = r.read_accum()
    assert np.isfinite(full).all()
    for rank in (0, 3, 7):
        with kl.Renderer(w, h) as r:
            _setup(r, million, textures)
            r.set_shard(rank, 8, 8)
            r.render_frames(n)
            rows = [y for y in range(h) if (y // 8) % 8 == rank]
            assert r.local_rows == len(rows)
            assert_bitexact(r.read_accum(), full[rows], "rank %d of 8" % rank)


def test_ten_million_primitive_mixed_scene(textures):
    """config 5's scene: 9990 blobs x 1000 triangles + 10000 analytic spheres in the seven-quad room"""
    s, q, t, m = kl.synth_scene(blobs=9990, tris_per_blob=1000, analytic_spheres=10000, seed=1)
    assert len(t) == 9_990_000 and len(s) == 10_000 and len(q) == 7
    with kl.Renderer(3840, 2160) as r:
        r.upload_scene(s, q, t, m)
        st = r.bvh_stats()
        assert (st.spheres, st.quads, st.tris) == (len(s), len(q), len(t))
        assert st.max_depth <= 32 and 1 <= st.max_leaf_size <= 8 and st.node_bytes == st.nodes * 64
        print("10 M primitives: %d nodes, build %.2f s, %d flat quads" % (st.nodes, st.build_seconds, st.flat_quads))
        o, d = random_rays(192, 9)
        tb, pb, ub = r.trace_rays(o, d)
        r.update_params(accel=kl.ACCEL_BRUTE)            # 192 rays x 10 M primitives through the flat loop
        tf, pf, uf = r.trace_rays(o, d)
        assert np.array_equal(pb, pf)
        assert_bitexact(tb, tf, "10 M primitives: tree vs flat loop, t")
        assert_bitexact(ub, uf, "10 M primitives: tree vs flat loop, uv")
        r.update_params(accel=kl.ACCEL_BVH, max_bounces=8, max_frames=0)
        r.upload_texture(kl.TEX_HDRI, textures["hdri"]); r.upload_texture(kl.TEX_ALBEDO, textures["albedo"])
        r.render_frames(1)
        raw = r.read_raw()
        cn = r.counters()
    assert np.isfinite(raw).all() and raw[..., :3].max() > 0 and cn.paths == 3840 * 2160
    # frame 0 at the configuration's full size, three full-width row bands, bit for bit against the CPU oracle (its own tree)
    sc = ko.Scene(s, q, t, m)
    sc.set_texture(ko.TEX_HDRI, textures["hdri"]); sc.set_texture(ko.TEX_ALBEDO, textures["albedo"])
    p = ko.default_params(width=3840, height=2160, mode=ko.MODE_PT, max_frames=0, max_bounces=8, use_bvh=1)
    for y0 in (0, 1079, 2157):
        want = sc.render_raw(ko.default_camera(), p, 0, rows=(y0, y0 + 3))
        assert_bitexact(raw[y0:y0 + 3], want[y0:y0 + 3], "10 M primitives, rows %d..%d" % (y0, y0 + 2))


def test_dist_partitions_through_rccl_with_one_rank(textures):
    """klrender/dist.py on the HIP path through the real collectives: torch.distributed "nccl" (= RCCL) with the one rank a
    one-GPU box allows. Image tiles: all_gather_into_tensor + index un-tile, bit-identical to the plain render; sample ranges:
    all_reduce(sum) + divide, the plain mean up to fp32 re-association; both twice in a row with reused buffers."""
    import os
    import torch
    import torch.distributed as dist
    from klrender import dist as kdist
    scene = kl.synth_scene(blobs=12, tris_per_blob=200, analytic_spheres=10, seed=5)
    w, h, n = 96, 54, 6
    port = free_port()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        with kl.Renderer(w, h) as r:
            _setup(r, scene, textures)
            r.render_frames(n)
            mean = r.read_accum()
            tg = kdist.TileGather(h, w, 8, torch.device("cuda", 0))
            for _ in range(2):
                img = kdist.render_tiles(r, n, 8, gatherer=tg).cpu().numpy()
                assert_bitexact(img, mean, "tiles through RCCL")
            r.set_shard(0, 1, 1)
            out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
            for _ in range(2):
                got = kdist.render_sample_ranges(r, n, out=out).cpu().numpy()
                assert np.allclose(got[..., :3], mean[..., :3], rtol=2e-5, atol=1e-6) and np.all(got[..., 3] == 1.0)
            # SAMPLES = 4: the frames past the freeze add nothing, the divisor is the four live frames
            r.update_params(max_frames=4)
            r.set_accum_mode(kl.ACCUM_MEAN)
            r.render_frames(n, first_frame=0)
            frozen = r.read_accum()
            got = kdist.render_sample_ranges(r, n, out=out).cpu().numpy()
            assert np.allclose(got[..., :3], frozen[..., :3], rtol=2e-5, atol=1e-6)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("partition,scaling", [("samples", "weak"), ("samples", "strong"), ("tiles", "strong")])
def test_bench_with_two_ranks_on_one_gpu(partition, scaling):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), rehearsed with two ranks
    that share this box's one GPU: gloo carries the one collective instead of RCCL (which refuses two ranks on a device).
    Small workload; checks the contract fields, the whole-job sample count and that both ranks' work is in the figure."""
    import json
    import subprocess
    import sys
    port = free_port()
    env = dict(os.environ, KL_BENCH_BACKEND="gloo")
    # the other BASELINE multi-GPU lines ride along as sub-records (toy sizes here) in one of the three runs
    extras = "small" if (partition, scaling) == ("samples", "weak") else "none"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "320", "--height", "180", "--spp", "8",
           "--blobs", "20", "--tris-per-blob", "200", "--partition", partition, "--scaling", scaling, "--skip-kernels", "--cpu-seconds", "0",
           "--extras", extras]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["unit"] == "samples/s" and j["vs_baseline"] is None and j["dtype"] == "f32"
    assert j["scaling"] == scaling and j["value"] > 0 and j["higher_is_better"] is True
    frames_total = 16 if (partition, scaling) == ("samples", "weak") else 8
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 - 320 * 180 * frames_total) <= 1e-6 * 320 * 180 * frames_total     # value = whole-job samples / time
    # the tree is built once: rank 0 uploads, the other rank imports the broadcast block
    assert "built once on rank 0" in j["config"]["scene_upload"] and j["config"]["triangles"] == 20 * 200
    if extras == "small":
        m = j["multi_gpu"]
        for key, mode, image, frames in (("strong", "samples", "320x180", 8), ("config4_tiles", "tiles", "640x360", 16), ("config5_samples", "samples", "640x360", 32)):
            c = m[key]
            assert c["partition"] == mode and c["image"] == image and c["frames_in_all"] == frames and c["rccl_ranks"] == 2
            assert c["samples_per_s"] > 0 and c["ms_per_step"] > 0 and c["exchange_ms"] > 0 and c["mrays_per_s"] > 0
            w, h = (int(v) for v in image.split("x"))
            assert abs(c["samples_per_s"] * c["ms_per_step"] * 1e-3 - w * h * frames) <= 1e-6 * w * h * frames
        assert m["config5_samples"]["primitives"] == 60 * 200 + 50 + 7 and m["config5_samples"]["packed_scene_mb"] > 1
    else:
        assert j["multi_gpu"] is None
    assert "gloo" in j["config"]["partition"]


@pytest.mark.parametrize("failing_rank", [0, 1])
def test_bench_line_survives_a_failed_extra(failing_rank):
    """The timed result is printed exactly once whatever happens in the untimed multi-GPU extras: a rank that raises there
    leaves at once (rank 0 after printing the line), the rank left waiting in a collective is released by its time budget."""
    import json
    import subprocess
    import sys
    port = free_port()
    env = dict(os.environ, KL_BENCH_BACKEND="gloo", KL_BENCH_TEST_FAIL_EXTRAS=str(failing_rank), KL_BENCH_EXTRAS_SECONDS="15")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--width", "320", "--height", "180", "--spp", "8",
           "--blobs", "20", "--tris-per-blob", "200", "--skip-kernels", "--skip-shared", "--cpu-seconds", "0", "--extras", "small"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=200)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and "error" in j["multi_gpu"]
    err = j["multi_gpu"]["error"]
    if failing_rank == 0:
        assert "failed on rank 0" in err and "KL_BENCH_TEST_FAIL_EXTRAS" in err
    else:       # rank 0 was left in a collective: its backend reports the lost peer, or its time budget runs out
        assert "failed on rank 0" in err or "did not finish within" in err

#!/usr/bin/env python3
"""Headline benchmark: path samples/s at 1080p, 8 bounces, on N MI355X.

Workload (BASELINE.json configs[2], `--config 3`): the synthetic 1M-triangle textured scene, 1920x1080, 64 spp, bounce cap 8,
path-traced mode. One step = one full pass: 64 frames traced and accumulated (132.7 M path samples). Scene, BVH and textures are
resident in HBM before the timed region, and no per-launch events are recorded inside it. `--config 4` / `--config 5` select
configs[3] (3840x2160, 256 spp, image tiles) and configs[4] (10 M primitives, 4K, 1024 spp, sample ranges).

N > 1 (one process per GPU, torch.distributed "nccl" = RCCL): `--partition samples` (default): every rank renders its frame
range of the same image as a sum, one all-reduce + divide per step (`--scaling weak`: spp frames per rank, N*spp in all;
`--scaling strong`: spp frames in all); `--partition tiles`: every rank renders all frames of its row bands, one all-gather +
un-tile per step. The exchange is inside the timed region; its buffers are allocated before it.

After the timed region (none of it enters `value`): one single-lane step with HIP events around every kernel launch and one
with the traversal kernel's own wave-level tallies (the `roofline` speed-of-light model), the library-default pass with camera
rays shared by a batch's frames (`value_shared_camera_rays`), and the CPU oracle on a bounded sample (`cpu_baseline`).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
VALU_PEAK_TFLOPS = 157.3       # MI355X vector fp32 peak (packed multiply-add on every lane of 256 CUs at 2.4 GHz)
QUEUE_BYTES_PER_RAY = 112      # per ray and bounce: the queue entry (ray, hit, throughput) written once and read twice, as first budgeted;
                               # the current layout moves 48 B written + 32 B (traversal) + 48 B (shade) read + 8 B for one ray in five
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9
SOL_INPUTS = "r02_sol_inputs.json"   # per-ray instruction, L2 and fabric figures from the counter passes kept under profiles/
# lower bound of one dependent record fetch of a traversal step: LDS-DMA issue -> landed from a warm L2 (250 clocks, the low end
# of the measured 250-400) + the LDS read-back (50)
STEP_LATENCY_CLOCKS = 300.0
RESIDENT_TRAVERSAL_WAVES = 256 * 24   # six workgroups of four waves on each of 256 CUs


def load_textures():
    """the reference's two images if the fixtures are present, else seeded noise of the same shape"""
    try:
        from PIL import Image
        bg = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "assets", "Background.png")))
        tx = np.load(os.path.join(ROOT, "tests", "golden", "assets", "Texture_stb_rgb8.npz"))["rgb"]
        return bg, tx, "reference Background.png / Texture.png (stb_image-decoded bytes)"
    except Exception:
        rng = np.random.default_rng(0)
        return (rng.integers(0, 256, (2048, 4096, 4), dtype=np.uint8),
                rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8), "seeded noise textures")


def cpu_baseline(scene_arrays, textures, width, height, bounces, budget_s):
    """Times the CPU oracle (oracle/kl_oracle.c, OpenMP) on the host cores over a bounded sample of
    the same workload: whole frames 0, 1, ... for as long as the time budget lasts. Test infrastructure
    used as a reported baseline only; nothing it computes enters the GPU result."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kloracle as ko
    # every core this process may use: OMP_NUM_THREADS when the box sets it (its share of the machine), else the affinity mask
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, int(os.environ.get("OMP_NUM_THREADS", "0")) or avail)
    s, q, t, m = scene_arrays
    sc = ko.Scene(s, q, t, m)
    sc.set_texture(ko.TEX_HDRI, textures[0])
    sc.set_texture(ko.TEX_ALBEDO, textures[1])
    p = ko.default_params(width=width, height=height, mode=ko.MODE_PT, max_frames=0, max_bounces=bounces, use_bvh=1)
    cam = ko.default_camera()
    # one untimed row band to page the scene in, then whole frames until the time budget is used
    sc.render_raw(cam, p, 0, rows=(height // 2, height // 2 + 8), nthreads=threads)
    cn = ko.Counters()
    t0 = time.perf_counter()
    nframes = 0
    while nframes < 64:
        sc.render_raw(cam, p, nframes, nthreads=threads, counters=cn)
        nframes += 1
        dt = time.perf_counter() - t0
        if dt + dt / nframes > budget_s:
            break
    dt = time.perf_counter() - t0
    what = "frames 0..%d (all rows)" % (nframes - 1)
    return {"value": cn.paths / dt, "unit": "samples/s", "cores": threads, "kind": "port", "nproc": os.cpu_count(), "affinity_cores": avail,
            "mrays_per_s": cn.rays / dt / 1e6, "seconds": round(dt, 2),
            "sample": "%s of the same %dx%d %d-bounce workload (%d paths, %d rays), oracle C port with its own median-split BVH (NOT the product's SAH tree: the GPU/CPU ratio is not like for like), %d OpenMP threads"
                      % (what, width, height, bounces, cn.paths, cn.rays, threads)}


PRESETS = {
    # BASELINE.json configs[2..4]
    3: dict(width=1920, height=1080, spp=64, blobs=1000, spheres=0, partition="samples",
            name="BASELINE configs[2]: synthetic 1M-triangle textured scene"),
    4: dict(width=3840, height=2160, spp=256, blobs=1000, spheres=0, partition="tiles",
            name="BASELINE configs[3]: synthetic 1M-triangle scene, image tiles"),
    5: dict(width=3840, height=2160, spp=1024, blobs=9990, spheres=10000, partition="samples", scaling="strong",
            name="BASELINE configs[4]: synthetic 10M-primitive mixed sphere/mesh scene, sample-range split"),
}


def sol_inputs():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", SOL_INPUTS)))
    except Exception:
        return None


def speed_of_light(name, kin, units, measured_ms, wave_steps=None):
    """t_min = max(VALU issue, L2 request bytes / L2 rate, fabric bytes / achievable HBM rate, dependent latency) for `units`
    rays of kernel `kin` (per-ray figures from the counter passes under profiles/), against the measured single-lane time."""
    issue_clk = kin["issue_clocks_per_unit"] * units
    lim = {"valu_issue_ms": issue_clk / N_SIMD / PEAK_CLOCK_HZ * 1e3,
           "l2_ms": kin.get("l2_bytes_per_unit", 0.0) * units / 34.5e12 * 1e3,
           "fabric_ms": kin.get("fabric_bytes_per_unit", 0.0) * units / 6.3e12 * 1e3}
    if wave_steps:
        lim["latency_ms"] = wave_steps * STEP_LATENCY_CLOCKS / RESIDENT_TRAVERSAL_WAVES / PEAK_CLOCK_HZ * 1e3
    limiter = max(lim, key=lim.get)
    t_min = lim[limiter]
    return {"kernel": name, "limiter": limiter.replace("_ms", ""), "t_min_ms": round(t_min, 3), "measured_ms": round(measured_ms, 3),
            "frac": round(t_min / measured_ms, 4) if measured_ms else None,
            "limits_ms": {k: round(v, 3) for k, v in lim.items()},
            "issue_clocks_per_ray": round(kin["issue_clocks_per_unit"], 2), "avg_clocks_per_valu_inst": round(kin["avg_clocks_per_valu_inst"], 3),
            "achieved_gclk_per_s": round(issue_clk / (measured_ms * 1e-3) / 1e9, 1) if measured_ms else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(PRESETS), help="BASELINE.json configuration (1-based): 3 = the headline "
                    "(1M triangles, 1080p, 64 spp), 4 = 4K 256 spp image tiles, 5 = 10M primitives 4K 1024 spp sample ranges")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--blobs", type=int, default=0)
    ap.add_argument("--tris-per-blob", type=int, default=1000)
    ap.add_argument("--partition", choices=("samples", "tiles"), default=None,
                    help="N > 1: sample ranges (RCCL all-reduce of sums) or image tiles (row bands, RCCL all-gather)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="N > 1, sample ranges: weak = every rank renders --spp frames of an N*spp image; strong = --spp frames in all, "
                         "split over the ranks. Image tiles are always strong (every rank renders all frames of its row bands)")
    ap.add_argument("--band-rows", type=int, default=8)
    ap.add_argument("--frames-in-flight", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=0, help="batches traced concurrently on separate HIP streams (0 = library default)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="time budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--builder", choices=("host", "gpu"), default="host",
                    help="BVH builder: host = binned SAH on the host cores (headline), gpu = Morton-ordered tree built on the GPU")
    ap.add_argument("--skip-kernels", action="store_true", help="skip the untimed single-lane step that times every kernel with HIP events")
    ap.add_argument("--skip-shared", action="store_true", help="skip the second timed pass (camera-ray hits shared by a batch's frames)")
    args = ap.parse_args()
    pre = PRESETS[args.config]
    width, height = args.width or pre["width"], args.height or pre["height"]
    spp, blobs = args.spp or pre["spp"], args.blobs or pre["blobs"]
    partition_mode = args.partition or pre["partition"]
    scaling = "strong" if partition_mode == "tiles" else (args.scaling or pre.get("scaling", "weak"))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import klrender as kl
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one process per GPU. Rehearsal on a box with fewer GPUs than ranks: KL_BENCH_BACKEND=gloo lets several
    # ranks share a device (RCCL refuses that), everything else being the same code path.
    backend = os.environ.get("KL_BENCH_BACKEND", "nccl")
    device = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    bg, tx, tex_note = load_textures()
    scene = kl.synth_scene(blobs=blobs, tris_per_blob=args.tris_per_blob, analytic_spheres=pre["spheres"], seed=1)
    n_tris, n_spheres = len(scene[2]), len(scene[0])
    r = kl.Renderer(width, height, device=device)
    r.update_params(bvh_builder=kl.BUILD_GPU_LBVH if args.builder == "gpu" else kl.BUILD_HOST_SAH)
    r.upload_scene(*scene)
    r.upload_texture(kl.TEX_HDRI, bg)
    r.upload_texture(kl.TEX_ALBEDO, tx)
    bvh = r.bvh_stats()
    # `value` is measured with every frame's camera ray traced, as the reference shader does, and with no per-launch events in
    # the timed region; the library's default (one camera-ray trace per pixel and batch, shared by the batch's frames: same
    # image) is timed afterwards and reported next to it as `value_shared_camera_rays`
    r.update_params(max_bounces=args.bounces, max_frames=0, frames_in_flight=args.frames_in_flight,
                    batch_lanes=args.lanes, profile_kernels=0, trace_primary_per_frame=1)
    npix = width * height
    gatherer = out_sum = None
    if world > 1:
        from klrender import dist as kdist
        dev = torch.device("cuda", device)
        if partition_mode == "tiles":
            gatherer = kdist.TileGather(height, width, args.band_rows, dev)      # buffers allocated once, outside the timed region
        else:
            out_sum = torch.empty((height, width, 4), dtype=torch.float32, device=dev)
    frames_total = spp * world if (world > 1 and partition_mode == "samples" and scaling == "weak") else spp

    def step():
        # N = 1: frames 0..spp-1 into the running mean, as the reference loop does.
        # N > 1, sample ranges: rank g owns its share of the frame range as a sum; one RCCL all-reduce adds the sums, one divide.
        # N > 1, image tiles: rank g renders every frame of its row bands; one RCCL all-gather + un-tile (klrender/dist.py).
        if world > 1 and partition_mode == "tiles":
            kdist.render_tiles(r, spp, args.band_rows, gatherer=gatherer)
        elif world > 1:
            kdist.render_sample_ranges(r, frames_total, out=out_sum)
        else:
            r.reset()
            r.render_frames(spp, first_frame=0)
        r.synchronize()

    for _ in range(args.warmup):
        step()
    r.reset_counters()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    cn = r.counters()
    rays_rank = cn.rays
    if dist is not None:
        tr = torch.tensor([float(rays_rank)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        rays_total = float(tr.item())
    else:
        rays_total = float(rays_rank)
    samples_total = float(npix) * frames_total * args.steps
    value = samples_total / elapsed

    # Untimed: one more step on a single batch lane (kernels run one after the other, so an event-bracketed launch lasts
    # what the kernel itself takes) with HIP events around every launch and the traversal kernel's own wave-level tallies.
    # On a shard (image tiles) the same, over the rank's rows.
    kernels = None
    if not args.skip_kernels and rank == 0:
        n_k = min(spp, 64)
        # (a) durations: the kernels exactly as timed above; (b) tallies: the same traversal instance with wave-level counting
        # switched on (its ballots make it slower, so its durations are not used)
        r.update_params(batch_lanes=1, profile_kernels=1, collect_stats=0)
        r.reset_counters()
        r.reset()
        r.render_frames(n_k, first_frame=0)
        r.synchronize()
        t1 = r.timing()
        r.update_params(batch_lanes=1, profile_kernels=0, collect_stats=2)
        r.reset_counters()
        r.reset()
        r.render_frames(n_k, first_frame=0)
        r.synchronize()
        c1 = r.counters()
        r.update_params(batch_lanes=args.lanes, profile_kernels=0, collect_stats=0)
        rays1 = max(c1.rays, 1)
        kernels = {"frames": n_k, "rays": int(c1.rays), "paths": int(c1.paths),
                   "intersect_ms": round(t1.intersect_ms, 3), "intersect_launches": int(t1.intersect_launches),
                   "shade_ms": round(t1.shade_ms, 3), "shade_launches": int(t1.shade_launches),
                   "generate_ms": round(t1.generate_ms, 3), "accumulate_ms": round(t1.accumulate_ms, 3),
                   "render_ms": round(t1.render_ms, 3),
                   "nodes_per_ray": c1.nodes / rays1, "tris_per_ray": c1.tri_tests / rays1, "spheres_per_ray": c1.sphere_tests / rays1,
                   "quads_per_ray": c1.quad_tests / rays1, "steps_per_ray": (c1.nodes + c1.tri_tests + c1.sphere_tests + c1.quad_tests) / rays1,
                   "lane_active_frac": c1.lane_steps_active / max(c1.lane_steps_total, 1), "wave_steps": c1.lane_steps_total // 64}

    roofline = None
    sol = sol_inputs()
    if kernels and kernels["intersect_launches"]:
        k = kernels
        n_node, n_tri, n_sph, n_quad = k["nodes_per_ray"], k["tris_per_ray"], k["spheres_per_ray"], k["quads_per_ray"]
        b_ray = 64.0 * n_node + 48.0 * n_tri + 16.0 * n_sph + 48.0 * n_quad + QUEUE_BYTES_PER_RAY
        f_ray = 100.0 * n_node + 45.0 * n_tri + 25.0 * n_sph + 90.0 * n_quad
        isect_s = k["intersect_ms"] * 1e-3
        algo = {"bytes_per_ray": round(b_ray, 1), "gb_per_s": round(k["rays"] * b_ray / isect_s / 1e9, 1), "hbm_peak_gb_per_s": HBM_PEAK_GBS,
                "ratio_to_hbm_peak": round(k["rays"] * b_ray / isect_s / 1e9 / HBM_PEAK_GBS, 4),
                "flop_per_ray": round(f_ray, 1), "tflop_per_s": round(k["rays"] * f_ray / isect_s / 1e12, 2), "valu_peak_tflop_per_s": VALU_PEAK_TFLOPS,
                "note": "algorithmic bytes (64 B per node visited, 48 B per triangle/quad and 16 B per sphere tested, 112 B of queue traffic per "
                        "ray) over the traversal kernel's time. Nodes and triangles of this scene (77 MB) are served from L2 / Infinity Cache, so "
                        "this is NOT a fraction of a ceiling that binds (it can exceed 1); kept as a labelled extra"}
        roofline = {"bound": "valu_issue", "kernel": "k_intersect_bvh", "unit": "G SIMD issue-clocks/s", "peak": N_SIMD * PEAK_CLOCK_HZ / 1e9,
                    "achieved": None, "frac": None, "traffic": None,
                    "rays_per_launch": round(k["rays"] / k["intersect_launches"], 1), "launches": k["intersect_launches"],
                    "avg_launch_ms": round(k["intersect_ms"] / k["intersect_launches"], 4),
                    "grays_per_s_in_kernel": round(k["rays"] / isect_s / 1e9, 3),
                    "nodes_per_ray": round(n_node, 2), "tris_per_ray": round(n_tri, 2), "spheres_per_ray": round(n_sph, 3), "quads_per_ray": round(n_quad, 3),
                    "steps_per_ray": round(k["steps_per_ray"], 2), "lane_active_frac": round(k["lane_active_frac"], 3),
                    "flat_prims_per_ray": int(bvh.flat_spheres + bvh.flat_quads + bvh.flat_tris),
                    "kernel_pass": {kk: k[kk] for kk in ("frames", "rays", "intersect_ms", "shade_ms", "generate_ms", "accumulate_ms", "render_ms")},
                    "kernel_share_of_single_lane_step": round((k["intersect_ms"] + k["shade_ms"] + k["generate_ms"] + k["accumulate_ms"]) / max(k["render_ms"], 1e-9), 3),
                    "hbm_algorithmic": algo,
                    "note": "speed-of-light model: t_min = max(VALU issue clocks / (1024 SIMDs x 2.4 GHz), L2 request bytes / 34.5 TB/s, fabric bytes / "
                            "6.3 TB/s, wave-steps x 300 clocks / 6144 resident waves); frac = t_min / single-lane kernel time (HIP events on the "
                            "kernel's stream, untimed extra step). Per-ray instruction-class, L2 and fabric figures: profiles/" + SOL_INPUTS +
                            " (counter passes over one step of this workload), issue cost per instruction class: profiles/r02_valu_issue.txt"}
        if sol and "intersect" in sol.get("kernels", {}) and args.config == 3:
            ki = sol["kernels"]["intersect"]
            m = speed_of_light("k_intersect_bvh", ki, k["rays"], k["intersect_ms"], k["wave_steps"])
            roofline.update({"achieved": m["achieved_gclk_per_s"], "frac": m["frac"], "limiter": m["limiter"], "limits_ms": m["limits_ms"],
                             "t_min_ms": m["t_min_ms"], "measured_ms": m["measured_ms"], "issue_clocks_per_ray": m["issue_clocks_per_ray"],
                             "avg_clocks_per_valu_inst": m["avg_clocks_per_valu_inst"],
                             "traffic": round(ki.get("fabric_bytes_per_unit", 0.0) * k["rays"] / k["intersect_launches"]),
                             "l2_hit_rate": round(ki.get("l2_hit_rate", 0.0), 3)})
            if "shade" in sol["kernels"]:
                roofline["shade"] = speed_of_light("k_shade", sol["kernels"]["shade"], k["rays"], k["shade_ms"])
                roofline["shade"]["traffic_per_launch"] = round(sol["kernels"]["shade"].get("fabric_bytes_per_unit", 0.0) * k["rays"] / max(k["shade_launches"], 1))
            # the whole step against the sum of its kernels' VALU issue time (two batch lanes overlap kernels, not issue slots)
            issue_all = sum(sol["kernels"][kk]["issue_clocks_per_unit"] * (k["rays"] if sol["kernels"][kk]["unit"] == "ray" else k["paths"])
                            for kk in sol["kernels"]) / N_SIMD / PEAK_CLOCK_HZ * 1e3
            step_ms = 1e3 * elapsed / args.steps
            roofline["step"] = {"valu_issue_ms_all_kernels": round(issue_all * spp / k["frames"], 3), "ms_per_step": round(step_ms, 3),
                                "frac": round(issue_all * spp / k["frames"] / step_ms, 4)}

    # second timed pass: the library default, camera-ray hits shared by the frames of a batch
    elapsed_shared = None
    if not args.skip_shared:
        r.update_params(trace_primary_per_frame=0, profile_kernels=0)
        step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed_shared = max_over_ranks(time.perf_counter() - t0)

    # untimed: what handing the finished image to the host costs (device -> host copy of the float accumulation over PCIe);
    # `value` has inputs and outputs resident in HBM, this is the figure for a caller that reads every step's image back
    readback_ms = None
    if rank == 0:
        r.synchronize()
        r.read_accum()
        t0 = time.perf_counter()
        for _ in range(3):
            r.read_accum()
        readback_ms = 1e3 * (time.perf_counter() - t0) / 3

    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        cpu = cpu_baseline(scene, (bg, tx), width, height, args.bounces, args.cpu_seconds)

    if rank == 0:
        if world == 1:
            part = "single GPU, running mean"
        elif partition_mode == "tiles":
            part = "image tiles: row bands of %d rows round-robin over %d ranks, every rank renders all %d frames of its bands, one %s all-gather + un-tile per step" % (
                args.band_rows, world, spp, "RCCL" if backend == "nccl" else backend)
        else:
            part = "sample ranges: %d frames per rank (%d in all), one %s all-reduce(sum) + divide per step" % (
                frames_total // world, frames_total, "RCCL" if backend == "nccl" else backend)
        out = {
            "metric": "path samples/sec @%s 8-bounce" % ("1080p" if height == 1080 else "%dx%d" % (width, height)), "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "mrays_per_s": rays_total / elapsed / 1e6, "rays_per_sample": rays_total / samples_total,
            "value_shared_camera_rays": samples_total / elapsed_shared if elapsed_shared else None,
            "ms_per_step_shared_camera_rays": 1e3 * elapsed_shared / args.steps if elapsed_shared else None,
            "host_readback_ms": round(readback_ms, 3), "value_with_host_readback": samples_total / (elapsed + 1e-3 * readback_ms * args.steps),
            "config": {"workload": "%s, %dx%d, %d spp, %d bounces, path-traced" % (pre["name"], width, height, frames_total, args.bounces),
                       "triangles": int(n_tris), "analytic_spheres": int(n_spheres), "bvh_nodes": int(bvh.nodes), "bvh_max_depth": int(bvh.max_depth),
                       "bvh_build_s": round(bvh.build_seconds, 4),
                       "bvh_builder": "GPU LBVH" if bvh.builder == kl.BUILD_GPU_LBVH else "host binned SAH", "frames_in_flight": int(r.get_params().frames_in_flight),
                       "textures": tex_note, "partition": part},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

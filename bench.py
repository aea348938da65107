#!/usr/bin/env python3
"""Headline benchmark: path samples/s at 1080p, 8 bounces, on N MI355X.

Workload (BASELINE.json configs[2]): the synthetic 1M-triangle textured scene, 1920x1080,
64 spp, bounce cap 8, path-traced mode. One step = one full pass: 64 frames traced and
accumulated (132.7 M path samples). Scene, BVH and textures are resident in HBM before the
timed region. With N > 1 every rank renders its own 64-frame sample range of the same image
(weak scaling, sample-range partition) and the per-rank sums are added with one RCCL
all-reduce per step, inside the timed region.

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
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    threads = max(1, min(threads, 16))
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
    return {"value": cn.paths / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "mrays_per_s": cn.rays / dt / 1e6, "seconds": round(dt, 2),
            "sample": "%s of the same %dx%d %d-bounce workload (%d paths, %d rays), oracle C port with its own median-split BVH, %d OpenMP threads"
                      % (what, width, height, bounces, cn.paths, cn.rays, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--blobs", type=int, default=1000)
    ap.add_argument("--tris-per-blob", type=int, default=1000)
    ap.add_argument("--frames-in-flight", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=0, help="batches traced concurrently on separate HIP streams (0 = library default)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="time budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--builder", choices=("host", "gpu"), default="host",
                    help="BVH builder: host = binned SAH on the host cores (headline), gpu = Morton-ordered tree built on the GPU")
    ap.add_argument("--skip-isolated", action="store_true", help="skip the extra untimed single-lane step behind roofline.isolated")
    ap.add_argument("--skip-shared", action="store_true", help="skip the second timed pass (camera-ray hits shared by a batch's frames)")
    ap.add_argument("--no-profile", action="store_true", help="skip per-launch HIP events around the intersect kernel")
    args = ap.parse_args()

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
    scene = kl.synth_scene(blobs=args.blobs, tris_per_blob=args.tris_per_blob, seed=1)
    n_tris = len(scene[2])
    r = kl.Renderer(args.width, args.height, device=device)
    r.update_params(bvh_builder=kl.BUILD_GPU_LBVH if args.builder == "gpu" else kl.BUILD_HOST_SAH)
    r.upload_scene(*scene)
    r.upload_texture(kl.TEX_HDRI, bg)
    r.upload_texture(kl.TEX_ALBEDO, tx)
    bvh = r.bvh_stats()
    # `value` is measured with every frame's camera ray traced, as the reference shader does; the library's default
    # (one camera-ray trace per pixel and batch, shared by the batch's frames: same image) is timed afterwards and
    # reported next to it as `value_shared_camera_rays`
    r.update_params(max_bounces=args.bounces, max_frames=0, frames_in_flight=args.frames_in_flight,
                    batch_lanes=args.lanes, profile_kernels=0 if args.no_profile else 1, trace_primary_per_frame=1)
    npix = args.width * args.height
    if world > 1:
        from klrender import dist as kdist

    def step():
        # N = 1: frames 0..spp-1 into the running mean, as the reference loop does.
        # N > 1: rank g owns frames [g*spp, (g+1)*spp) of a world*spp-sample image; the per-rank
        #        sums are added with one RCCL all-reduce and divided (klrender/dist.py).
        if world > 1:
            kdist.render_sample_ranges(r, world * args.spp)
        else:
            r.reset()
            r.render_frames(args.spp, first_frame=0)
        r.synchronize()

    # algorithmic bytes per ray: one instrumented, untimed pass over a few frames
    r.update_params(collect_stats=1)
    r.reset_counters()
    r.reset()
    r.render_frames(min(4, args.spp), first_frame=rank * args.spp)
    cs = r.counters()
    r.update_params(collect_stats=0)
    rays_stat = max(cs.rays, 1)
    # per ray: nodes visited and primitives tested INSIDE the traversal kernel. The flat primitives (room
    # walls: tested for every ray by the separate coherent pass, through uniform loads) are not part of it.
    n_node = cs.nodes / rays_stat
    n_tri = cs.tri_tests / rays_stat - bvh.flat_tris
    n_sph = cs.sphere_tests / rays_stat - bvh.flat_spheres
    n_quad = cs.quad_tests / rays_stat - bvh.flat_quads
    b_ray = 64.0 * n_node + 48.0 * n_tri + 16.0 * n_sph + 48.0 * n_quad + QUEUE_BYTES_PER_RAY
    f_ray = 100.0 * n_node + 45.0 * n_tri + 25.0 * n_sph + 90.0 * n_quad
    lane_active = cs.lane_steps_active / max(cs.lane_steps_total, 1)

    for _ in range(args.warmup):
        step()
    r.reset_counters()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    cn = r.counters()
    tm = r.timing()
    rays_rank = cn.rays
    if dist is not None:
        tr = torch.tensor([float(rays_rank)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        rays_total = float(tr.item())
    else:
        rays_total = float(rays_rank)
    samples_total = float(npix) * args.spp * args.steps * world
    value = samples_total / elapsed

    # With the library's default of two batch lanes the traversal launches of the two lanes share the GPU with each
    # other's shade launches, so a launch bracketed by events lasts longer than the same launch alone. One more step,
    # untimed, on a single lane gives the kernel's own figure next to the timed region's.
    isolated = None
    if tm.intersect_launches and r.get_params().batch_lanes != 1 and not args.skip_isolated:
        r.update_params(batch_lanes=1)
        r.reset_counters()
        step()
        fence()
        cn1, tm1 = r.counters(), r.timing()
        if tm1.intersect_launches:
            s1 = tm1.intersect_ms * 1e-3
            isolated = {"lanes": 1, "launches": int(tm1.intersect_launches), "rays_per_launch": round(cn1.rays / tm1.intersect_launches, 1),
                        "avg_launch_ms": round(tm1.intersect_ms / tm1.intersect_launches, 4),
                        "achieved": round(cn1.rays * b_ray / s1 / 1e9, 1), "frac": round(cn1.rays * b_ray / s1 / 1e9 / HBM_PEAK_GBS, 4),
                        "grays_per_s_in_kernel": round(cn1.rays / s1 / 1e9, 3),
                        "valu_tflops": round(cn1.rays * f_ray / s1 / 1e12, 2)}
        r.update_params(batch_lanes=args.lanes)

    roofline = None
    if tm.intersect_launches:
        isect_s = tm.intersect_ms * 1e-3
        achieved = rays_rank * b_ray / isect_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("fabric_bytes_per_intersect_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "k_intersect_bvh", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "bytes_per_ray": round(b_ray, 1), "nodes_per_ray": round(n_node, 2), "tris_per_ray": round(n_tri, 2),
                    "spheres_per_ray": round(n_sph, 3), "quads_per_ray": round(n_quad, 3),
                    "flat_prims_per_ray": int(bvh.flat_spheres + bvh.flat_quads + bvh.flat_tris),
                    "note": "algorithmic bytes (64 B per node visited, 48 B per triangle/quad and 16 B per sphere tested, 112 B of "
                            "queue traffic per ray) / HIP-event time of k_intersect_bvh; nodes + triangles of this scene fit in "
                            "L2/Infinity Cache, so this figure may exceed the HBM peak: `traffic` is what actually crossed the fabric",
                    "rays_per_launch": round(rays_rank / tm.intersect_launches, 1),
                    "launches": int(tm.intersect_launches), "avg_launch_ms": round(tm.intersect_ms / tm.intersect_launches, 4),
                    "intersect_share_of_step": round(isect_s / elapsed, 3), "lane_active_frac": round(lane_active, 3),
                    "grays_per_s_in_kernel": round(rays_rank / isect_s / 1e9, 3),
                    "batch_lanes_param": int(r.get_params().batch_lanes),
                    "timing_note": "timed region = library default (two batch lanes): traversal launches run beside the other lane's kernels, so an "
                                   "event-bracketed launch lasts longer than the same launch alone; `isolated` = one untimed step on a single lane",
                    "isolated": isolated,
                    # the other ceiling of this kernel: vector fp32. 100 flop per 4-wide node (two 2-child slab tests),
                    # 45 per triangle, 25 per sphere, 90 per quad
                    "valu": {"flop_per_ray": round(f_ray, 1), "achieved": round(rays_rank * f_ray / isect_s / 1e12, 2),
                             "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(rays_rank * f_ray / isect_s / 1e12 / VALU_PEAK_TFLOPS, 4)}}

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
        elapsed_shared = time.perf_counter() - t0
        if dist is not None:
            te = torch.tensor([elapsed_shared], dtype=torch.float64, device="cuda")
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            elapsed_shared = float(te.item())

    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        cpu = cpu_baseline(scene, (bg, tx), args.width, args.height, args.bounces, args.cpu_seconds)

    if rank == 0:
        out = {
            "metric": "path samples/sec @1080p 8-bounce", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "mrays_per_s": rays_total / elapsed / 1e6, "rays_per_sample": rays_total / samples_total,
            "value_shared_camera_rays": samples_total / elapsed_shared if elapsed_shared else None,
            "ms_per_step_shared_camera_rays": 1e3 * elapsed_shared / args.steps if elapsed_shared else None,
            "config": {"workload": "BASELINE configs[2]: synthetic 1M-triangle textured scene, %dx%d, %d spp, %d bounces, path-traced"
                                   % (args.width, args.height, args.spp, args.bounces),
                       "triangles": int(n_tris), "bvh_nodes": int(bvh.nodes), "bvh_max_depth": int(bvh.max_depth),
                       "bvh_build_s": round(bvh.build_seconds, 4),
                       "bvh_builder": "GPU LBVH" if bvh.builder == kl.BUILD_GPU_LBVH else "host binned SAH", "frames_in_flight": int(r.get_params().frames_in_flight),
                       "textures": tex_note,
                       "partition": "single GPU, running mean" if world == 1 else
                                    "sample ranges: %d frames per rank, %s all-reduce(sum) per step" % (args.spp, "RCCL" if backend == "nccl" else backend)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: path samples/s at 1080p, 8 bounces, on N MI355X.

Workload (BASELINE.json configs[2], `--config 3`): the synthetic 1M-triangle textured scene, 1920x1080, 64 spp, bounce cap 8,
path-traced mode. One step = one full pass: 64 frames traced and accumulated (132.7 M path samples). Scene, BVH and textures are
resident in HBM before the timed region, and no per-launch events are recorded inside it. `--config 4` / `--config 5` select
configs[3] (3840x2160, 256 spp, image tiles) and configs[4] (10 M primitives, 4K, 1024 spp, sample ranges).

N > 1 (one process per GPU, torch.distributed "nccl" = RCCL): rank 0 builds the tree once and one broadcast hands the packed
scene to the other ranks (klrender.dist.upload_scene_once). `--partition samples` (default): every rank renders its frame
range of the same image as a sum, one all-reduce + divide per step (`--scaling weak`: spp frames per rank, N*spp in all;
`--scaling strong`: spp frames in all); `--partition tiles`: every rank renders all frames of its row bands, one all-gather +
un-tile per step. The exchange is inside the timed region; its buffers are allocated before it.

After the timed region (none of it enters `value`):
  N = 1: one single-lane step with HIP events around every kernel launch and one with the traversal kernel's own wave-level
  tallies (`roofline`), the library-default pass with camera rays shared by a batch's frames (`value_shared_camera_rays`), the
  CPU oracle on a bounded sample (`cpu_baseline`), and BASELINE configs[0..1] on the built-in scene with the oracle port timed
  beside each (`builtin`).
  N > 1: the other BASELINE multi-GPU lines as sub-records of `multi_gpu`, each with its own ms_per_step and exchange time:
  `strong` (configs[2], 64 spp in all, split over the ranks), `config4_tiles` (configs[3]: 4K, 256 spp, image tiles, all-gather),
  `config5_samples` (configs[4]: 10 M primitives, 4K, 1024 spp in all, sample ranges, all-reduce).

`roofline` names three ceilings for the dominant kernel (the BVH traversal) and says which binds: FP32 vector flops
(algorithmic flops per ray x rays / kernel time / 157.3 TFLOP/s), HBM (fabric bytes from the counter passes / kernel time /
8 TB/s; the algorithmic bytes are given beside it and are served by the caches), and vector-instruction issue (executed
instructions by class x measured issue clocks / 1024 SIMDs x 2.4 GHz). The per-ray instruction and traffic figures come from
counter passes kept under profiles/; each is stamped with the hash of the kernel's machine code it was measured on, and
is used only if the library loaded here holds the same code (`stale_sol_inputs` otherwise).

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
HBM_ACHIEVABLE_GBS = 6300.0    # what streaming kernels reach
L2_PEAK_GBS = 34500.0          # aggregate of the eight XCDs' L2s
VALU_PEAK_TFLOPS = 157.3       # MI355X vector fp32 peak (packed multiply-add on every lane of 256 CUs at 2.4 GHz)
QUEUE_BYTES_PER_RAY = 112      # per ray and bounce: the queue entry (ray, hit, throughput) written once and read twice, as first budgeted;
                               # the current layout moves 48 B written + 32 B (traversal) + 48 B (shade) read + 8 B for one ray in five
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9
# per-ray instruction, L2 and fabric figures from the counter passes kept under profiles/, one file per BASELINE configuration
SOL_INPUTS = {3: "r03_sol_inputs.json", 4: "r03_config4_sol_inputs.json", 5: "r03_config5_sol_inputs.json"}
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


def host_threads():
    """every core this process may use: OMP_NUM_THREADS when the box sets it (its share of the machine), else the affinity mask"""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, int(os.environ.get("OMP_NUM_THREADS", "0")) or avail), avail


def oracle():
    """the CPU oracle (oracle/kl_oracle.c): test infrastructure, used here as a reported baseline only"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kloracle as ko
    return ko


def cpu_baseline(scene_arrays, textures, width, height, bounces, budget_s):
    """Times the CPU oracle (oracle/kl_oracle.c, OpenMP) on the host cores over a bounded sample of
    the same workload: whole frames 0, 1, ... for as long as the time budget lasts. Test infrastructure
    used as a reported baseline only; nothing it computes enters the GPU result."""
    ko = oracle()
    threads, avail = host_threads()
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
            "sample": "%s of the same %dx%d %d-bounce workload (%d paths, %d rays), oracle C port with its own median-split BVH (NOT the "
                      "product's SAH tree: the GPU/CPU ratio is not like for like), %d OpenMP threads"
                      % (what, width, height, bounces, cn.paths, cn.rays, threads)}


def builtin_figures(kl, device, textures):
    """BASELINE configs[0] and configs[1] on this GPU with the oracle C port timed on this host beside each: the reference's
    built-in scene (RawFrame.glsl:171-192), default camera, the reference's two images, bounce cap 32 as shipped --
    path traced 640x360 16 spp; ZDepth and AO 1920x1080 1 spp. Untimed extras: none of it enters `value`."""
    ko = oracle()
    threads, _ = host_threads()
    cam = ko.default_camera()
    sc = ko.Scene(builtin=True)
    sc.set_texture(ko.TEX_HDRI, textures[0])
    sc.set_texture(ko.TEX_ALBEDO, textures[1])
    cases = (("config1_pathtraced_640x360_16spp", 640, 360, kl.MODE_PT, ko.MODE_PT, 16, 20),
             ("config2_zdepth_1920x1080_1spp", 1920, 1080, kl.MODE_Z, ko.MODE_Z, 1, 50),
             ("config2_ao_1920x1080_1spp", 1920, 1080, kl.MODE_AO, ko.MODE_AO, 1, 50))
    out = {}
    for name, w, h, mode, omode, frames, reps in cases:
        with kl.Renderer(w, h, device=device) as r:
            r.load_builtin_scene()
            r.upload_texture(kl.TEX_HDRI, textures[0])
            r.upload_texture(kl.TEX_ALBEDO, textures[1])
            r.set_mode(mode)
            r.update_params(max_bounces=32, max_frames=0, trace_primary_per_frame=1)
            r.render_frames(frames, first_frame=0)
            r.synchronize()
            r.reset_counters()
            t0 = time.perf_counter()
            for _ in range(reps):
                r.reset()
                r.render_frames(frames, first_frame=0)
            r.synchronize()
            gpu_s = (time.perf_counter() - t0) / reps
            cn = r.counters()
            gpu_rays = cn.rays / reps
        p = ko.default_params(width=w, height=h, mode=omode, max_frames=0, max_bounces=32)
        sc.render_raw(cam, p, 0, rows=(h // 2, h // 2 + 8), nthreads=threads)
        ocn = ko.Counters()
        t0 = time.perf_counter()
        ko.render_mean(sc, cam, p, frames, nthreads=threads, counters=ocn)
        cpu_s = time.perf_counter() - t0
        samples = float(w) * h * frames
        out[name] = {"gpu_ms": round(gpu_s * 1e3, 4), "gpu_samples_per_s": samples / gpu_s, "gpu_mrays_per_s": gpu_rays / gpu_s / 1e6,
                     "rays_per_sample": gpu_rays / samples, "cpu_port_ms": round(cpu_s * 1e3, 2), "cpu_port_samples_per_s": samples / cpu_s,
                     "cpu_port_mrays_per_s": ocn.rays / cpu_s / 1e6, "cpu_port_rays": int(ocn.rays), "gpu_rays": int(round(gpu_rays))}
    out["cpu_port"] = "oracle/kl_oracle.c (kind: port), %d OpenMP threads on this host, flat loop over the 13 primitives as the shader does" % threads
    out["reference_shader"] = ("the reference's own GLSL on Mesa llvmpipe cannot run on this box (it does not travel); measured in the build container "
                               "(8 vCPU): profiles/llvmpipe_baseline.json -- 4.7 M samples/s on config 1, 46 M (ZDepth) / 35 M (AO) on config 2")
    return out


PRESETS = {
    # BASELINE.json configs[2..4]
    3: dict(width=1920, height=1080, spp=64, blobs=1000, spheres=0, partition="samples",
            name="BASELINE configs[2]: synthetic 1M-triangle textured scene"),
    4: dict(width=3840, height=2160, spp=256, blobs=1000, spheres=0, partition="tiles",
            name="BASELINE configs[3]: synthetic 1M-triangle scene, image tiles"),
    5: dict(width=3840, height=2160, spp=1024, blobs=9990, spheres=10000, partition="samples", scaling="strong",
            name="BASELINE configs[4]: synthetic 10M-primitive mixed sphere/mesh scene, sample-range split"),
}


def sol_inputs(config):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", SOL_INPUTS[config])))
    except Exception:
        return None


def kernel_model(name, kin, units, measured_ms, wave_steps=None):
    """Ceilings of one kernel from its per-unit counter figures `kin` (profiles/*_sol_inputs.json): fraction of the VALU issue
    rate, of the L2 rate, of the HBM peak (fabric bytes) and of a dependent-latency floor, each = the time that ceiling alone
    would need / the measured single-lane kernel time."""
    t = measured_ms * 1e-3
    if t <= 0:
        return None
    per_unit = kin["issue_clocks_per_unit"]
    ratio = units / float(kin["issue_clocks"] / per_unit)          # this pass's units / the counter pass's units
    sec = lambda clocks: clocks * ratio / N_SIMD / PEAK_CLOCK_HZ   # noqa: E731
    out = {"kernel": name, "measured_ms": round(measured_ms, 3),
           "frac_valu_issue": round(sec(kin["issue_clocks"]) / t, 4),
           "frac_valu_issue_if_all_2clk": round(sec(kin["issue_clocks_if_rest_all_2clk"]) / t, 4),
           "frac_valu_issue_if_all_4clk": round(sec(kin["issue_clocks_if_rest_all_4clk"]) / t, 4),
           "issue_clocks_per_unit": round(per_unit, 2), "valu_insts_per_unit": round(kin["valu_insts"] * per_unit / kin["issue_clocks"], 2),
           "avg_clocks_per_valu_inst": round(kin["avg_clocks_per_valu_inst"], 3),
           "achieved_gclk_per_s": round(per_unit * units / t / 1e9, 1)}
    if "fabric_bytes_per_unit" in kin:
        fb = kin["fabric_bytes_per_unit"] * units
        out.update({"fabric_bytes_per_unit": round(kin["fabric_bytes_per_unit"], 1), "fabric_gb_per_s": round(fb / t / 1e9, 1),
                    "frac_hbm_fabric": round(fb / t / 1e9 / HBM_PEAK_GBS, 4), "frac_hbm_fabric_of_achievable": round(fb / t / 1e9 / HBM_ACHIEVABLE_GBS, 4),
                    "l2_hit_rate": round(kin.get("l2_hit_rate", 0.0), 3),
                    "frac_l2": round(kin.get("l2_bytes_per_unit", 0.0) * units / t / 1e9 / L2_PEAK_GBS, 4)})
    if wave_steps:
        out["frac_step_latency"] = round(wave_steps * STEP_LATENCY_CLOCKS / RESIDENT_TRAVERSAL_WAVES / PEAK_CLOCK_HZ / t, 4)
    return out


def build_roofline(k, bvh, sol, code_now, config, sol_file):
    """The `roofline` record of the dominant kernel (BVH traversal) from the single-lane kernel pass `k`."""
    n_node, n_tri, n_sph, n_quad = k["nodes_per_ray"], k["tris_per_ray"], k["spheres_per_ray"], k["quads_per_ray"]
    # algorithmic figures: 64 B and 100 flop per 4-wide node visited (2 x the 50 flop of a two-child node), 48 B / 45 flop per
    # triangle, 16 B / 25 flop per sphere, 48 B / 90 flop per quad tested, 112 B of queue traffic per ray
    b_ray = 64.0 * n_node + 48.0 * n_tri + 16.0 * n_sph + 48.0 * n_quad + QUEUE_BYTES_PER_RAY
    f_ray = 100.0 * n_node + 45.0 * n_tri + 25.0 * n_sph + 90.0 * n_quad
    t = k["intersect_ms"] * 1e-3
    rays = k["rays"]
    tflops = rays * f_ray / t / 1e12
    algo_gbs = rays * b_ray / t / 1e9
    rl = {"kernel": "k_intersect_bvh", "bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
          "frac_fp32_flops": round(tflops / VALU_PEAK_TFLOPS, 4), "fp32_tflop_per_s": round(tflops, 2), "flop_per_ray": round(f_ray, 1),
          "frac_hbm_fabric": None, "frac_valu_issue": None, "frac_valu_issue_if_all_2clk": None,
          "hbm_algorithmic": {"bytes_per_ray": round(b_ray, 1), "gb_per_s": round(algo_gbs, 1), "ratio_to_hbm_peak": round(algo_gbs / HBM_PEAK_GBS, 4),
                              "note": "algorithmic bytes over the traversal kernel's time. Nodes and primitives are re-read from L1 / L2 / Infinity "
                                      "Cache, so this is NOT a fraction of a ceiling that binds (it can exceed 1); frac_hbm_fabric is the measured traffic"},
          "rays_per_launch": round(rays / k["intersect_launches"], 1), "launches": k["intersect_launches"],
          "avg_launch_ms": round(k["intersect_ms"] / k["intersect_launches"], 4), "grays_per_s_in_kernel": round(rays / t / 1e9, 3),
          "nodes_per_ray": round(n_node, 2), "tris_per_ray": round(n_tri, 2), "spheres_per_ray": round(n_sph, 3), "quads_per_ray": round(n_quad, 3),
          "steps_per_ray": round(k["steps_per_ray"], 2), "lane_active_frac": round(k["lane_active_frac"], 3),
          "flat_prims_per_ray": int(bvh.flat_spheres + bvh.flat_quads + bvh.flat_tris),
          "kernel_pass": {kk: k[kk] for kk in ("frames", "rays", "paths", "intersect_ms", "shade_ms", "generate_ms", "accumulate_ms", "render_ms")},
          "kernel_share_of_single_lane_step": round((k["intersect_ms"] + k["shade_ms"] + k["generate_ms"] + k["accumulate_ms"]) / max(k["render_ms"], 1e-9), 3),
          "code": code_now, "sol_inputs": None, "stale_sol_inputs": None}
    ceilings = {"fp32_valu_flops": (rl["frac_fp32_flops"], tflops, VALU_PEAK_TFLOPS, "TFLOP/s")}
    usable = bool(sol) and sol.get("config", 3) == config and "intersect" in sol.get("kernels", {})
    if usable:
        ki = sol["kernels"]["intersect"]
        stale = {kk: (sol["kernels"][kk].get("code") != code_now.get(kk) or not code_now.get(kk)) for kk in sol["kernels"]}
        rl["sol_inputs"] = "profiles/" + sol_file
        rl["stale_sol_inputs"] = bool(stale.get("intersect"))
        if not stale.get("intersect"):
            m = kernel_model("k_intersect_bvh", ki, rays, k["intersect_ms"], k["wave_steps"])
            rl.update({kk: m[kk] for kk in m if kk not in ("kernel", "measured_ms")})
            rl["traffic"] = round(ki.get("fabric_bytes_per_unit", 0.0) * rays / k["intersect_launches"])
            ceilings["hbm_fabric"] = (m.get("frac_hbm_fabric"), m.get("fabric_gb_per_s"), HBM_PEAK_GBS, "GB/s")
            ceilings["valu_issue"] = (m["frac_valu_issue"], m["achieved_gclk_per_s"], N_SIMD * PEAK_CLOCK_HZ / 1e9, "G SIMD issue-clocks/s")
        if "shade" in sol["kernels"]:
            if stale.get("shade"):
                rl["shade"] = {"kernel": "k_shade", "measured_ms": round(k["shade_ms"], 3), "stale_sol_inputs": True}
            else:
                rl["shade"] = kernel_model("k_shade", sol["kernels"]["shade"], rays, k["shade_ms"])
                rl["shade"]["traffic_per_launch"] = round(sol["kernels"]["shade"].get("fabric_bytes_per_unit", 0.0) * rays / max(k["shade_launches"], 1))
        if not any(stale.values()):
            # the single-lane step against the sum of its kernels' VALU issue time
            issue_all = sum(sol["kernels"][kk]["issue_clocks_per_unit"] * (rays if sol["kernels"][kk]["unit"] == "ray" else k["paths"])
                            for kk in sol["kernels"]) / N_SIMD / PEAK_CLOCK_HZ * 1e3
            rl["valu_issue_ms_all_kernels"] = round(issue_all, 3)
    # the ceiling that binds = the one the kernel is closest to
    live = {n: c for n, c in ceilings.items() if c[0] is not None}
    bound = max(live, key=lambda n: live[n][0])
    rl.update({"bound": bound, "frac": live[bound][0], "achieved": round(live[bound][1], 2), "peak": live[bound][2], "unit": live[bound][3]})
    rl["note"] = ("three ceilings for the traversal kernel, each = what the kernel does per second / the chip's rate, over the single-lane kernel "
                  "time (HIP events on the kernel's stream, untimed extra step): frac_fp32_flops (algorithmic flops, live tallies), frac_hbm_fabric "
                  "(bytes through the fabric per ray from the counter passes x live rays, against the 8 TB/s HBM peak), frac_valu_issue (executed "
                  "VALU instructions by class x measured issue clocks, against 1024 SIMDs x 2.4 GHz; _if_all_2clk prices every unclassed "
                  "instruction at the guide's 2 clocks). `frac` is the largest of them, `bound` names it. Counter figures are used only if "
                  "the kernel's machine code in the loaded library hashes to the value stamped in the file (stale_sol_inputs otherwise).")
    return rl


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
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="time budget of the CPU baseline leg (0 = skip; also skips the built-in scene figures)")
    ap.add_argument("--builder", choices=("host", "gpu"), default="host",
                    help="BVH builder: host = binned SAH on the host cores (headline), gpu = Morton-ordered tree built on the GPU")
    ap.add_argument("--skip-kernels", action="store_true", help="skip the untimed single-lane step that times every kernel with HIP events")
    ap.add_argument("--skip-shared", action="store_true", help="skip the second timed pass (camera-ray hits shared by a batch's frames)")
    ap.add_argument("--extras", choices=("auto", "none", "small"), default="auto",
                    help="N > 1: the other BASELINE multi-GPU lines as untimed sub-records (auto: at the BASELINE sizes when the run itself is the "
                         "default configuration; small: the same code at toy sizes, for tests; none)")
    ap.add_argument("--skip-extras", action="store_true", help="same as --extras none, and no built-in scene figures")
    args = ap.parse_args()
    if args.skip_extras:
        args.extras = "none"
    pre = PRESETS[args.config]
    width, height = args.width or pre["width"], args.height or pre["height"]
    spp, blobs = args.spp or pre["spp"], args.blobs or pre["blobs"]
    partition_mode = args.partition or pre["partition"]
    scaling = "strong" if partition_mode == "tiles" else (args.scaling or pre.get("scaling", "weak"))
    default_run = not (args.width or args.height or args.spp or args.blobs) and args.config == 3

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import klrender as kl
    from klrender import codeid
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one process per GPU. Rehearsal on a box with fewer GPUs than ranks: KL_BENCH_BACKEND=gloo lets several
    # ranks share a device (RCCL refuses that), everything else being the same code path.
    backend = os.environ.get("KL_BENCH_BACKEND", "nccl")
    device = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    dist = kdist = None
    if world > 1:
        import torch.distributed as dist
        from klrender import dist as kdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    collective = "RCCL" if backend == "nccl" else backend

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

    def sum_over_ranks(x):
        if dist is None:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def make_scene(r, n_blobs, n_spheres, tris_per_blob, builder):
        """uploads the synthetic scene: one rank builds the tree, the others receive the packed scene (N > 1).
        Returns (scene arrays or None, packed-scene block or None, seconds spent)."""
        t0 = time.perf_counter()
        scene = block = None
        r.update_params(bvh_builder=kl.BUILD_GPU_LBVH if builder == "gpu" else kl.BUILD_HOST_SAH)
        if rank == 0 or dist is None:
            scene = kl.synth_scene(blobs=n_blobs, tris_per_blob=tris_per_blob, analytic_spheres=n_spheres, seed=1)
        if dist is None:
            r.upload_scene(*scene)
        else:
            block = kdist.upload_scene_once(r, scene, src=0)
        return scene, block, time.perf_counter() - t0

    bg, tx, tex_note = load_textures()
    r = kl.Renderer(width, height, device=device)
    scene, scene_block, scene_s = make_scene(r, blobs, pre["spheres"], args.tris_per_blob, args.builder)
    r.upload_texture(kl.TEX_HDRI, bg)
    r.upload_texture(kl.TEX_ALBEDO, tx)
    bvh = r.bvh_stats()
    n_tris, n_spheres = int(bvh.tris), int(bvh.spheres)
    # `value` is measured with every frame's camera ray traced, as the reference shader does, and with no per-launch events in
    # the timed region; the library's default (one camera-ray trace per pixel and batch, shared by the batch's frames: same
    # image) is timed afterwards and reported next to it as `value_shared_camera_rays`
    r.update_params(max_bounces=args.bounces, max_frames=0, frames_in_flight=args.frames_in_flight,
                    batch_lanes=args.lanes, profile_kernels=0, trace_primary_per_frame=1)
    npix = width * height

    def partition_runner(rr, mode, frames_all, h, w):
        """step function of one multi-rank partition of `frames_all` frames on renderer rr (buffers allocated here, once)"""
        if mode == "tiles":
            tg = kdist.TileGather(h, w, args.band_rows, dev)
            return lambda timing=None: kdist.render_tiles(rr, frames_all, args.band_rows, gatherer=tg, timing=timing)
        out_sum = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
        return lambda timing=None: kdist.render_sample_ranges(rr, frames_all, out=out_sum, timing=timing)

    frames_total = spp * world if (world > 1 and partition_mode == "samples" and scaling == "weak") else spp
    multi_step = partition_runner(r, partition_mode, frames_total, height, width) if world > 1 else None

    def step():
        # N = 1: frames 0..spp-1 into the running mean, as the reference loop does.
        # N > 1, sample ranges: rank g owns its share of the frame range as a sum; one RCCL all-reduce adds the sums, one divide.
        # N > 1, image tiles: rank g renders every frame of its row bands; one RCCL all-gather + un-tile (klrender/dist.py).
        if world > 1:
            multi_step()
        else:
            r.reset()
            r.render_frames(spp, first_frame=0)
        r.synchronize()

    for _ in range(args.warmup):
        step()
    r.reset_counters()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    cn = r.counters()
    rays_total = sum_over_ranks(cn.rays)
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
    if kernels and kernels["intersect_launches"]:
        code_now = codeid.render_code_identity(kl.LIB_PATH, codeid.traversal_types(bvh))
        roofline = build_roofline(kernels, bvh, sol_inputs(args.config), code_now, args.config, SOL_INPUTS[args.config])
        if roofline.get("valu_issue_ms_all_kernels") is not None and world == 1:
            # the timed (two-lane) step against the sum of its kernels' VALU issue time (lanes overlap kernels, not issue slots)
            step_ms = 1e3 * elapsed / args.steps
            issue_step = roofline["valu_issue_ms_all_kernels"] * spp / kernels["frames"]
            roofline["step"] = {"valu_issue_ms_all_kernels": round(issue_step, 3), "ms_per_step": round(step_ms, 3), "frac": round(issue_step / step_ms, 4)}

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
        r.update_params(trace_primary_per_frame=1)

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

    # ---- the line, as far as the timed region and its single-GPU extras go. It is built BEFORE the remaining extras and printed
    # exactly once: by the normal path with the extras' sub-records filled in, or -- if an extra raises or does not come back
    # within its time budget (a collective that one rank left) -- by the guard below, with the reason in place of the sub-record.
    out = None
    if rank == 0:
        if world == 1:
            part = "single GPU, running mean"
        elif partition_mode == "tiles":
            part = "image tiles: row bands of %d rows round-robin over %d ranks, every rank renders all %d frames of its bands, one %s all-gather + un-tile per step" % (
                args.band_rows, world, spp, collective)
        else:
            part = "sample ranges: %d frames per rank (%d in all), one %s all-reduce(sum) + divide per step" % (
                frames_total // world, frames_total, collective)
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
                       "triangles": n_tris, "analytic_spheres": n_spheres, "bvh_nodes": int(bvh.nodes), "bvh_max_depth": int(bvh.max_depth),
                       "bvh_build_s": round(bvh.build_seconds, 4),
                       "bvh_builder": "GPU LBVH" if bvh.builder == kl.BUILD_GPU_LBVH else "host binned SAH",
                       "scene_upload": ("one build, this GPU" if world == 1 else "tree built once on rank 0, packed scene broadcast to %d ranks over %s" % (world - 1, collective)),
                       "scene_seconds": round(scene_s, 3), "textures": tex_note, "partition": part},
            "roofline": roofline, "cpu_baseline": None, "builtin": None, "multi_gpu": None,
        }
    import threading
    emitted = threading.Lock()

    def emit(**extra):
        """prints the line once (rank 0), whoever gets here first"""
        if not emitted.acquire(blocking=False):
            return
        if out is not None:
            out.update(extra)
            print(json.dumps(out))
            sys.stdout.flush()

    def give_up(reason):
        """an extra did not finish: the line goes out without it and the process ends here (the other ranks' guards do the same;
        a collective that a rank has left cannot be waited for)"""
        emit(multi_gpu={"error": reason})
        sys.stderr.write("bench.py: %s\n" % reason)
        sys.stderr.flush()
        os._exit(0)

    # ---- N > 1: the other BASELINE multi-GPU lines, untimed extras (every rank takes part in their collectives)
    multi = None
    if world > 1 and args.extras != "none" and (default_run or args.extras == "small"):
        small = args.extras == "small"
        budget_s = float(os.environ.get("KL_BENCH_EXTRAS_SECONDS", "180"))
        guard = threading.Timer(budget_s, give_up, args=("the multi-GPU extras did not finish within %.0f s" % budget_s,))
        guard.daemon = True
        guard.start()

        try:
            def run_case(rr, mode, frames_all, h, w, steps):
                runner = partition_runner(rr, mode, frames_all, h, w)
                runner()                                      # warm-up (buffers, first touch)
                rr.synchronize()
                rr.reset_counters()
                fence()
                t0 = time.perf_counter()
                for _ in range(steps):
                    runner()
                    rr.synchronize()
                fence()
                dt = max_over_ranks(time.perf_counter() - t0)
                rays = sum_over_ranks(rr.counters().rays)
                # one more step with a wait between render and exchange, to time the exchange by itself
                tm = {}
                runner(timing=tm)
                rr.synchronize()
                return {"samples_per_s": float(h) * w * frames_all * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
                        "mrays_per_s": rays / dt / 1e6, "frames_in_all": frames_all, "frames_per_rank": frames_all if mode == "tiles" else frames_all / world,
                        "exchange_ms": 1e3 * max_over_ranks(tm.get("exchange_s", 0.0)), "render_ms_rank_max": 1e3 * max_over_ranks(tm.get("render_s", 0.0)),
                        "rccl_ranks": dist.get_world_size(), "backend": collective, "partition": mode, "image": "%dx%d" % (w, h),
                        "exchange": ("all-gather of %.1f MB per rank + un-tile" % (h * w * 16 / world / 1e6)) if mode == "tiles"
                                    else ("all-reduce(sum) of %.1f MB + divide" % (h * w * 16 / 1e6))}

            multi = {"note": "untimed extras (not in `value`): the other BASELINE multi-GPU lines, each timed like the main region (barrier + "
                             "synchronize on both sides, max over ranks); exchange_ms = the collective and its un-tile / divide alone, from one "
                             "further step with a wait between render and exchange"}
            if os.environ.get("KL_BENCH_TEST_FAIL_EXTRAS") == str(rank):      # test hook: this rank leaves the extras
                raise RuntimeError("requested by KL_BENCH_TEST_FAIL_EXTRAS")
            # (1) strong scaling of the headline workload: spp frames in all, split over the ranks
            multi["strong"] = run_case(r, "samples", spp, height, width, args.steps)
            multi["strong"]["workload"] = "%s, %dx%d, %d spp in all, sample ranges" % (pre["name"], width, height, spp)
            r.close()
            r = None
            # (2) BASELINE configs[3]: the same scene at 4K, 256 spp, image tiles; the packed scene is imported again, not rebuilt
            w4, h4, spp4 = (3840, 2160, 256) if not small else (2 * width, 2 * height, 2 * spp)
            r4 = kl.Renderer(w4, h4, device=device)
            r4.scene_import(scene_block.data_ptr(), scene_block.numel())
            r4.upload_texture(kl.TEX_HDRI, bg)
            r4.upload_texture(kl.TEX_ALBEDO, tx)
            r4.update_params(max_bounces=args.bounces, max_frames=0, trace_primary_per_frame=1)
            multi["config4_tiles"] = run_case(r4, "tiles", spp4, h4, w4, 1 if not small else 2)
            multi["config4_tiles"]["workload"] = "%s, %dx%d, %d spp, row bands of %d rows round-robin" % (PRESETS[4]["name"], w4, h4, spp4, args.band_rows)
            r4.close()
            scene_block = None
            # (3) BASELINE configs[4]: 10 M primitives at 4K, 1024 spp in all, sample ranges; one host SAH build on rank 0, broadcast
            w5, h5, spp5 = (3840, 2160, 1024) if not small else (2 * width, 2 * height, 4 * spp)
            blobs5, spheres5 = (PRESETS[5]["blobs"], PRESETS[5]["spheres"]) if not small else (3 * blobs, 50)
            r5 = kl.Renderer(w5, h5, device=device)
            _, block5, scene5_s = make_scene(r5, blobs5, spheres5, args.tris_per_blob, args.builder)
            r5.upload_texture(kl.TEX_HDRI, bg)
            r5.upload_texture(kl.TEX_ALBEDO, tx)
            r5.update_params(max_bounces=args.bounces, max_frames=0, trace_primary_per_frame=1)
            b5 = r5.bvh_stats()
            multi["config5_samples"] = run_case(r5, "samples", spp5, h5, w5, 1 if not small else 2)
            multi["config5_samples"].update({"workload": "%s, %dx%d, %d spp in all" % (PRESETS[5]["name"], w5, h5, spp5),
                                             "primitives": int(b5.tris + b5.spheres + b5.quads), "bvh_nodes": int(b5.nodes),
                                             "scene_seconds_rank0_build_plus_broadcast": round(scene5_s, 2), "packed_scene_mb": round(block5.numel() / 1e6, 1)})
            r5.close()
            block5 = None

        except Exception as e:          # noqa: BLE001 -- whatever it was, the timed result must still be reported
            give_up("the multi-GPU extras failed on rank %d: %s: %s" % (rank, type(e).__name__, e))
        guard.cancel()
    cpu = builtin = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        try:
            cpu = cpu_baseline(scene, (bg, tx), width, height, args.bounces, args.cpu_seconds)
        except Exception as e:          # noqa: BLE001
            cpu = {"error": "%s: %s" % (type(e).__name__, e)}
        if args.extras != "none":
            try:
                builtin = builtin_figures(kl, device, (bg, tx))
            except Exception as e:      # noqa: BLE001
                builtin = {"error": "%s: %s" % (type(e).__name__, e)}

    emit(cpu_baseline=cpu, builtin=builtin, multi_gpu=multi)
    if r is not None:
        r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

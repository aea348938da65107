// kl_bench: headless benchmark / render tool over the C ABI (the reference has no headless mode; its
// only figure of merit is the 1/frame_time it prints in the window title, GLSL_Renderer.cpp:350-355).
//
//   kl_bench --scene builtin --mode pt --w 640 --h 360 --frames 16 --out frame.png
//   kl_bench --scene synth --blobs 1000 --tris-per-blob 1000 --w 1920 --h 1080 --frames 64 --bounces 8
//   kl_bench --scene synth --builder gpu ...         (tree built on the GPU instead of the host cores)
//   kl_bench --scene builtin --interactive-demo      (drives the GLSL_Renderer-shaped host class)
//   kl_bench --scene builtin --mode pt --w 3840 --h 2019 --frames 40 --shipped --pos 10.58809 4.05715 29.47826 --yaw -112.42518 --pitch -0.82758
//            --hdri Background.png --albedo Texture.png --out s4.png
//       a view of the application as shipped: camera from position / yaw / pitch (Camera::f_compile), SAMPLES freeze at 30 frames,
//       8-bit attachments (this line reproduces resources/Screenshots/S4.png, whose bottom 45 rows the capture lost)
//   kl_bench --decode in.png out.raw                 (the host's image loader alone: writes the 8-bit pixels, prints "w h channels")
//   kl_bench --scene synth --gpus 8 --partition tiles --w 3840 --h 2160 --frames 256 --bounces 8      (BASELINE configs[3])
//   kl_bench --scene synth --gpus 8 --partition samples --blobs 9990 --spheres 10000 --w 3840 --h 2160 --frames 1024 --bounces 8
//       several GPUs from this one process (kl_group_*): image tiles + RCCL all-gather, or sample ranges + RCCL all-reduce;
//       --exchange copy replaces RCCL by device copies and lets --devices name one GPU several times (rehearsal)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "Math/kl_render.h"
#include "kl_host.hpp"

using namespace klhost;

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(call) do { kl_status s_ = (call); if (s_ != KL_OK) { std::fprintf(stderr, "%s: %s (%s)\n", #call, kl_status_string(s_), ctx ? kl_last_error(ctx) : ""); return 1; } } while (0)

int main(int argc, char **argv) {
    std::string scene = "builtin", mode = "pt", out, hdri, albedo, albedo_raw, decode_in, decode_out, builder = "host", partition = "tiles", exchange = "rccl", devlist;
    uint32_t gpus = 0, band_rows = 8;
    int araw[3] = {0, 0, 0};
    uint32_t w = 1920, h = 1080, frames = 16, bounces = 0, blobs = 1000, tpb = 1000, spheres = 0, warmup = 1, device = 0, fif = 0;
    bool demo = false, shipped = false, have_cam = false;
    double cam_pos[3] = {0, 1, 5}, cam_yaw = -90, cam_pitch = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--scene") scene = next(); else if (a == "--mode") mode = next();
        else if (a == "--w") w = std::atoi(next()); else if (a == "--h") h = std::atoi(next());
        else if (a == "--frames") frames = std::atoi(next()); else if (a == "--bounces") bounces = std::atoi(next());
        else if (a == "--blobs") blobs = std::atoi(next()); else if (a == "--tris-per-blob") tpb = std::atoi(next());
        else if (a == "--spheres") spheres = std::atoi(next()); else if (a == "--warmup") warmup = std::atoi(next());
        else if (a == "--device") device = std::atoi(next()); else if (a == "--frames-in-flight") fif = std::atoi(next());
        else if (a == "--out") out = next(); else if (a == "--hdri") hdri = next(); else if (a == "--albedo") albedo = next();
        else if (a == "--albedo-raw") { albedo_raw = next(); for (int k = 0; k < 3; k++) araw[k] = std::atoi(next()); }
        else if (a == "--builder") builder = next();
        else if (a == "--gpus") gpus = std::atoi(next()); else if (a == "--partition") partition = next();
        else if (a == "--exchange") exchange = next(); else if (a == "--devices") devlist = next(); else if (a == "--band-rows") band_rows = std::atoi(next());
        else if (a == "--interactive-demo") demo = true;
        else if (a == "--shipped") shipped = true;
        else if (a == "--pos") { for (int k = 0; k < 3; k++) cam_pos[k] = std::atof(next()); have_cam = true; }
        else if (a == "--yaw") { cam_yaw = std::atof(next()); have_cam = true; }
        else if (a == "--pitch") { cam_pitch = std::atof(next()); have_cam = true; }
        else if (a == "--decode") { decode_in = next(); decode_out = next(); }
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (!decode_in.empty()) {
        Image img;
        if (!img.f_load(decode_in)) { std::fprintf(stderr, "cannot decode %s\n", decode_in.c_str()); return 1; }
        FILE *f = std::fopen(decode_out.c_str(), "wb");
        if (!f || std::fwrite(img.data.data(), 1, img.data.size(), f) != img.data.size()) { std::fprintf(stderr, "cannot write %s\n", decode_out.c_str()); return 1; }
        std::fclose(f);
        std::printf("%d %d %d\n", img.width, img.height, img.channels);
        return 0;
    }
    if (demo) {
        // the reference's session in miniature: accumulate, press V (ZDepth), strafe with D, accumulate again
        GLSL_Renderer r;
        r.iResolution[0] = w; r.iResolution[1] = h;
        if (!r.f_init((int)device, hdri, albedo)) { std::fprintf(stderr, "%s\n", r.error.c_str()); return 1; }
        if (!r.f_run(frames)) { std::fprintf(stderr, "%s\n", r.error.c_str()); return 1; }
        std::printf("{\"iFrame\": %zu, \"mode\": %d}\n", r.iFrame, (int)r.render_mode);
        r.key_callback(KEY_V, PRESS); r.key_callback(KEY_V, RELEASE);
        r.key_callback(KEY_D, PRESS); r.f_frame(); r.key_callback(KEY_D, RELEASE);
        if (!r.f_run(2)) { std::fprintf(stderr, "%s\n", r.error.c_str()); return 1; }
        std::printf("{\"iFrame\": %zu, \"mode\": %d, \"camera_x\": %.3f}\n", r.iFrame, (int)r.render_mode, r.camera.position.x);
        if (!out.empty() && !r.f_save_png(out)) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
        return 0;
    }
    if (gpus > 0) {
        // one process, `gpus` contexts: scene and textures to every member, frames split by tiles or sample ranges, one exchange
        std::vector<int32_t> devs;
        if (!devlist.empty()) { for (size_t p0 = 0; p0 < devlist.size();) { devs.push_back(std::atoi(devlist.c_str() + p0)); size_t c = devlist.find(',', p0); if (c == std::string::npos) break; p0 = c + 1; } }
        else for (uint32_t k = 0; k < gpus; k++) devs.push_back((int32_t)k);
        gpus = (uint32_t)devs.size();
        kl_ctx *ctx = nullptr;        // (CK prints a context's error text; group errors are printed below)
        kl_group *g = nullptr;
        const uint32_t split = (partition == "samples" ? KL_SPLIT_SAMPLES : KL_SPLIT_TILES) | (exchange == "copy" ? KL_EXCHANGE_COPY : 0u);
        CK(kl_group_create(&g, devs.data(), gpus, split, w, h, band_rows));
#define GK(call) do { kl_status s_ = (call); if (s_ != KL_OK) { std::fprintf(stderr, "%s: %s (%s)\n", #call, kl_status_string(s_), kl_group_last_error(g)); return 1; } } while (0)
        kl_params p; kl_default_params(&p);
        p.max_frames = 0; if (bounces) p.max_bounces = bounces; p.frames_in_flight = fif;
        p.bvh_builder = builder == "gpu" ? KL_BUILD_GPU_LBVH : KL_BUILD_HOST_SAH;
        GK(kl_group_set_params(g, &p));
        if (scene == "builtin") GK(kl_group_load_builtin_scene(g));
        else {
            kl_synth_params sp = {blobs, tpb, spheres, 1};
            uint32_t ns, nq, nt, nm;
            CK(kl_synth_scene_counts(&sp, &ns, &nq, &nt, &nm));
            std::vector<kl_sphere> s(ns); std::vector<kl_quad> q(nq); std::vector<kl_tri> t(nt); std::vector<kl_material> m(nm);
            CK(kl_synth_scene_fill(&sp, s.data(), q.data(), t.data(), m.data()));
            GK(kl_group_upload_scene(g, s.data(), ns, q.data(), nq, t.data(), nt, m.data(), nm));
        }
        Image img;
        if (!hdri.empty()) { if (!img.f_load(hdri)) { std::fprintf(stderr, "cannot read %s\n", hdri.c_str()); return 1; } GK(kl_group_upload_texture(g, KL_TEX_HDRI, img.data.data(), img.width, img.height, img.channels)); }
        if (!albedo.empty()) { if (!img.f_load(albedo)) { std::fprintf(stderr, "cannot read %s\n", albedo.c_str()); return 1; } GK(kl_group_upload_texture(g, KL_TEX_ALBEDO, img.data.data(), img.width, img.height, img.channels)); }
        GK(kl_group_set_mode(g, mode == "ao" ? KL_MODE_AO : (mode == "z" ? KL_MODE_Z : KL_MODE_PT)));
        std::vector<float> image((size_t)w * h * 4);
        for (uint32_t i = 0; i < warmup; i++) { GK(kl_group_reset(g)); GK(kl_group_render_frames(g, frames, 0)); GK(kl_group_gather(g, nullptr, 0)); }
        GK(kl_group_reset(g));
        double t0 = now_s();
        GK(kl_group_render_frames(g, frames, 0));
        GK(kl_group_gather(g, nullptr, 0));          // the exchange is inside the timed region; the copy to the host is not
        double dt = now_s() - t0;
        GK(kl_group_gather(g, image.data(), image.size() * sizeof(float)));
        uint64_t rays = 0, paths = 0;
        for (uint32_t k = 0; k < gpus; k++) { kl_ctx *mc = nullptr; kl_counters cn; GK(kl_group_member(g, k, &mc)); CK(kl_get_counters(mc, &cn)); rays += cn.rays; paths += cn.paths; }
        double sum = 0.0; for (size_t i = 0; i < image.size(); i += 4) sum += image[i] + image[i + 1] + image[i + 2];
        std::printf("{\"scene\": \"%s\", \"gpus\": %u, \"partition\": \"%s\", \"exchange\": \"%s\", \"w\": %u, \"h\": %u, \"frames\": %u, \"seconds\": %.6f, "
                    "\"samples_per_s\": %.1f, \"mrays_per_s\": %.2f, \"image_mean\": %.9g}\n", scene.c_str(), gpus, partition.c_str(), exchange.c_str(), w, h, frames, dt,
                    (double)w * h * frames / dt, (double)rays / (warmup + 1.0) / dt / 1e6, sum / (3.0 * w * h));
        (void)paths;
        if (!out.empty()) {
            Image o; o.width = (int)w; o.height = (int)h; o.channels = 4; o.data.resize((size_t)w * h * 4);
            for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w * 4; x++) {
                float v = image[(size_t)y * w * 4 + x]; v = !(v >= 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);
                o.data[(size_t)(h - 1 - y) * w * 4 + x] = (uint8_t)(v * 255.0f + 0.5f);
            }
            if (!o.f_save_png(out)) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
        }
        kl_group_destroy(g);
        return 0;
    }
    kl_ctx *ctx = nullptr;
    CK(kl_create(&ctx, (int)device, w, h));
    kl_bvh_stats bs; std::memset(&bs, 0, sizeof bs);
    kl_params p; kl_default_params(&p);
    if (!shipped) p.max_frames = 0;              // --shipped keeps #define SAMPLES 30: frames from the 30th on change nothing
    if (bounces) p.max_bounces = bounces;
    p.frames_in_flight = fif;
    p.bvh_builder = builder == "gpu" ? KL_BUILD_GPU_LBVH : KL_BUILD_HOST_SAH;      // read by the upload below
    CK(kl_set_params(ctx, &p));
    if (shipped) CK(kl_set_accum_mode(ctx, KL_ACCUM_MEAN_UNORM8));      // the application's unsized GL_RGBA attachments (FBT.cpp:6)
    if (scene == "builtin") {
        CK(kl_load_builtin_scene(ctx));
    } else {
        kl_synth_params sp = {blobs, tpb, spheres, 1};
        uint32_t ns, nq, nt, nm;
        CK(kl_synth_scene_counts(&sp, &ns, &nq, &nt, &nm));
        std::vector<kl_sphere> s(ns); std::vector<kl_quad> q(nq); std::vector<kl_tri> t(nt); std::vector<kl_material> m(nm);
        CK(kl_synth_scene_fill(&sp, s.data(), q.data(), t.data(), m.data()));
        CK(kl_upload_scene(ctx, s.data(), ns, q.data(), nq, t.data(), nt, m.data(), nm));
    }
    CK(kl_get_bvh_stats(ctx, &bs));
    Image img;
    if (!hdri.empty()) { if (!img.f_load(hdri)) { std::fprintf(stderr, "cannot read %s\n", hdri.c_str()); return 1; } CK(kl_upload_texture(ctx, KL_TEX_HDRI, img.data.data(), img.width, img.height, img.channels)); }
    if (!albedo.empty()) { if (!img.f_load(albedo)) { std::fprintf(stderr, "cannot read %s\n", albedo.c_str()); return 1; } CK(kl_upload_texture(ctx, KL_TEX_ALBEDO, img.data.data(), img.width, img.height, img.channels)); }
    if (!albedo_raw.empty()) { if (!img.f_load_raw(albedo_raw, araw[0], araw[1], araw[2])) { std::fprintf(stderr, "cannot read %s\n", albedo_raw.c_str()); return 1; } CK(kl_upload_texture(ctx, KL_TEX_ALBEDO, img.data.data(), img.width, img.height, img.channels)); }
    CK(kl_set_mode(ctx, mode == "ao" ? KL_MODE_AO : (mode == "z" ? KL_MODE_Z : KL_MODE_PT)));
    if (have_cam) {
        Camera cam;                                  // the reference's camera state: position + yaw / pitch (Camera.cpp:26-43)
        cam.position = {cam_pos[0], cam_pos[1], cam_pos[2]};
        cam.rotation = {cam_yaw, cam_pitch, 0.0};
        cam.f_compile();
        const kl_camera kc = cam.f_upload();
        CK(kl_set_camera(ctx, &kc));
    }
    for (uint32_t i = 0; i < warmup; i++) { CK(kl_reset(ctx)); CK(kl_render_frames(ctx, frames, 0)); }
    CK(kl_synchronize(ctx)); CK(kl_reset_counters(ctx)); CK(kl_reset(ctx));
    double t0 = now_s();
    CK(kl_render_frames(ctx, frames, 0));
    CK(kl_synchronize(ctx));
    double dt = now_s() - t0;
    kl_counters cn; CK(kl_get_counters(ctx, &cn));
    std::printf("{\"scene\": \"%s\", \"mode\": \"%s\", \"w\": %u, \"h\": %u, \"frames\": %u, \"bounces\": %u, \"seconds\": %.6f, "
                "\"samples_per_s\": %.1f, \"mrays_per_s\": %.2f, \"rays_per_sample\": %.3f, \"bvh_nodes\": %u, \"tris\": %u, "
                "\"bvh_builder\": \"%s\", \"bvh_build_s\": %.4f}\n",
                scene.c_str(), mode.c_str(), w, h, frames, p.max_bounces, dt, (double)cn.paths / dt, (double)cn.rays / dt / 1e6,
                (double)cn.rays / (double)(cn.paths ? cn.paths : 1), bs.nodes, bs.tris,
                bs.builder == KL_BUILD_GPU_LBVH ? "gpu" : "host", bs.build_seconds);
    if (!out.empty()) {
        std::vector<uint8_t> px((size_t)w * h * 4);
        CK(kl_resolve(ctx, KL_TONEMAP_CLAMP, px.data(), px.size()));
        Image o; o.width = (int)w; o.height = (int)h; o.channels = 4; o.data.resize(px.size());
        for (uint32_t y = 0; y < h; y++) std::memcpy(&o.data[(size_t)w * 4 * (h - 1 - y)], &px[(size_t)w * 4 * y], (size_t)w * 4);
        if (!o.f_save_png(out)) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    }
    kl_destroy(ctx);
    return 0;
}

// Sanitizer driver for the host-side code that handles caller data (CPU only, no GPU): the mesh loaders on the files named
// on the command line (well-formed or not: a refused file is fine, a sanitizer report is not), then the SAH builder, the
// 4-wide collapse and the synthetic scene generator over what they produced. Built by `make asan` with
// -fsanitize=address,undefined; tools/asan/run.py feeds it malformed inputs.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "Math/kl_render.h"
#include "kl_bvh.hpp"
#include "kl_scene.hpp"
#include "kl_types.hpp"

int main(int argc, char **argv) {
    size_t loaded = 0, refused = 0, tris_total = 0;
    std::vector<kl_tri> all;
    for (int i = 1; i < argc; i++) {
        kl_tri *t = nullptr; uint32_t n = 0;
        kl_status s = kl_load_mesh(argv[i], 0, 4, &t, &n);
        if (s == KL_OK) { loaded++; tris_total += n; if (all.size() + n < 200000) all.insert(all.end(), t, t + n); kl_free(t); }
        else { refused++; if (!kl_load_mesh_error()[0]) { std::fprintf(stderr, "%s refused without a message\n", argv[i]); return 3; } }
    }
    // the builders over the loaded triangles plus the built-in and a synthetic scene
    std::vector<kl_sphere> sph; std::vector<kl_quad> quads; std::vector<kl_material> mats;
    kl::builtin_scene(sph, quads, mats);
    kl_synth_params sp = {6, 200, 9, 7};
    uint32_t ns, nq, nt, nm;
    kl::synth_counts(sp, ns, nq, nt, nm);
    std::vector<kl_sphere> s2(ns); std::vector<kl_quad> q2(nq); std::vector<kl_tri> t2(nt); std::vector<kl_material> m2(nm);
    kl::synth_fill(sp, s2.data(), q2.data(), t2.data(), m2.data());
    t2.insert(t2.end(), all.begin(), all.end());
    for (unsigned threads : {1u, 4u}) {
        kl::HostBVH bvh;
        kl::build_bvh(s2.data(), ns, q2.data(), nq, t2.data(), (uint32_t)t2.size(), bvh, threads, 0.02f, 32);
        kl::collapse_bvh4(bvh);
        if (bvh.nodes4.size() % 16 != 0 || bvh.tri_order.size() != t2.size()) { std::fprintf(stderr, "inconsistent tree\n"); return 4; }
        kl::HostBVH small;
        kl::build_bvh(sph.data(), (uint32_t)sph.size(), quads.data(), (uint32_t)quads.size(), nullptr, 0, small, threads, 0.0f, 32);
        kl::collapse_bvh4(small);
    }
    std::printf("{\"loaded\": %zu, \"refused\": %zu, \"triangles\": %zu}\n", loaded, refused, tris_total);
    return 0;
}

#!/usr/bin/env python3
"""Wall time of kl_upload_scene (tree build + record packing + transfers) for both builders."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import klrender as kl

for blobs, spheres in ((1000, 0), (9990, 10000)):
    t0 = time.perf_counter()
    scene = kl.synth_scene(blobs=blobs, tris_per_blob=1000, analytic_spheres=spheres, seed=1)
    gen = time.perf_counter() - t0
    with kl.Renderer(64, 36) as r:
        for name, b in (("gpu", kl.BUILD_GPU_LBVH), ("host", kl.BUILD_HOST_SAH), ("gpu", kl.BUILD_GPU_LBVH)):
            r.update_params(bvh_builder=b)
            t0 = time.perf_counter()
            r.upload_scene(*scene)
            r.synchronize()
            dt = time.perf_counter() - t0
            st = r.bvh_stats()
            print("%8d prims  builder %-4s  upload %.3f s  (tree %.3f s, %d nodes, depth %d)  [scene generation %.2f s]"
                  % (st.tris + st.spheres + st.quads, name, dt, st.build_seconds, st.nodes, st.max_depth, gen), flush=True)

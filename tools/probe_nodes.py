#!/usr/bin/env python3
"""Traversal statistics of the bench workload (instrumented kernel): nodes, tests, lane use per ray."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import klrender as kl
scene = kl.synth_scene(blobs=1000, tris_per_blob=1000, seed=1)
r = kl.Renderer(1920, 1080)
r.upload_scene(*scene)
r.update_params(max_bounces=8, max_frames=0, trace_primary_per_frame=1, collect_stats=1)
r.render_frames(4, first_frame=0)
c = r.counters()
print("rays %d nodes/ray %.3f tri/ray %.3f  sph-slot/ray %.4f quad-slot/ray %.4f lane %.3f" % (
    c.rays, c.nodes / c.rays, c.tri_tests / c.rays, c.sphere_tests / c.rays, c.quad_tests / c.rays, c.lane_steps_active / max(1, c.lane_steps_total)))
st = r.bvh_stats()
print("flat", st.flat_spheres, st.flat_quads, st.flat_tris, "nodes", st.nodes)

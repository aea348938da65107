#!/usr/bin/env python3
"""The bench workload, once, for counter passes: BASELINE configs[2] (1M-triangle scene, 1920x1080, 8 bounces), exactly ONE
64-spp step with every frame's camera ray traced, library-default batching. Prints the step's ray and path counts as JSON so
that counter totals can be divided into per-ray figures. No statistics pass, no warm-up: every kernel launch in the trace
belongs to the step."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, ROOT)
import klrender as kl  # noqa: E402
from bench import load_textures  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bg, tx, _ = load_textures()
scene = kl.synth_scene(blobs=1000, tris_per_blob=1000, seed=1)
r = kl.Renderer(1920, 1080)
r.upload_scene(*scene)
r.upload_texture(kl.TEX_HDRI, bg)
r.upload_texture(kl.TEX_ALBEDO, tx)
r.update_params(max_bounces=8, max_frames=0, trace_primary_per_frame=1, batch_lanes=lanes)
r.reset_counters()
r.render_frames(spp, first_frame=0)
r.synchronize()
c = r.counters()
print(json.dumps({"spp": spp, "paths": c.paths, "rays": c.rays, "batch_lanes": lanes}))
r.close()

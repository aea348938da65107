#!/usr/bin/env python3
"""The bench workload, once, for counter passes: BASELINE configs[2] (1M-triangle scene, 1920x1080, 8 bounces), exactly ONE
64-spp step with every frame's camera ray traced, library-default batching. Prints the step's ray and path counts as JSON so
that counter totals can be divided into per-ray figures. No statistics pass, no warm-up: every kernel launch in the trace
belongs to the step.

usage: pmc_workload.py [spp [lanes [config]]]; config = 3 (default), 4 or 5 as in bench.py --config (scene and image size of
BASELINE configs[2..4]); the environment variables KL_PMC_SPP / KL_PMC_LANES / KL_PMC_CONFIG give the same three when the
command line cannot (tools/pmc_pass.sh)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, ROOT)
import klrender as kl  # noqa: E402
from bench import PRESETS, load_textures  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else int(os.environ.get("KL_PMC_SPP", "64"))
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else int(os.environ.get("KL_PMC_LANES", "0"))
config = int(sys.argv[3]) if len(sys.argv) > 3 else int(os.environ.get("KL_PMC_CONFIG", "3"))
pre = PRESETS[config]
bg, tx, _ = load_textures()
scene = kl.synth_scene(blobs=pre["blobs"], tris_per_blob=1000, analytic_spheres=pre["spheres"], seed=1)
r = kl.Renderer(pre["width"], pre["height"])
if os.environ.get("KL_PMC_BUILDER", "host") == "gpu":
    r.update_params(bvh_builder=kl.BUILD_GPU_LBVH)
r.upload_scene(*scene)
r.upload_texture(kl.TEX_HDRI, bg)
r.upload_texture(kl.TEX_ALBEDO, tx)
r.update_params(max_bounces=8, max_frames=0, trace_primary_per_frame=1, batch_lanes=lanes)
r.reset_counters()
r.render_frames(spp, first_frame=0)
r.synchronize()
c = r.counters()
from klrender import codeid  # noqa: E402
types = codeid.traversal_types(r.bvh_stats())
print(json.dumps({"spp": spp, "paths": c.paths, "rays": c.rays, "batch_lanes": lanes, "config": config,
                  "width": pre["width"], "height": pre["height"], "primitives": len(scene[0]) + len(scene[1]) + len(scene[2]),
                  "traversal_types": types, "code": codeid.render_code_identity(kl.LIB_PATH, types)}))
r.close()

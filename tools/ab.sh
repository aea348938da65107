#!/bin/bash
# usage: tools/ab.sh <variant> [<variant> ...] : one short bench per library variant (tools/build_variant.sh), two rounds interleaved
for round in 1 2; do
for v in "$@"; do
  KLRENDER_LIB=$GRAFT_REPO_ROOT/path-tracer_amd/lib/variants/libklrender_$v.so timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; k=r['kernel_pass']
print('%-12s %7.1f Msamples/s (shared %7.1f)  single lane: isect %.2f ms shade %.2f ms generate %.2f ms  %.3f Grays/s in-kernel  lane %.3f  steps/ray %.2f' % ('$v', j['value']/1e6, (j['value_shared_camera_rays'] or 0)/1e6, k['intersect_ms'], k['shade_ms'], k['generate_ms'], r['grays_per_s_in_kernel'], r['lane_active_frac'], r['steps_per_ray']))"
done; done

#!/bin/bash
# usage: tools/sweep_args.sh "<bench.py args>" ...   one short bench per argument set
for a in "$@"; do
  timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --cpu-seconds 0 --skip-shared $a 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-30s %7.1f Msamples/s  isect %.3f ms  nodes %.2f tris %.2f  bvh_nodes %d depth %d build %.4fs (%s)' % ('$a', j['value']/1e6, r['avg_launch_ms'], r['nodes_per_ray'], r['tris_per_ray'], j['config']['bvh_nodes'], j['config']['bvh_max_depth'], j['config']['bvh_build_s'], j['config']['bvh_builder']))" || exit 1
done

#!/bin/bash
# usage: [BENCH_ARGS="--builder gpu"] tools/sweep_env.sh "VAR=x VAR2=y" ...   one short bench per environment setting, two rounds
for round in 1 2; do
for e in "$@"; do
  ( export $e; timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --cpu-seconds 0 $BENCH_ARGS 2>/dev/null ) | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; k=r['kernel_pass']
print('%-34s %7.1f Msamples/s (shared %7.1f)  single lane: isect %.2f shade %.2f ms  nodes %.2f tris %.2f steps %.2f lane %.3f' % ('$e', j['value']/1e6, (j['value_shared_camera_rays'] or 0)/1e6, k['intersect_ms'], k['shade_ms'], r['nodes_per_ray'], r['tris_per_ray'], r['steps_per_ray'], r['lane_active_frac']))" || exit 1
done; done

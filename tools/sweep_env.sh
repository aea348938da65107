#!/bin/bash
# usage: [BENCH_ARGS="--builder gpu"] tools/sweep_env.sh "VAR=x VAR2=y" ...   one short bench per environment setting
for e in "$@"; do
  ( export $e; timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --cpu-seconds 0 $BENCH_ARGS 2>/dev/null ) | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-44s %7.1f Msamples/s  isect %.3f ms  nodes %.2f tris %.2f  bvh_nodes %d build %.2fs' % ('$e', j['value']/1e6, r['avg_launch_ms'], r['nodes_per_ray'], r['tris_per_ray'], j['config']['bvh_nodes'], j['config']['bvh_build_s']))" || exit 1
done

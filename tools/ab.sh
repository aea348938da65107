#!/bin/bash
# usage: tools/ab.sh <variant> [<variant> ...] : one short bench per library variant, two rounds interleaved
for round in 1 2; do
for v in "$@"; do
  KLRENDER_LIB=$GRAFT_REPO_ROOT/path-tracer_amd/lib/variants/libklrender_$v.so timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-12s %7.1f Msamples/s  isect %.3f ms/launch  %.3f Grays/s in-kernel  share %.2f  lane %.3f' % ('$v', j['value']/1e6, r['avg_launch_ms'], r['grays_per_s_in_kernel'], r['intersect_share_of_step'], r['lane_active_frac']))"
done; done

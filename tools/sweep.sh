#!/bin/bash
# usage: tools/sweep.sh "<bench args A>" "<bench args B>" ...   one short bench per argument set
for a in "$@"; do
  timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --cpu-seconds 0 $a 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-40s %7.1f Msamples/s  %.2f ms/step  isect %.3f ms/launch  fif %d' % ('$a', j['value']/1e6, j['ms_per_step'], r['avg_launch_ms'], j['config']['frames_in_flight']))" || exit 1
done

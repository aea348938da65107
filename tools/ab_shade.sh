#!/bin/bash
# per-kernel totals for library variants: usage tools/ab_shade.sh <variant>...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  rm -rf gpurun_out/abs_$v
  lib=$GRAFT_REPO_ROOT/path-tracer_amd/lib/variants/libklrender_$v.so
  [ "$v" = default ] && lib=$GRAFT_REPO_ROOT/path-tracer_amd/lib/libklrender.so
  KLRENDER_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/abs_$v -o x -- python3 bench.py --steps 1 --warmup 0 --spp 16 --cpu-seconds 0 --no-profile > /dev/null 2>&1
  echo "== $v"; python3 tools/rocpd_summary.py gpurun_out/abs_$v/x_results.db | cut -c1-60,200-400 | head -6
done

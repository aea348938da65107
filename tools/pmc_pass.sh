#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counter> [<counter> ...]
# One rocprofv3 counter pass over a short run of the benchmark workload (one 64-spp step);
# per-kernel results land in gpurun_out/pmc_<tag>/ as CSV.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -o $tag -- \
  python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --skip-shared --no-profile > gpurun_out/pmc_$tag.log 2>&1
echo "pmc pass $tag exit $?"

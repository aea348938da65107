#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counter> [<counter> ...]
# One rocprofv3 counter pass over ONE 64-spp step of the benchmark workload (tools/pmc_workload.py);
# per-kernel results land in gpurun_out/pmc_<tag>/ as CSV, the step's ray count in gpurun_out/pmc_<tag>.json.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -o $tag -- \
  python3 tools/pmc_workload.py > gpurun_out/pmc_$tag.json 2> gpurun_out/pmc_$tag.log
echo "pmc pass $tag exit $?"

#!/bin/bash
# usage: tools/profile_all.sh <tag>
# The measurement set behind profiles/: the benchmark command under rocprofv3 --kernel-trace --stats,
# then three separate counter passes (fetch bytes, write bytes, fabric read requests) over a short run
# of the same workload. Every step has its own time limit; the first failure ends the script.
set -e
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- \
  python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-shared --skip-isolated > $out/bench_under_rocprof.json 2> $out/trace.log
python3 tools/rocpd_summary.py $(find $out/trace -name '*.db' | head -1) > $out/kernel_stats.md
tools/pmc_pass.sh ${tag}_fetch FETCH_SIZE
tools/pmc_pass.sh ${tag}_write WRITE_SIZE
tools/pmc_pass.sh ${tag}_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
{ for p in fetch write rd; do echo "### pass $p"; python3 tools/pmc_summary.py $(find gpurun_out/pmc_${tag}_$p -name '*counter_collection.csv' | head -1); done; } > $out/pmc_intersect.txt
echo "profile set $tag done"

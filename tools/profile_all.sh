#!/bin/bash
# usage: tools/profile_all.sh <tag> [config [spp]]
# The measurement set behind profiles/: the benchmark command under rocprofv3 --kernel-trace --stats, a kernel trace of
# one single-lane step (kernels one after the other: their own durations), then separate counter passes over one step of
# the same workload (instruction classes; wave time; L2 and fabric requests; written bytes; fetched bytes). Every step has
# its own time limit; the first failure ends the script.
# config: 3 (default) = BASELINE configs[2], one 64-spp step; 4 / 5 = configs[3] / configs[4] at `spp` frames (default 16: the
# 4K configurations take 16 frames per batch, so that is one whole batch per lane).
set -e
tag=${1:-run}
config=${2:-3}
if [ "$config" = 3 ]; then spp=${3:-64}; else spp=${3:-16}; fi
export KL_PMC_CONFIG=$config KL_PMC_SPP=$spp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
if [ "$config" = 3 ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- \
    python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-shared --skip-kernels > $out/bench_under_rocprof.json 2> $out/trace.log
else
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- \
    python3 bench.py --config $config --spp $spp --steps 2 --warmup 1 --cpu-seconds 0 --skip-shared --skip-kernels --skip-extras > $out/bench_under_rocprof.json 2> $out/trace.log
fi
python3 tools/rocpd_summary.py $(find $out/trace -name '*.db' | head -1) > $out/kernel_stats.md
echo "trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace1 -o trace1 -- python3 tools/pmc_workload.py $spp 1 $config > $out/single_lane_step.json 2> $out/trace1.log
python3 tools/rocpd_summary.py $(find $out/trace1 -name '*.db' | head -1) > $out/kernel_stats_single_lane.md
echo "single-lane trace done"
tools/pmc_pass.sh ${tag}_mix SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU
tools/pmc_pass.sh ${tag}_time SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
tools/pmc_pass.sh ${tag}_l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
tools/pmc_pass.sh ${tag}_write WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
tools/pmc_pass.sh ${tag}_fetch FETCH_SIZE
{ for p in mix time l2 write fetch; do echo "### pass $p: $(tail -1 gpurun_out/pmc_${tag}_$p.json)"; python3 tools/pmc_summary.py $(find gpurun_out/pmc_${tag}_$p -name '*counter_collection.csv' | head -1); done; } > $out/pmc_passes.txt
echo "profile set $tag done"

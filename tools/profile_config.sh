#!/bin/bash
# usage: tools/profile_config.sh <tag> <config> [spp]
# The counter set of tools/profile_all.sh without the bench trace: a kernel trace of one single-lane step of BASELINE
# configuration <config> (3, 4 or 5 as in bench.py --config) at `spp` frames, then the five counter passes over one
# library-default step of the same workload. Results under gpurun_out/prof_<tag>/.
set -e
tag=$1; config=$2; spp=${3:-16}
export KL_PMC_CONFIG=$config KL_PMC_SPP=$spp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace1 -o trace1 -- python3 tools/pmc_workload.py $spp 1 $config > $out/single_lane_step.json 2> $out/trace1.log
python3 tools/rocpd_summary.py $(find $out/trace1 -name '*.db' | head -1) > $out/kernel_stats_single_lane.md
echo "single-lane trace done"
tools/pmc_pass.sh ${tag}_mix SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU
tools/pmc_pass.sh ${tag}_time SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
tools/pmc_pass.sh ${tag}_l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
tools/pmc_pass.sh ${tag}_write WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
tools/pmc_pass.sh ${tag}_fetch FETCH_SIZE
{ for p in mix time l2 write fetch; do echo "### pass $p: $(tail -1 gpurun_out/pmc_${tag}_$p.json)"; python3 tools/pmc_summary.py $(find gpurun_out/pmc_${tag}_$p -name '*counter_collection.csv' | head -1); done; } > $out/pmc_passes.txt
echo "profile set $tag (config $config, $spp spp) done"

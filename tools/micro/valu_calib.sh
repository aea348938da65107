#!/bin/bash
# SQ counters over instruction streams of known content (tools/micro/valu_issue c2): what SQ_ACTIVE_INST_VALU tallies per
# instruction of each issue class, and which SQ_INSTS_VALU_* class each instruction falls into
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/valu_calib; mkdir -p gpurun_out/valu_calib
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/valu_calib -o a -- ./tools/micro/valu_issue c1 > gpurun_out/valu_calib/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d gpurun_out/valu_calib -o b -- ./tools/micro/valu_issue c2 > gpurun_out/valu_calib/b.log 2>&1
for f in a b; do echo "### pass $f"; python3 tools/pmc_summary.py gpurun_out/valu_calib/${f}_counter_collection.csv; done > gpurun_out/valu_calib/summary.txt

#!/bin/bash
# Collect SQ counters for the encode kernel on one corpus class (run ON the GPU box via gpurun).
# usage: tools/pmc_encode.sh <class> <tag> [n_blocks]
R=${GRAFT_REPO_ROOT:-/root/repo}
CLS=${1:-T_TEXT}; TAG=${2:-pmc}; NB=${3:-4096}
cd /tmp && export TMPDIR=/tmp
export PROBE_DBG=${PROBE_DBG:-0}
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $R/gpurun_out/${TAG}_a -- python3 $R/tools/encode_probe.py $NB $CLS > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES \
  --output-format csv -d $R/gpurun_out/${TAG}_b -- python3 $R/tools/encode_probe.py $NB $CLS > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_WAVES \
  --output-format csv -d $R/gpurun_out/${TAG}_c -- python3 $R/tools/encode_probe.py $NB $CLS > /dev/null 2>&1
python3 $R/tools/pmc_report.py $R/gpurun_out/${TAG}_a $R/gpurun_out/${TAG}_b $R/gpurun_out/${TAG}_c $NB

#!/bin/bash
# Vector-memory-side counters of the encode kernel on one corpus class (run ON the GPU box via gpurun): is the CU's
# address / cache pipeline what the rounds wait for?  usage: tools/pmc_encode_vmem.sh <class> <tag> [n_blocks]
R=${GRAFT_REPO_ROOT:-/root/repo}
CLS=${1:-T_TEXT}; TAG=${2:-pmcv}; NB=${3:-8192}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE \
  --output-format csv -d $R/gpurun_out/${TAG}_a -- python3 $R/tools/encode_probe.py $NB $CLS > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum \
  --output-format csv -d $R/gpurun_out/${TAG}_b -- python3 $R/tools/encode_probe.py $NB $CLS > /dev/null 2>&1
python3 $R/tools/pmc_report.py $R/gpurun_out/${TAG}_a $R/gpurun_out/${TAG}_b $NB

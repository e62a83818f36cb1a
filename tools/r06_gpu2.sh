#!/bin/bash
# round 6, GPU call 2: the advisor's fixes under test, the overlap probe, SQ counters per phase of the ring kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r06b
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_faults.py -x -q -m gpu -k "last_slice or splitter" 2>&1 | tail -5
timeout 600 python3 -m pytest tests/test_gpu_framed_device.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python3 tools/probes/overlap_probe.py 32768 MIX > $O/overlap_mix.txt 2>&1; grep -v amdgpu.ids $O/overlap_mix.txt | tail -12
timeout 600 python3 tools/probes/overlap_probe.py 16384 T_TEXT > $O/overlap_text.txt 2>&1; grep -v amdgpu.ids $O/overlap_text.txt | tail -12
# SQ counters of the ring kernel by phase (debug build: SNAPPY_HIP_DBG 2 = no resolvers, 4 = no front end, 6 = neither)
export SNAPPY_HIP_LIBRARY=$PWD/tools/probes/lib_dbg.so
for dbg in 0 2 4 6; do
  export PROBE_DBG=$dbg
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM \
    --output-format csv -d $O/pmc_dbg$dbg -- python3 tools/decode_probe.py 8192 T_TEXT > $O/pmc_dbg$dbg.log 2>&1
  echo "== dbg $dbg"; python3 tools/pmc_report.py $O/pmc_dbg$dbg 8192 | grep decode_indexed | tee $O/pmc_dbg$dbg.txt
  rm -rf $O/pmc_dbg$dbg
done

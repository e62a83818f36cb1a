#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
export PROBE_DBG=0 SNAPPY_HIP_STATS=1
for v in dbg pkdbg; do
  echo "== $v"
  SNAPPY_HIP_LIBRARY=$PWD/tools/probes/lib_$v.so timeout 300 python3 tools/decode_probe.py 8192 T_TEXT 2>&1 | grep -v amdgpu.ids | grep "STATS\|T_TEXT" | tail -6
done

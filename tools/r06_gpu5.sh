#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r06e
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_faults.py -x -q -m gpu -k "not bench and not full_size" 2>&1 | tail -3
export PROBE_DBG=0
bash tools/ab.sh "python3 tools/decode_probe.py 16384 T_TEXT,T_HTML" base trim1 trim2 trim3 2>&1 | grep -v "^nim-snappy" | tee $O/ab_trim3.txt

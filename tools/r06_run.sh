#!/bin/bash
# round 6: the GPU test-suite, same-box A/B of the encoder (round 5's library against this one), bench + rocprof + PMC traffic
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r06g
mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_gpu.txt
bash tools/ab.sh "python3 tools/encode_probe.py 16384 T_TEXT,T_HTML" base cur 2>&1 | grep -v "^nim-snappy" | tee $O/ab_encode.txt
bash tools/profile_bench.sh r06a 2>&1 | tail -60

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r06c
mkdir -p $O
timeout 600 python3 tools/probes/overlap_probe.py 32768 MIX > $O/overlap_mix.txt 2>&1; grep -v amdgpu.ids $O/overlap_mix.txt | tail -14
timeout 600 python3 tools/probes/overlap_probe.py 16384 T_TEXT > $O/overlap_text.txt 2>&1; grep -v amdgpu.ids $O/overlap_text.txt | tail -14

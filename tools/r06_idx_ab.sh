mkdir -p gpurun_out/r06c
PROBE_DBG=0 bash tools/ab.sh "timeout 300 python tools/decode_probe.py 16384 T_TEXT,T_HTML,MIX | grep dbg" "$@" > gpurun_out/r06c/ab_idx_$2.txt 2>&1
cat gpurun_out/r06c/ab_idx_$2.txt | cut -c1-60

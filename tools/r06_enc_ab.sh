mkdir -p gpurun_out/r06c
bash tools/ab.sh "timeout 300 python tools/encode_probe.py 8192 T_TEXT,T_HTML,MIX | grep encode" "$@" > gpurun_out/r06c/ab_enc_$1_$2.txt 2>&1
cat gpurun_out/r06c/ab_enc_$1_$2.txt
for v in "$@"; do
SNAPPY_HIP_LIBRARY=$PWD/tools/probes/lib_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -k "enc or compress or round" -x -q 2>&1 | tail -2
done

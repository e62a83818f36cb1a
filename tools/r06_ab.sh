# usage: tools/r06_ab.sh <tag> <classes> <variants...>   (decode_probe A/B + parity tests of the last variant)
mkdir -p gpurun_out/r06c
tag=$1; cls=$2; shift 2
PROBE_DBG=0 bash tools/ab.sh "timeout 300 python tools/decode_probe.py 16384 $cls | grep dbg" "$@" > gpurun_out/r06c/ab_$tag.txt 2>&1
cut -c1-120 gpurun_out/r06c/ab_$tag.txt
last=${@: -1}
SNAPPY_HIP_LIBRARY=$PWD/tools/probes/lib_$last.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_faults.py -x -q 2>&1 | tail -2

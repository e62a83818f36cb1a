#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r06h
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_framed_device.py tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_faults.py -x -q -m gpu -k "not bench" 2>&1 | tail -4
bash tools/ab.sh "python3 tools/framed_probe.py 65536 5" cur crc1 2>&1 | grep -v "^nim-snappy" | tee $O/ab_crc1.txt
export SNAPPY_HIP_LIBRARY=$PWD/tools/probes/lib_crc1.so
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp -- python3 tools/framed_probe.py 65536 5 > $O/fp.log 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/r06h/fp/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-70s calls %5s avg ms %8.4f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e6))
PY
rm -rf $O/fp

#!/bin/bash
# L2 <-> fabric traffic of ONE full-size encode launch by the number of second waves (run ON the GPU box via gpurun):
# FETCH_SIZE / WRITE_SIZE in separate passes (MI355X_MICROARCH.md), KB per launch; FETCH_SIZE x 2 on gfx950.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for g in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmct
    SNAPPY_HIP_ENC_GWAVES=$g timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmct -- python3 $R/tools/encode_full_probe.py > /dev/null 2>&1
    python3 - "$g" "$c" <<'PY'
import csv, glob, sys
g, c = sys.argv[1], sys.argv[2]
tot, n = 0.0, set()
for f in glob.glob("/tmp/pmct/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "encode_blocks" in row["Kernel_Name"] and row["Counter_Name"] == c:
            tot += float(row["Counter_Value"]); n.add(row["Dispatch_Id"])
per = tot / max(1, len(n)) * 1024 * (2 if c == "FETCH_SIZE" else 1)
print("second waves per four workgroups %s: %s -> %.1f GB per launch (%d launches)" % (g, c, per / 1e9, len(n)))
PY
  done
done

"""Kernel timeline of the last snappy_hip_uncompress_d call over one raw multi-block buffer in a
`rocprofv3 --kernel-trace --output-format csv` run of tools/probes/raw64.py.  usage: tools/raw_timeline.py <dir>   Not a test."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
i0 = [i for i, n in enumerate(names) if "split_init_kernel" in n][-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %8.1f us (gap %6.1f)  %s  grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3,
          r["Kernel_Name"].replace("snappy_hip::", "")[:70], r.get("Grid_Size", "")))
    prev_end = e

"""Kernel timeline of the last uncompress_framed call of a `rocprofv3 --kernel-trace --output-format csv` run of
tools/framed_probe.py.  usage: tools/framed_timeline.py <dir>   Not a test."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
di = [i for i, n in enumerate(names) if "decode_indexed_kernel" in n and "16384" in n]
i0 = di[-1]
lo, hi = max(0, i0 - 24), min(len(rows), i0 + 10)
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    print("%8.1f %8.1f us  %s  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:80], r.get("Grid_Size", "")))

"""Aggregates rocprofv3 PC-sampling output (csv) per kernel and per instruction / source line.
usage: tools/pcs_report.py <dir with *pc_sampling*.csv and *kernel_trace.csv> <kernel name substring> [out.txt]
Columns are found by name (the format is beta): the decoded instruction, its comment (file:line with
-gline-tables-only), whether the sampled wave issued, the stall reason, the instruction type."""
import csv, glob, os, sys, collections
d, kname = sys.argv[1], sys.argv[2]
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
csv.field_size_limit(1 << 30)
def find(pattern):
    return sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
disp = {}
for f in find("*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r.get("Dispatch_Id") or r.get("Dispatch_ID")
        if k is not None:
            disp[k] = r.get("Kernel_Name", "")
files = find("*pc_sampling*.csv")
print("files:", files, "dispatches:", len(disp), file=out)
for f in files:
    rd = csv.DictReader(open(f))
    cols = rd.fieldnames
    print("columns of %s: %s" % (os.path.basename(f), cols), file=out)
    def col(*subs):
        for c in cols:
            if all(s.lower() in c.lower() for s in subs):
                return c
        return None
    c_ins, c_cmt, c_disp = col("instruction"), col("comment"), col("dispatch")
    c_iss, c_stall, c_type, c_wc = col("issued"), col("stall"), col("type"), col("wave", "count")
    n = 0
    by_line = collections.defaultdict(collections.Counter)
    by_ins = collections.defaultdict(collections.Counter)
    stalls = collections.Counter()
    types = collections.Counter()
    kernels = collections.Counter()
    for r in rd:
        kn = disp.get(r.get(c_disp, ""), "?") if c_disp else "?"
        kernels[kn[:60]] += 1
        if kname not in kn:
            continue
        n += 1
        iss = (r.get(c_iss) or "").strip() if c_iss else ""
        st = (r.get(c_stall) or "").strip() if c_stall else ""
        key = "issued" if iss in ("1", "true", "True") else ("stall:" + st if st else "not_issued")
        line = (r.get(c_cmt) or "").strip() if c_cmt else ""
        line = line.split("/")[-1]
        by_line[line][key] += 1
        by_ins[(line, (r.get(c_ins) or "").strip())][key] += 1
        stalls[key] += 1
        if c_type:
            types[(r.get(c_type) or "").strip() + ("/issued" if key == "issued" else "")] += 1
    print("samples per kernel:", kernels.most_common(12), file=out)
    print("== %s: %d samples" % (kname, n), file=out)
    print("by state:", stalls.most_common(), file=out)
    print("by instruction type:", types.most_common(), file=out)
    print("-- by source line (samples, share, issued, top stall reasons)", file=out)
    for line, c in sorted(by_line.items(), key=lambda kv: -sum(kv[1].values()))[:150]:
        t = sum(c.values())
        print("%7d %5.2f%% issued %5d  %-38s %s" % (t, 100.0 * t / max(n, 1), c["issued"], line,
              " ".join("%s=%d" % kv for kv in c.most_common(4) if kv[0] != "issued")), file=out)
    print("-- by instruction (top 200)", file=out)
    for (line, ins), c in sorted(by_ins.items(), key=lambda kv: -sum(kv[1].values()))[:200]:
        t = sum(c.values())
        print("%7d %5.2f%% issued %5d  %-34s %-60s %s" % (t, 100.0 * t / max(n, 1), c["issued"], line, ins[:60],
              " ".join("%s=%d" % kv for kv in c.most_common(3) if kv[0] != "issued")), file=out)

"""Summarise rocprofv3 counter_collection.csv files per kernel (sum over XCDs, mean over dispatches)."""
import collections, csv, glob, sys
dirs, nb = sys.argv[1:-1], int(sys.argv[-1])
for d in dirs:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "decode_indexed" not in k and "index_units" not in k:
                continue
            k = "decode_indexed" if "decode_indexed" in k else "index_units"
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
        for k, v in acc.items():
            for c, val in sorted(v.items()):
                n = max(1, len(disp[(k, c)]))
                print("%-16s %-26s per dispatch %.4e  per block %10.1f" % (k, c, val / n, val / n / nb))

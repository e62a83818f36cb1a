"""Summarise rocprofv3 counter_collection.csv files per kernel (sum over XCDs, mean over dispatches)."""
import collections, csv, glob, sys
dirs, nb = sys.argv[1:-1], int(sys.argv[-1])
for d in dirs:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            names = ("decode_indexed", "index_units", "encode_blocks")
            if not any(x in k for x in names):
                continue
            k = [x for x in names if x in k][0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
        for k, v in acc.items():
            for c, val in sorted(v.items()):
                n = max(1, len(disp[(k, c)]))
                print("%-16s %-26s per dispatch %.4e  per block %10.1f" % (k, c, val / n, val / n / nb))

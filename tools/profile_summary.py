"""Condense the rocprofv3 outputs of tools/profile_bench.sh into one markdown summary."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_id
out = sys.argv[1]
def short(k):
    # the indexed decoder's two instantiations: ring window first, whole-block window for what it passes on
    for win in ("16384", "32768"):
        if "decode_indexed_kernel<%su, true" % win in k or "decode_indexed_kernelILj%sELb1" % win in k:
            return "decode_indexed_kernel<%s,crc>" % win  # (the ring instantiation that checksums the rows it flushes)
    for win in ("16384", "32768", "65536"):
        if "decode_indexed_kernel<%s" % win in k or "decode_indexed_kernelILj%s" % win in k:
            return "decode_indexed_kernel<%s>" % win
    for name in ("decode_indexed_kernel", "index_units_kernel", "decode_units_kernel", "encode_blocks_kernel",
                 "crc32c_units_kernel", "gather_slots_kernel", "scan_sizes_kernel", "region_counts_kernel",
                 "frame_chase_kernel", "frame_stitch_kernel", "frame_fill_kernel", "frame_scatter_kernel",
                 "frame_scan_kernel", "frame_verdict_kernel", "copy_units_kernel", "split_walk_kernel", "decode_sparse_kernel",
                 "status_list_kernel", "passed_on_list_kernel",
                 "split_check_kernel", "encode_sketch_kernel", "order_count_kernel", "order_scan_kernel",
                 "order_scatter_kernel"):
        if name in k:
            return name
    return k[:60]
print("# rocprofv3 summary of `python3 bench.py` (1 x MI355X)\n")
for f in ("bench.json", "bench_profiled.json"):
    p = os.path.join(out, f)
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                j = json.loads(line)
                print("* `%s`: value %.1f %s, ms_per_step %.3f, roofline %s, compress %.2f GB/s" % (
                    f, j["value"], j["unit"], j["ms_per_step"], json.dumps(j["roofline"]), j.get("compress_GBps", 0)))
print("\n## kernel-trace --stats (bench.py --steps 5 --warmup 1 --no-cpu --shard-gib 0)\n")
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    print("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|")
    for row in csv.DictReader(open(f)):
        print("| %s | %s | %.3f | %.4f | %s |" % (short(row["Name"]), row["Calls"], float(row["TotalDurationNs"]) / 1e6,
                                                 float(row["AverageNs"]) / 1e6, row["Percentage"]))
# the library's kernels by grid size: the bench launches some of them on batches of several sizes (the staged sharded
# compress encodes 16 384 blocks a launch, the headline legs 65 536), and an average over all of them says nothing
print("\n## the library's kernels by launch size (same pass: kernel_trace.csv)\n")
ours = ("decode_indexed_kernel", "index_units_kernel", "decode_units_kernel", "encode_blocks_kernel", "crc32c_units_kernel",
        "gather_slots_kernel", "decode_sparse_kernel")
by = collections.defaultdict(list)
for f in glob.glob(out + "/stats/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if any(k.startswith(o) or o in k for o in ours):
            by[(k, int(row["Grid_Size_X"]) // max(1, int(row["Workgroup_Size_X"])))].append(
                (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
print("| kernel | workgroups | calls | avg ms | min ms |\n|---|---|---|---|---|")
for (k, g), v in sorted(by.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    print("| %s | %d | %d | %.4f | %.4f |" % (k, g, len(v), sum(v) / len(v), min(v)))
print("\n## HBM traffic per launch (separate --pmc passes; FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md; the launches of the LARGEST grid of each kernel)\n")
tr = collections.defaultdict(dict)
for name, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for f in glob.glob(out + "/" + d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(float)
        disp = collections.defaultdict(set)
        big = collections.defaultdict(int)
        rows = [row for row in csv.DictReader(open(f)) if row["Counter_Name"] == name]
        for row in rows:
            k = short(row["Kernel_Name"])
            big[k] = max(big[k], int(row["Grid_Size"]))
        for row in rows:
            k = short(row["Kernel_Name"])
            if int(row["Grid_Size"]) != big[k]:
                continue  # (a smaller batch of the same kernel)
            acc[k] += float(row["Counter_Value"])
            disp[k].add(row["Dispatch_Id"])
        for k in acc:
            tr[k][name] = acc[k] / max(1, len(disp[k])) * 1024.0  # KB -> bytes, mean per dispatch
print("| kernel | read bytes (2 x FETCH_SIZE) | write bytes | total |\n|---|---|---|---|")
for k, v in sorted(tr.items()):
    rd = 2.0 * v.get("FETCH_SIZE", 0.0)
    wr = v.get("WRITE_SIZE", 0.0)
    print("| %s | %.4e | %.4e | %.4e |" % (k, rd, wr, rd + wr))

# machine-readable copy (bench.py reports the newest profiles/*_traffic.json as roofline.traffic)
kern = {k: {"read_bytes": 2.0 * v.get("FETCH_SIZE", 0.0), "write_bytes": v.get("WRITE_SIZE", 0.0),
            "total_bytes": 2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)}
        for k, v in tr.items() if "kernel" in k and not k.startswith("void")}
with open(os.path.join(out, "traffic.json"), "w") as fh:
    json.dump({"csrc_sha256": build_id.csrc_sha256(),  # the kernel sources this was measured on (bench.py checks it)
               "workload": "bench.py --steps 2 --warmup 1 --no-cpu --shard-gib 0 (65536 x 64 KiB blocks, class mix default, seed 0x5EED5AA9), 1 x MI355X",
               "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = KB x 1024, "
                         "FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md HBM section); mean per dispatch.  FETCH_SIZE counts "
                         "the L2's requests to the fabric, Infinity Cache hits included",
               "kernels": kern}, fh, indent=1)

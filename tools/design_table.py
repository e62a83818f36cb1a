"""DESIGN.md 8's table and per-class paragraph from a committed bench line: tools/design_table.py <tag>  (profiles/<tag>_bench_full.json,
profiles/<tag>_traffic.json).  Not a test; the numbers in DESIGN.md are pasted from its output."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
d = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench_full.json")))
t = json.load(open(os.path.join(ROOT, "profiles", tag + "_traffic.json")))
r, c, h, pc = d["roofline"], d["cpu_baseline"], d["host_api"], d["per_class"]
a29 = d["config1_alice29"]
def n(x, k=0):
    s = ("%%.%df" % k) % x
    i, _, f = s.partition(".")
    if len(i) > 3: i = i[:-3] + " " + i[-3:]
    return i + ("." + f if f else "")
gb = lambda k, w: t["kernels"][k][w + "_bytes"] / 1e9
print("`profiles/%s_bench_full.json` (the whole line), `%s_rocprof_summary.md` (kernel-trace stats + PMC traffic of the same\nsources, `csrc_sha256` %s…), same box:\n" % (tag, tag, t["csrc_sha256"][:8]))
print("| configs | what | GB/s of uncompressed bytes | (stream + U) / t of the 8 TB/s peak |\n|---|---|---|---|")
print("| [1] | block decompress, the survey's mix (`value`) | **%s** (%.2f ms a step: index pass %.2f + ring kernel %.2f + passed-on units %.2f) | %.3f (the ring kernel on its own units: %.3f) |"
      % (n(d["value"], 1), d["ms_per_step"], r["index_pass_kernel_ms"], r["kernel_ms"], r["passed_on_units_kernel_ms"], r["decompress_step"]["frac"], r["frac"]))
print("| [1] read literally | random bytes (`class_R`, 16 384 blocks) | %s | %.2f |" % (n(r["class_R"]["value"]), r["class_R"]["frac"]))
print("| [2] | block compress | **%s** (`encode_blocks_kernel` %.1f ms) | %.3f |" % (n(r["compress"]["value"], 1), r["compress"]["kernel_ms"], r["compress"]["frac"]))
print("| [1] + [2] | round trip | %s | %.3f |" % (n(r["round_trip"]["value"], 1), r["round_trip"]["frac"]))
print("| [3] | one framed stream of 65 536 chunks: compress / decompress with CRC verification (mean of %d calls) | %s / **%s** (%.3f of the raw units' rate) | %.3f / %.3f |"
      % (r["framed"]["calls"], n(r["framed_compress"]["value"], 1), n(r["framed"]["value"], 1), r["framed"]["over_value"], r["framed_compress"]["frac"], r["framed"]["frac"]))
print("| [4], N = 1 | 32 GiB block-range sharded compress, host concatenate | %s | link-bound |" % n(d["sharded_compress"]["strong_GBps"], 1))
print("| | one raw multi-block buffer resident in HBM: 1 GiB / 64 MiB | %s / %s | %.3f / %.3f |"
      % (n(r["raw_buffer_1GiB"]["value"]), n(r["raw_buffer_64MiB"]["value"]), r["raw_buffer_1GiB"]["frac"], r["raw_buffer_64MiB"]["frac"]))
print("| | host-buffer C ABI, 1 GiB (PCIe included): compress / uncompress, framed / raw | %.1f / %.1f, %.1f / %.1f | |"
      % (h["compress_framed_GBps"], h["uncompress_framed_GBps"], h["compress_GBps"], h["uncompress_GBps"]))
print("| [0] | `alice29.txt` through the C ABI, ms per call (encode / decode): HIP %.2f / %.2f, oracle on one core %.2f / %.2f, the reference's README 0.33 / 0.19 | | |"
      % (a29["hip_host_api_raw"][0], a29["hip_host_api_raw"][1], a29["oracle_inMemory_raw"][0], a29["oracle_inMemory_raw"][1]))
cm = c["compress_threads_value_min_max"]
print("| | CPU oracle, same box: one thread %.2f (decompress) / %.2f (compress); all %d host CPUs %.1f / %.1f–%.1f | | GPU / all CPUs: %.1f× / %.1f× |"
      % (c["value"], c["compress_value"], c["host_cpus"], c["threads_value"], cm[0], cm[1], d["value"] / c["threads_value"], r["compress"]["value"] / cm[1]))
names = [("T_TEXT", "text"), ("T_HTML", "html"), ("RS", "repeated strings"), ("R", "random bytes"), ("P10", "period 10"), ("Z", "zeros"), ("RAMP", "ramp")]
print("\nPer class alone (16 384 blocks; decompress raw / framed, compress, GB/s): " + "; ".join(
    "%s %s / %s, %s" % (nm, n(pc[k]["decompress_GBps"]), n(pc[k]["framed_decompress_GBps"]), n(pc[k]["compress_GBps"], 1 if pc[k]["compress_GBps"] < 100 else 0)) for k, nm in names) + ".")
ring, idx, po, enc = "decode_indexed_kernel<16384>", "index_units_kernel", "decode_indexed_kernel<65536>", "encode_blocks_kernel"
step = sum(gb(k, w) for k in (ring, idx, po) for w in ("read", "write"))
print("PMC per launch (`%s_traffic.json`): ring kernel %.2f GB read + %.2f GB written (its units' C + U = %.2f GB: %.2f×), index pass %.2f + %.2f GB, passed-on units %.2f + %.2f GB — the step %.1f GB against %.2f GB algorithmic (%.2f×); `encode_blocks_kernel` %.0f + %.0f GB."
      % (tag, gb(ring, "read"), gb(ring, "write"), r["algorithmic_bytes_per_launch"] / 1e9, (gb(ring, "read") + gb(ring, "write")) / (r["algorithmic_bytes_per_launch"] / 1e9),
         gb(idx, "read"), gb(idx, "write"), gb(po, "read"), gb(po, "write"), step, r["algorithmic_bytes_per_step"] / 1e9, step / (r["algorithmic_bytes_per_step"] / 1e9), gb(enc, "read"), gb(enc, "write")))

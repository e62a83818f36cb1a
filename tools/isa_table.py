"""Static instruction table of a kernel by SOURCE REGION (no GPU needed).
usage: tools/isa_table.py [--md]
Compiles csrc/snappy_hip.hip to assembly with -gline-tables-only, attributes every instruction of
decode_indexed_kernel<16384> to the region of decode2_kernel.h whose source lines it carries (an inlined helper's
instructions -- common.h's scans, HIP's atomics -- go to the region of the nearest decode2_kernel.h instruction in
front of them), and counts vector / scalar / LDS / memory instructions, branches, waits (s_waitcnt, s_nop) and the
v_readlane / v_writelane traffic of spilled scalar registers per region.  A straight count over the text: both sides of
a branch are counted; the dynamic figures beside it in profiles/r06_instruction_table.md come from SQ counters."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "nim-snappy_amd", "csrc", "snappy_hip.hip")
out = os.path.join(tempfile.gettempdir(), "snappy_isa_table.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "--cuda-device-only",
                "-S", "-gline-tables-only", "-o", out, SRC], check=True, stderr=subprocess.DEVNULL)
text = open(os.path.join(ROOT, "nim-snappy_amd", "csrc", "decode2_kernel.h")).read().split("\n")


def line_of(marker, nth=1):
    hits = [i + 1 for i, l in enumerate(text) if marker in l]
    return hits[nth - 1]


# region boundaries by markers in the source (so the table follows the file as it changes)
marks = [
    ("start of the workgroup: parameters, first bytes, one-literal / one-period units", "const uint32_t tid = threadIdx.x;"),
    ("step loop: top, ring flush, prefetch hand-over", "for (uint32_t s = 0; s <= n_chunks; s++) {"),
    ("front end: start counts (scan over both halves)", "// =================================== front end"),
    ("front end, per trip: index entry, my bytes", "for (uint32_t trip = 0; trip < kFeTrips; trip++) {"),
    ("front end, per trip: the two elements (tag table, lengths, offsets)", "// ---- my (up to) two elements"),
    ("front end, per trip: output positions, list slots", "// ---- output positions: the region's first one"),
    ("front end, per trip: list entries, boundary slots", "// ---- list entries; the slot at each 256-byte"),
    ("front end, per trip: literal payload stores", "// ---- literal payloads ----"),
    ("front end: literals with length bytes (whole wave, from HBM)", "// long literals: whole wave, straight from HBM"),
    ("resolvers: per step set-up, per group top (skip looks, E0)", "// =================================== resolvers"),
    ("resolvers, per group: zero the scratch, scatter the elements' values", "const uint32_t E0 = g > cb"),
    ("resolvers, per group: keys, prefix maximum, per-byte offset / source", "uint32_t kj[B];"),
    ("resolvers, per group: run detection (rare)", "uint32_t run_off = 0, run_end = 0;"),
    ("resolvers, per group: pointer-doubling rounds", "bool dep[B];"),
    ("resolvers, per group: read-backs from HBM (ring), run / skip checks", "uint32_t far_m = 0, far_v = 0;"),
    ("resolvers, per group: turn addresses", "// ---- everything the turn needs is worked out before the wait"),
    ("resolvers, per group: the wait for the turn (looks)", "const uint32_t expect = g > cb ? g : cb;"),
    ("resolvers, per group: gather, store, run extension, publish", "if (front > expect) {  // covered by a run extension meanwhile"),
    ("step loop: next step's prefetch, barrier", "const unsigned long long tm1 = SNAPPY_STATS(prm)"),
    ("end of the unit: flush, window CRC", "// ---- flush ----"),
]
bounds = [(name, line_of(m)) for name, m in marks]
bounds.sort(key=lambda t: t[1])


def region_of(line):
    r = None
    for name, lo in bounds:
        if line >= lo:
            r = name
    return r or "(prologue)"


lines = open(out).read().split("\n")
kname = "decode_indexed_kernelILj16384"
k0 = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and kname in l)
k1 = next(i for i in range(k0 + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
cur_region = "(prologue)"
per = collections.OrderedDict((name, collections.Counter()) for name, _ in bounds)
per["(prologue)"] = collections.Counter()
for l in lines[k0:k1]:
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        if files.get(int(m.group(1))) == "decode2_kernel.h" and int(m.group(2)) > 0:
            cur_region = region_of(int(m.group(2)))
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = ("spill" if op in ("v_readlane_b32", "v_writelane_b32") else "valu" if op.startswith("v_") else
            "wait" if op in ("s_waitcnt", "s_nop") else "branch" if op.startswith("s_cbranch") or op == "s_branch" else
            "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem")
    per[cur_region][kind] += 1
cols = ["valu", "salu", "branch", "wait", "lds", "vmem", "spill"]
print("| region of `decode2_kernel.h` (from line) | " + " | ".join(cols) + " |")
print("|---|" + "---|" * len(cols))
tot = collections.Counter()
for name, lo in bounds:
    c = per[name]
    tot.update(c)
    print("| %s (%d) | %s |" % (name, lo, " | ".join(str(c[k]) for k in cols)))
print("| **whole kernel** | %s |" % " | ".join(str(tot[k] + per["(prologue)"][k]) for k in cols))

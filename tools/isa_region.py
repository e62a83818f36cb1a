"""Instruction mix of a region of a kernel's ISA (no GPU needed).
usage: tools/isa_region.py <kernel name substring> <start regex> <end regex> [nth start] [--dump] [hipcc flags...]
Compiles csrc/snappy_hip.hip to assembly, takes the lines of the first kernel whose name contains the
substring from the nth line matching <start> to the next line matching <end>, and counts vector / scalar /
LDS / memory instructions (straight count over the text: both sides of a branch are counted)."""
import os, re, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
dump = "--dump" in args
args = [a for a in args if a != "--dump"]
name, start, end = args[0], args[1], args[2]
nth = int(args[3]) if len(args) > 3 and args[3].isdigit() else 1
flags = [a for a in args[3:] if not a.isdigit()]
out = os.path.join(tempfile.gettempdir(), "snappy_isa_%s.s" % "_".join(f.strip("-") for f in flags))
src = os.path.join(ROOT, "nim-snappy_amd", "csrc", "snappy_hip.hip")
deps = [os.path.join(ROOT, "nim-snappy_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "nim-snappy_amd", "csrc"))]
if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                    "--cuda-device-only", "-S", "-o", out, src] + flags, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
k0 = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and name in l)
k1 = next(i for i in range(k0 + 1, len(lines)) if lines[i].startswith(".Lfunc_end") or re.match(r"^_Z\w*:", lines[i]))
body = lines[k0:k1]
hits = [i for i, l in enumerate(body) if re.search(start, l)]
a = hits[nth - 1]
b = next(i for i in range(a + 1, len(body)) if re.search(end, body[i]))
cnt = collections.Counter()
for l in body[a:b + 1]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":") or re.match(r"^\.?LBB", t) or re.match(r"^\d+:$", t):
        continue
    op = t.split()[0]
    kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
            "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    cnt[kind] += 1
    cnt["v_readlane/writelane"] += op in ("v_readlane_b32", "v_writelane_b32")
    cnt["s_nop"] += op == "s_nop"
    if dump:
        print(l[:120])
print("lines %d..%d of %s:" % (a, b, name), dict(cnt))

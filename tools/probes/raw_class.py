"""Probe: snappy_hip_uncompress_d of one raw buffer of corpus blocks of one class: status, which blocks differ.
usage: raw_class.py CLASS [blocks]   (debug library: SNAPPY_HIP_STATS=1 prints the split's rounds)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
only = None if sys.argv[1] == "mix" else sys.argv[1]
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
src = corpus.make_blocks(0, nb, only=only).tobytes()
raw = hip.encode(src)
ctx = hip.Context(0)
d_in = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
d_out = torch.empty(len(src), dtype=torch.uint8, device="cuda")
st, w = ctx.uncompress(d_in, len(raw), d_out, len(src))
got = d_out.cpu().numpy().tobytes()
print("status", st, "written", w, "of", len(src), "equal", got == src)
import time
ts = []
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.uncompress(d_in, len(raw), d_out, len(src))
    ts.append(time.perf_counter() - t0)
print("%s x %d: %.3f ms = %.1f GB/s of output (stream %d bytes)" % (sys.argv[1], nb, min(ts) * 1e3, len(src) / min(ts) / 1e9, len(raw)))
if got != src:
    bad = [i // 65536 for i in range(0, len(src), 65536) if got[i:i + 65536] != src[i:i + 65536]]
    print("blocks that differ:", bad[:20], len(bad), "of", (len(src) + 65535) // 65536)

import importlib, os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
ctx = hip.Context(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
src = corpus.make_blocks(0, nb).tobytes()
raw = hip.encode(src)
d_in = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
d_out = torch.empty(len(src), dtype=torch.uint8, device="cuda")
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st, w = ctx.uncompress(d_in, len(raw), d_out, len(src))
    print("ms %.3f" % ((time.perf_counter() - t0) * 1e3), flush=True)
assert (st, w) == (0, len(src))

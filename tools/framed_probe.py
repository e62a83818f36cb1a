"""Times uncompressFramed / compressFramed of ONE framed stream resident in HBM (BASELINE configs[3]); run it
under `rocprofv3 --kernel-trace --stats` to see where the framed path's time goes.  Not a test.
    python tools/framed_probe.py [n_blocks] [reps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
ctx = hip.Context(0)
d_in = torch.empty(nb * 65536, dtype=torch.uint8, device=dev)
for b0 in range(0, nb, 4096):
    c = min(4096, nb - b0)
    d_in[b0 * 65536:(b0 + c) * 65536] = corpus.make_blocks_torch(torch, b0, c, dev).reshape(-1)
cap = hip.max_compressed_len_framed(nb * 65536)
d_fs = torch.empty(cap, dtype=torch.uint8, device=dev)
flen = ctx.compress_framed(d_in, nb * 65536, d_fs, cap)
d_out = torch.empty(nb * 65536, dtype=torch.uint8, device=dev)
assert ctx.uncompress_framed(d_fs, flen, d_out, nb * 65536) == (0, flen, nb * 65536)
for name, f in (("uncompress_framed", lambda: ctx.uncompress_framed(d_fs, flen, d_out, nb * 65536)),
                ("compress_framed", lambda: ctx.compress_framed(d_in, nb * 65536, d_fs, cap))):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    print(name, "ms", ["%.3f" % (t * 1e3) for t in ts], "GB/s %.1f" % (nb * 65536 / min(ts) / 1e9), flush=True)
assert bool(torch.equal(d_out, d_in))

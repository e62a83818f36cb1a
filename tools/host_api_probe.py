"""End-to-end (host buffers, PCIe included) rates of the C-ABI host calls. Not a test."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import torch  # noqa: F401  (one HIP runtime)
hip = importlib.import_module("nim-snappy_amd")
import corpus
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
src = corpus.make_blocks(0, nb).reshape(-1).tobytes()
for name, enc, dec in (("framed", hip.encode_framed, hip.decode_framed),):
    enc(src[:1 << 20])  # warm
    t0 = time.perf_counter(); comp = enc(src); t1 = time.perf_counter()
    out = dec(comp); t2 = time.perf_counter()
    assert out == src
    print("%s host API, %d MiB: compress %.2f GB/s, uncompress %.2f GB/s (ratio %.3f)" % (
        name, len(src) >> 20, len(src) / (t1 - t0) / 1e9, len(src) / (t2 - t1) / 1e9, len(comp) / len(src)), flush=True)
# raw (unframed) multi-block buffer: no block delimiters in the stream
nbr = min(nb, 512)
src2 = src[:nbr * 65536]
comp = hip.encode(src2)
t0 = time.perf_counter(); out = hip.decode(comp); t1 = time.perf_counter()
assert out == src2
print("raw host API, %d MiB: uncompress %.3f GB/s" % (len(src2) >> 20, len(src2) / (t1 - t0) / 1e9), flush=True)

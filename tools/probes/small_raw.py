"""Probe: host-buffer uncompress / decode of small raw multi-block buffers (ms per call)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")]
hip = importlib.import_module("nim-snappy_amd")
import pyoracle as orc
text = open(os.path.join(ROOT, "tests", "golden", "data", "alice29.txt"), "rb").read() * 8
for n in (70000, 152089, 400000, 800000, 1216712):
    src = text[:n]
    comp = orc.encode(src)
    assert hip.decode(comp) == src
    t0 = time.perf_counter()
    for _ in range(50):
        hip.decode(comp)
    t = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(20):
        orc.decode(comp)
    to = (time.perf_counter() - t0) / 20
    print("%8d bytes: hip %.3f ms  oracle %.3f ms" % (n, t * 1e3, to * 1e3), flush=True)

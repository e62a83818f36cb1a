"""Latency of the host-buffer calls on the reference's small files (README.md:97-125), p50 over many calls, through the C ABI
on preallocated buffers (bench.py's AbiCaller).  Not a test.   python tools/small_call_probe.py [calls]"""
import importlib, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")]
import torch  # noqa
hip = importlib.import_module("nim-snappy_amd")
import pyoracle as orc
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
def p50(f):
    f(); ts = []
    for _ in range(calls):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3
for name in ("Mark.Twain-Tom.Sawyer.txt", "html", "alice29.txt", "fireworks.jpeg", "urls.10K", "kppkn.gtb"):
    src = open(os.path.join(ROOT, "tests", "golden", "data", name), "rb").read()
    h, o = bench._callers(hip, orc, src)
    print("%-28s %7d B  raw enc %.3f dec %.3f | framed enc %.3f dec %.3f ms   (oracle raw %.3f / %.3f)" % (
        name, len(src), p50(h.encode), p50(h.decode), p50(h.encode_framed), p50(h.decode_framed), p50(o.encode), p50(o.decode)), flush=True)

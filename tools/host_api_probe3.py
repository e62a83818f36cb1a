"""Host-buffer API end to end (PCIe copies included): GB/s of uncompressed bytes, and two host
threads at once.  Calls the C ABI with preallocated numpy buffers (no Python-side copies).
Not a test.  usage: host_api_probe3.py [blocks] [--json]"""
import ctypes, importlib, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
hip = importlib.import_module("nim-snappy_amd")
import corpus
lib = hip.lib
nb = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16384   # 1 GiB
src = np.ascontiguousarray(corpus.make_blocks(0, nb).reshape(-1))
n = src.size
PINNED = "--pinned" in sys.argv  # the caller's buffers are page-locked (hipHostMalloc via torch)
if PINNED:
    import torch
    _keep = []
    def empty(k):
        t = torch.empty(k, dtype=torch.uint8).pin_memory(); _keep.append(t)
        return t.numpy()
    p = empty(n); p[:] = src; src = p
else:
    empty = lambda k: np.empty(k, dtype=np.uint8)
P = lambda a: ctypes.c_void_p(a.ctypes.data)
def best(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return min(ts), r
cap = hip.max_compressed_len_framed(n)
fr = empty(cap); w = ctypes.c_size_t(); r = ctypes.c_size_t()
def cf(inp=src, out=fr, ww=w):
    st = lib.snappy_hip_compress_framed(ctypes.cast(P(inp), ctypes.c_char_p), inp.size, P(out), out.size, ctypes.byref(ww)); assert st == 0, st
    return ww.value
res = {}
t, flen = best(cf); res["host_compress_framed_GBps"] = n / t / 1e9
back = empty(n)
def uf():
    st = lib.snappy_hip_uncompress_framed(ctypes.cast(P(fr), ctypes.c_char_p), flen, P(back), n, 1, 1, ctypes.byref(r), ctypes.byref(w)); assert st == 0, st
t, _ = best(uf); res["host_uncompress_framed_GBps"] = n / t / 1e9
assert r.value == flen and w.value == n and np.array_equal(back, src)
if n < 2**32:
    raw = empty(hip.max_compressed_len(n))
    def cr():
        st = lib.snappy_hip_compress(ctypes.cast(P(src), ctypes.c_char_p), n, P(raw), raw.size, ctypes.byref(w)); assert st == 0, st
        return w.value
    t, rlen = best(cr); res["host_compress_GBps"] = n / t / 1e9
    def ur():
        st = lib.snappy_hip_uncompress(ctypes.cast(P(raw), ctypes.c_char_p), rlen, P(back), n, ctypes.byref(w)); assert st == 0, st
    back[:] = 0
    t, _ = best(ur); res["host_uncompress_GBps"] = n / t / 1e9
    assert w.value == n and np.array_equal(back, src)
# two host threads, each its own half, each with its own output
half = n // 2
outs = [empty(cap) for _ in range(2)]
ws = [ctypes.c_size_t(), ctypes.c_size_t()]
def work(i):
    cf(src[i * half:(i + 1) * half], outs[i], ws[i])
work(0)
t1, _ = best(lambda: work(0), reps=2)
def both():
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [x.start() for x in th]; [x.join() for x in th]
t2, _ = best(both, reps=2)
res["host_two_threads_ms"] = t2 * 1e3; res["host_one_thread_half_ms"] = t1 * 1e3
if "--json" in sys.argv:
    print(json.dumps({k: round(v, 2) for k, v in res.items()}))
else:
    for k, v in res.items(): print("%-32s %8.2f" % (k, v))

"""C-ABI host calls timed without Python buffer handling (numpy buffers passed by pointer)."""
import importlib, os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import torch  # noqa: F401
hip = importlib.import_module("nim-snappy_amd")
import corpus
lib = ctypes.CDLL(hip.LIB_PATH)  # a second handle without argtypes: raw pointers
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
src = corpus.make_blocks(0, nb).reshape(-1)
n = src.size
cap = int(lib.snappy_hip_max_compressed_len_framed(n))
comp = np.empty(cap, dtype=np.uint8)
out = np.empty(n, dtype=np.uint8)
w = ctypes.c_size_t(0); r = ctypes.c_size_t(0)
P = lambda a: ctypes.c_void_p(a.ctypes.data)
lib.snappy_hip_max_compressed_len_framed.restype = ctypes.c_uint64
lib.snappy_hip_max_compressed_len_framed.argtypes = [ctypes.c_int64]
for rep in range(3):
    t0 = time.perf_counter()
    st = lib.snappy_hip_compress_framed(P(src), ctypes.c_size_t(n), P(comp), ctypes.c_size_t(cap), ctypes.byref(w))
    t1 = time.perf_counter()
    w2 = ctypes.c_size_t(0)
    st2 = lib.snappy_hip_uncompress_framed(P(comp), ctypes.c_size_t(w.value), P(out), ctypes.c_size_t(n), 1, 1, ctypes.byref(r), ctypes.byref(w2))
    t2 = time.perf_counter()
    assert st == 0 and st2 == 0 and w2.value == n
    print("rep %d: %d MiB framed: compress %.2f GB/s (%.1f ms), uncompress %.2f GB/s (%.1f ms)" % (
        rep, n >> 20, n / (t1 - t0) / 1e9, (t1 - t0) * 1e3, n / (t2 - t1) / 1e9, (t2 - t1) * 1e3), flush=True)
assert np.array_equal(out, src)
# raw buffer through snappy_hip_compress / snappy_hip_uncompress
lib.snappy_hip_max_compressed_len.restype = ctypes.c_uint64
capr = n + n // 6 + 64
compr = np.empty(capr, dtype=np.uint8)
for rep in range(3):
    t0 = time.perf_counter()
    st = lib.snappy_hip_compress(P(src), ctypes.c_size_t(n), P(compr), ctypes.c_size_t(capr), ctypes.byref(w))
    t1 = time.perf_counter()
    w2 = ctypes.c_size_t(0)
    st2 = lib.snappy_hip_uncompress(P(compr), ctypes.c_size_t(w.value), P(out), ctypes.c_size_t(n), ctypes.byref(w2))
    t2 = time.perf_counter()
    assert st == 0 and st2 == 0 and w2.value == n, (st, st2)
    print("rep %d: %d MiB raw: compress %.2f GB/s (%.1f ms), uncompress %.2f GB/s (%.1f ms)" % (
        rep, n >> 20, n / (t1 - t0) / 1e9, (t1 - t0) * 1e3, n / (t2 - t1) / 1e9, (t2 - t1) * 1e3), flush=True)
assert np.array_equal(out, src)

"""Probe: do pageable H2D and D2H copies overlap when two host threads issue them on two streams,
and what does a chunked three-stage pipeline (H2D | kernel | D2H) reach end to end?"""
import ctypes, threading, time, numpy as np, torch
hip = ctypes.CDLL("libamdhip64.so")
n = 512 << 20
a = np.random.randint(0, 255, n, dtype=np.uint8); b = np.empty_like(a)
d1 = torch.empty(n, dtype=torch.uint8, device="cuda"); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
pa = ctypes.c_void_p(a.ctypes.data); pb = ctypes.c_void_p(b.ctypes.data)
s1 = ctypes.c_void_p(); s2 = ctypes.c_void_p()
hip.hipStreamCreateWithFlags(ctypes.byref(s1), 1); hip.hipStreamCreateWithFlags(ctypes.byref(s2), 1)
def h2d(chunks=1):
    c = n // chunks
    for i in range(chunks):
        hip.hipMemcpyAsync(ctypes.c_void_p(d1.data_ptr() + i * c), ctypes.c_void_p(a.ctypes.data + i * c), ctypes.c_size_t(c), 1, s1)
    hip.hipStreamSynchronize(s1)
def d2h(chunks=1):
    c = n // chunks
    for i in range(chunks):
        hip.hipMemcpyAsync(ctypes.c_void_p(b.ctypes.data + i * c), ctypes.c_void_p(d2.data_ptr() + i * c), ctypes.c_size_t(c), 2, s2)
    hip.hipStreamSynchronize(s2)
h2d(); d2h()
t0 = time.perf_counter(); h2d(); t1 = time.perf_counter(); d2h(); t2 = time.perf_counter()
print("alone: H2D %.1f GB/s, D2H %.1f GB/s" % (n / (t1 - t0) / 1e9, n / (t2 - t1) / 1e9))
for chunks in (1, 8):
    t0 = time.perf_counter()
    th = threading.Thread(target=h2d, args=(chunks,)); th.start(); d2h(chunks); th.join()
    t = time.perf_counter() - t0
    print("two threads, %d chunk(s): both directions in %.1f ms = %.1f GB/s each way" % (chunks, t * 1e3, n / t / 1e9))
# pinned staging
ph = ctypes.c_void_p()
r = hip.hipHostMalloc(ctypes.byref(ph), ctypes.c_size_t(64 << 20), 0)
t0 = time.perf_counter()
ctypes.memmove(ph, pa, 64 << 20)
t1 = time.perf_counter()
print("memmove into pinned 64 MiB: %.1f GB/s" % ((64 << 20) / (t1 - t0) / 1e9))
for sz in (64 << 20,):
    t0 = time.perf_counter()
    hip.hipMemcpyAsync(ctypes.c_void_p(d1.data_ptr()), ph, ctypes.c_size_t(sz), 1, s1); hip.hipStreamSynchronize(s1)
    t1 = time.perf_counter()
    print("pinned H2D %d MiB: %.1f GB/s" % (sz >> 20, sz / (t1 - t0) / 1e9))
# registration cost
t0 = time.perf_counter(); r = hip.hipHostRegister(pa, ctypes.c_size_t(n), 0); t1 = time.perf_counter()
print("hipHostRegister %d MiB rc %d: %.1f ms" % (n >> 20, r, (t1 - t0) * 1e3))
if r == 0:
    t0 = time.perf_counter(); h2d(); t1 = time.perf_counter()
    print("registered H2D %.1f GB/s" % (n / (t1 - t0) / 1e9))
    t0 = time.perf_counter(); hip.hipHostUnregister(pa); print("unregister %.1f ms" % ((time.perf_counter() - t0) * 1e3))
# both directions at once from REGISTERED memory, one host thread, two streams
hip.hipHostRegister(pa, ctypes.c_size_t(n), 0); hip.hipHostRegister(pb, ctypes.c_size_t(n), 0)
for rep in range(2):
    t0 = time.perf_counter()
    hip.hipMemcpyAsync(ctypes.c_void_p(d1.data_ptr()), pa, ctypes.c_size_t(n), 1, s1)
    hip.hipMemcpyAsync(pb, ctypes.c_void_p(d2.data_ptr()), ctypes.c_size_t(n), 2, s2)
    t_issue = time.perf_counter() - t0
    hip.hipStreamSynchronize(s1); hip.hipStreamSynchronize(s2)
    t = time.perf_counter() - t0
    print("registered, both directions: issue %.2f ms, done %.1f ms = %.1f GB/s each way" % (t_issue * 1e3, t * 1e3, n / t / 1e9))
t0 = time.perf_counter()
hip.hipMemcpyAsync(ctypes.c_void_p(d1.data_ptr()), pa, ctypes.c_size_t(n), 1, s1); hip.hipStreamSynchronize(s1)
print("registered H2D alone %.1f GB/s" % (n / (time.perf_counter() - t0) / 1e9))

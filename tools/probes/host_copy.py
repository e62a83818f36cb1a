"""Probe: pageable vs registered (pinned in place) vs staged host<->device copies of a large buffer."""
import ctypes, time, numpy as np, torch
hip = ctypes.CDLL("libamdhip64.so")
n = 256 << 20
a = np.random.randint(0, 255, n, dtype=np.uint8)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
p = ctypes.c_void_p(a.ctypes.data); dp = ctypes.c_void_p(d.data_ptr())
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
h2d = lambda: hip.hipMemcpy(dp, p, ctypes.c_size_t(n), 1)
d2h = lambda: hip.hipMemcpy(p, dp, ctypes.c_size_t(n), 2)
print("pageable H2D %.1f GB/s, D2H %.1f GB/s" % (n / t(h2d) / 1e9, n / t(d2h) / 1e9))
t0 = time.perf_counter(); r = hip.hipHostRegister(p, ctypes.c_size_t(n), 0); t1 = time.perf_counter()
print("hipHostRegister rc %d: %.1f ms for %d MiB (%.1f GB/s)" % (r, (t1 - t0) * 1e3, n >> 20, n / (t1 - t0) / 1e9))
if r == 0:
    print("registered H2D %.1f GB/s, D2H %.1f GB/s" % (n / t(h2d) / 1e9, n / t(d2h) / 1e9))
    t0 = time.perf_counter(); hip.hipHostUnregister(p); print("unregister %.1f ms" % ((time.perf_counter() - t0) * 1e3))
b = np.empty_like(a)
t0 = time.perf_counter(); np.copyto(b, a); print("host memcpy %.1f GB/s (one thread)" % (n / (time.perf_counter() - t0) / 1e9))
# async copies from pageable memory on a non-blocking stream (what a library call does)
s = ctypes.c_void_p()
hip.hipStreamCreateWithFlags(ctypes.byref(s), 1)
def h2d_async():
    hip.hipMemcpyAsync(dp, p, ctypes.c_size_t(n), 1, s); hip.hipStreamSynchronize(s)
def d2h_async():
    hip.hipMemcpyAsync(p, dp, ctypes.c_size_t(n), 2, s); hip.hipStreamSynchronize(s)
print("pageable ASYNC H2D %.1f GB/s, D2H %.1f GB/s" % (n / t(h2d_async) / 1e9, n / t(d2h_async) / 1e9))

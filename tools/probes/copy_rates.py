"""Probe: what a blocking hipMemcpy between pageable / page-locked host memory and the device moves (GB/s),
one thread and three threads (a third each)."""
import ctypes, threading, time
import numpy as np, torch
hipr = ctypes.CDLL("libamdhip64.so")
hipr.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
n = 1 << 30
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, h in (("pageable", np.ones(n, np.uint8)), ("page-locked", torch.ones(n, dtype=torch.uint8).pin_memory().numpy())):
    for kind, label in ((1, "H2D"), (2, "D2H")):
        def cp(lo, hi):
            a, b = (d.data_ptr() + lo, h.ctypes.data + lo) if kind == 1 else (h.ctypes.data + lo, d.data_ptr() + lo)
            assert hipr.hipMemcpy(a, b, hi - lo, kind) == 0
        cp(0, n)
        t0 = time.perf_counter(); cp(0, n); t1 = time.perf_counter() - t0
        def three():
            th = [threading.Thread(target=cp, args=(k * (n // 3), (k + 1) * (n // 3))) for k in range(3)]
            [x.start() for x in th]; [x.join() for x in th]
        three()
        t0 = time.perf_counter(); three(); t3 = time.perf_counter() - t0
        print("%-11s %s: one thread %.1f GB/s, three threads %.1f GB/s" % (name, label, n / t1 / 1e9, n / t3 / 1e9), flush=True)

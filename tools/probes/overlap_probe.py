"""Does the index pass of one half of a batch overlap with the indexed decoder of the other half?
Two contexts (two streams) decode the two halves of a corpus, the second started d microseconds after the
first, against one context decoding the halves back to back / the whole batch in one call.  Not a test.
usage: overlap_probe.py [blocks, default 32768] [class, default MIX]"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cls = sys.argv[2] if len(sys.argv) > 2 else "MIX"
dev = torch.device("cuda", 0)
A, B = hip.Context(0), hip.Context(0)


def prepare(first, n):
    d_in = corpus.make_blocks_torch(torch, first, n, dev, only=None if cls == "MIX" else cls).reshape(-1)
    d_slots = torch.empty(n * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
    d_sizes = torch.empty(n, dtype=torch.int32, device=dev)
    d_offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
    A.encode_blocks(d_in, n * 65536, d_slots, d_sizes); A.sync()
    tot = int(d_sizes.to(torch.int64).sum().item())
    d_packed = torch.empty(tot + 64, dtype=torch.uint8, device=dev)
    A.pack(d_slots, d_sizes, n, d_packed, d_offsets); A.sync()
    del d_slots
    u = dict(n=n, d_in=d_in, packed=d_packed, off=d_offsets[:n].contiguous(), sizes=d_sizes,
             out=torch.empty(n * 65536, dtype=torch.uint8, device=dev),
             out_off=torch.arange(n, dtype=torch.int64, device=dev) * 65536,
             cap=torch.full((n,), 65536, dtype=torch.int32, device=dev),
             olen=torch.zeros(n, dtype=torch.int32, device=dev), st=torch.zeros(n, dtype=torch.int32, device=dev))
    return u


def dec(ctx, u):
    ctx.decode_blocks(u["packed"], u["off"], u["sizes"], u["n"], u["out"], u["out_off"], u["cap"], u["olen"], u["st"])


def ok(u):
    return bool((u["out"] == u["d_in"]).all().item()) and int(u["st"].abs().sum().item()) == 0


h = nb // 2
whole, h1, h2 = prepare(0, nb), prepare(0, h), prepare(h, h)
torch.cuda.synchronize()


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        A.sync(); B.sync()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


print("%s %d blocks: one call %.3f ms" % (cls, nb, timed(lambda: dec(A, whole))), "ok" if ok(whole) else "WRONG", flush=True)
print("halves back to back on one context %.3f ms" % timed(lambda: (dec(A, h1), dec(A, h2))), flush=True)
for d_us in (0, 200, 400, 600, 800, 1000, 1300, 1600, 2000):
    def both():
        # one thread: the calls only enqueue (a decode call returns before the GPU has started on it)
        t_b = time.perf_counter() + d_us * 1e-6
        dec(A, h1)
        while time.perf_counter() < t_b:
            pass
        dec(B, h2)
    print("two contexts, second half enqueued %4d us after the first: %.3f ms" % (d_us, timed(both)), "ok" if ok(h1) and ok(h2) else "WRONG", flush=True)
t0 = time.perf_counter()
for _ in range(20):
    dec(A, h1)
t1 = time.perf_counter()
A.sync()
print("enqueueing one decode call takes %.0f us on the host" % ((t1 - t0) / 20 * 1e6))

"""Probe: many decode calls of the 4 GiB corpus, one at a time; prints every call that takes unusually long
(a spin time-out in the indexed decoder shows as +16 ms per time-out) and the slowest."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
ctx = hip.Context(0)
d_in = corpus.make_blocks_torch(torch, 0, nb, dev).reshape(-1)
d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
d_sizes = torch.empty(nb, dtype=torch.int32, device=dev)
d_offsets = torch.empty(nb + 1, dtype=torch.int64, device=dev)
ctx.encode_blocks(d_in, nb * 65536, d_slots, d_sizes); ctx.sync()
tot = int(d_sizes.to(torch.int64).sum().item())
d_packed = torch.empty(tot + 64, dtype=torch.uint8, device=dev)
ctx.pack(d_slots, d_sizes, nb, d_packed, d_offsets); ctx.sync()
del d_slots
d_out = torch.empty(nb * 65536, dtype=torch.uint8, device=dev)
d_out_off = torch.arange(nb, dtype=torch.int64, device=dev) * 65536
d_out_cap = torch.full((nb,), 65536, dtype=torch.int32, device=dev)
d_out_len = torch.zeros(nb, dtype=torch.int32, device=dev)
d_status = torch.zeros(nb, dtype=torch.int32, device=dev)
d_io = d_offsets[:nb].contiguous()
burst = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # calls enqueued back to back before the synchronisation
worst, slow = 0.0, 0
for it in range(calls):
    ctx.timing(True)
    t0 = time.perf_counter()
    for _ in range(burst):
        ctx.decode_blocks(d_packed, d_io, d_sizes, nb, d_out, d_out_off, d_out_cap, d_out_len, d_status)
    ctx.sync(); t = (time.perf_counter() - t0) / burst
    s = [round(ctx.kernel_ms(k)[0], 3) for k in (4, 0, 8, 5)]
    ctx.timing(False)
    worst = max(worst, t)
    if t > 0.014:
        slow += 1
        print("call", it, "ms %.2f" % (t * 1e3), "index/ring/passed-on/one-pass", s, flush=True)
print("turns given up on:", int(ctx.kernel_ms(9)[0]))
print("calls", calls, "slow", slow, "worst ms %.2f" % (worst * 1e3), "output", "ok" if bool((d_out == d_in).all().item()) else "WRONG", flush=True)

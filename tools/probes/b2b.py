import importlib, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
nb = 65536
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
for reps in (1, 1, 3, 5):
    ctx.timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        ctx.decode_blocks(d_packed, d_io, d_sizes, nb, d_out, d_out_off, d_out_cap, d_out_len, d_status)
    ctx.sync(); t = time.perf_counter() - t0
    print("reps", reps, "ms/step %.2f" % (t / reps * 1e3), "slots", [round(ctx.kernel_ms(k)[0], 3) for k in (4, 0, 8, 5)],
          "ok" if bool((d_out == d_in).all().item()) else "WRONG", flush=True)
    ctx.timing(False)

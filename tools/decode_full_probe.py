"""Decode step of the bench's full corpus: kernel times (index pass, ring launch, passed-on launch) and the step.  Not a test."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
ctx = hip.Context(0)
d_in = torch.empty(nb * 65536, dtype=torch.uint8, device=dev)
for b0 in range(0, nb, 4096):
    c = min(4096, nb - b0)
    d_in[b0 * 65536:(b0 + c) * 65536] = corpus.make_blocks_torch(torch, b0, c, dev).reshape(-1)
d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
d_sizes = torch.empty(nb, dtype=torch.int32, device=dev)
d_offsets = torch.empty(nb + 1, dtype=torch.int64, device=dev)
ctx.encode_blocks(d_in, nb * 65536, d_slots, d_sizes); ctx.sync()
tot = int(d_sizes.to(torch.int64).sum().item())
d_packed = torch.empty(tot + 64, dtype=torch.uint8, device=dev)
ctx.pack(d_slots, d_sizes, nb, d_packed, d_offsets); ctx.sync()
del d_slots
d_out = torch.empty(nb * 65536, dtype=torch.uint8, device=dev)
d_oo = torch.arange(nb, dtype=torch.int64, device=dev) * 65536
d_oc = torch.full((nb,), 65536, dtype=torch.int32, device=dev)
d_ol = torch.zeros(nb, dtype=torch.int32, device=dev)
d_st = torch.zeros(nb, dtype=torch.int32, device=dev)
d_io = d_offsets[:nb].contiguous()
def step(): ctx.decode_blocks(d_packed, d_io, d_sizes, nb, d_out, d_oo, d_oc, d_ol, d_st)
for _ in range(2): step()
ctx.sync(); ctx.timing(True); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
ctx.sync(); t = (time.perf_counter() - t0) / 10
print("step %.3f ms (%.1f GB/s): index %.3f ring %.3f passed-on %.3f" % (t * 1e3, nb * 65536 / t / 1e9, ctx.kernel_ms(4)[0], ctx.kernel_ms(0)[0], ctx.kernel_ms(8)[0]), flush=True)
assert bool(torch.equal(d_out, d_in))

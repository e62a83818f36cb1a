"""Does the order of the units matter?  Decode the default mix in corpus order and sorted by compressed length. Not a test."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
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
off0 = d_offsets[:nb].contiguous()
out_off0 = torch.arange(nb, dtype=torch.int64, device=dev) * 65536
cap0 = torch.full((nb,), 65536, dtype=torch.int32, device=dev)
cls = torch.from_numpy(corpus.block_classes(0, nb)).to(dev)
heavy = torch.nonzero(cls <= 1).flatten()
cheap = torch.nonzero(cls > 1).flatten()
k = min(heavy.numel(), cheap.numel())
alt = torch.stack([heavy[:k], cheap[:k]], dim=1).reshape(-1)
alt = torch.cat([alt, heavy[k:], cheap[k:]])
orders = {"corpus order": torch.arange(nb, device=dev),
          "text/html alternating with the rest": alt,
          "by compressed length, ascending": torch.argsort(d_sizes.to(torch.int64)),
          "by compressed length, descending": torch.argsort(d_sizes.to(torch.int64), descending=True)}
for name, perm in orders.items():
    in_off = off0[perm].contiguous(); in_len = d_sizes[perm].contiguous()
    out_off = out_off0[perm].contiguous(); cap = cap0[perm].contiguous()
    d_out_len = torch.zeros(nb, dtype=torch.int32, device=dev)
    d_status = torch.zeros(nb, dtype=torch.int32, device=dev)
    for it in range(3):
        d_out.zero_()
        ctx.timing(True)
        ctx.decode_blocks(d_packed, in_off, in_len, nb, d_out, out_off, cap, d_out_len, d_status)
        ctx.sync()
        ms, _ = ctx.kernel_ms(0); ims, _ = ctx.kernel_ms(4)
        ctx.timing(False)
    ok = bool(torch.equal(d_out, d_in)) and int((d_status != 0).sum().item()) == 0
    print("%-36s decode %.3f ms  index %.3f ms  ok %s" % (name, ms, ims, ok), flush=True)

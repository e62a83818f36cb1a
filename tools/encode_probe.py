"""Timing of the encode kernel per corpus class. Not a test."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
classes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["T_TEXT", "T_HTML"]
dev = torch.device("cuda", 0)
ctx = hip.Context(0)
for cls in classes:
    only = None if cls == "MIX" else cls
    d_in = corpus.make_blocks_torch(torch, 0, nb, dev, only=only).reshape(-1)
    d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
    d_sizes = torch.empty(nb, dtype=torch.int32, device=dev)
    for it in range(2):
        ctx.timing(True)
        ctx.encode_blocks(d_in, nb * 65536, d_slots, d_sizes)
        ctx.sync()
        ms, n = ctx.kernel_ms(1)
        ctx.timing(False)
    tot = int(d_sizes.to(torch.int64).sum().item())
    print(cls, "encode ms %.3f" % ms, "GB/s %.2f" % (nb * 65536 / ms / 1e6),
          "per-block ms (1024 concurrent) %.3f" % (ms * 1024 / nb), "C/block %d" % (tot // nb), flush=True)

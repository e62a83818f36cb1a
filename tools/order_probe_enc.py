"""Does the order of the blocks matter to the encoder?  The default mix in corpus order and grouped by class. Not a test."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
ctx = hip.Context(0)
d_in = torch.empty(nb * 65536, dtype=torch.uint8, device=dev)
for b0 in range(0, nb, 4096):
    c = min(4096, nb - b0)
    d_in[b0 * 65536:(b0 + c) * 65536] = corpus.make_blocks_torch(torch, b0, c, dev).reshape(-1)
cls = torch.from_numpy(corpus.block_classes(0, nb)).to(dev)
blocks = d_in.view(nb, 65536)
variants = {"corpus order": blocks,
            "grouped by class": blocks[torch.argsort(cls, stable=True)].contiguous(),
            "heavy classes last": blocks[torch.argsort(-cls, stable=True)].contiguous()}
d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
d_sizes = torch.empty(nb, dtype=torch.int32, device=dev)
for name, b in variants.items():
    flat = b.reshape(-1)
    for it in range(2):
        ctx.timing(True)
        ctx.encode_blocks(flat, nb * 65536, d_slots, d_sizes)
        ctx.sync()
        ms, _ = ctx.kernel_ms(1)
        ctx.timing(False)
    print("%-22s encode %.2f ms  %.2f GB/s" % (name, ms, nb * 65536 / ms / 1e6), flush=True)

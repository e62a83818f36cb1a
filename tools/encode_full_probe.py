"""Encode kernel time on the bench's full corpus (65 536 blocks of the class mix), per g (second waves per four workgroups).  Not a test."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
ctx = hip.Context(0)
if os.environ.get('PROBE_NO_ORDER'): ctx.launch_order(False)
d_in = torch.empty(nb * 65536, dtype=torch.uint8, device=dev)
for b0 in range(0, nb, 4096):
    c = min(4096, nb - b0)
    d_in[b0 * 65536:(b0 + c) * 65536] = corpus.make_blocks_torch(torch, b0, c, dev).reshape(-1)
d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
d_sizes = torch.empty(nb, dtype=torch.int32, device=dev)
for it in range(3):
    ctx.timing(True)
    ctx.encode_blocks(d_in, nb * 65536, d_slots, d_sizes)
    ctx.sync()
    ms, n = ctx.kernel_ms(1)
    ctx.timing(False)
print("MIX %d blocks: encode ms %.3f GB/s %.2f" % (nb, ms, nb * 65536 / ms / 1e6), flush=True)

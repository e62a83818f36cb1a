"""Find blocks whose GPU encoding differs from the oracle's (debug aid, not a test)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
hip = importlib.import_module("nim-snappy_amd")
import corpus, pyoracle as orc
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
only = sys.argv[2] if len(sys.argv) > 2 else None
ctx = hip.Context(0)
bad = 0
for b0 in range(0, nb, 2048):
    c = min(2048, nb - b0)
    blocks = corpus.make_blocks(b0, c, only=only)
    d_in = torch.from_numpy(blocks.reshape(-1)).cuda()
    d_slots = torch.empty(c * hip.SLOT_STRIDE, dtype=torch.uint8, device="cuda")
    d_sizes = torch.empty(c, dtype=torch.int32, device="cuda")
    ctx.encode_blocks(d_in, c * 65536, d_slots, d_sizes, unit=hip.UNIT_BODY); ctx.sync()
    slots = d_slots.cpu().numpy().reshape(c, hip.SLOT_STRIDE); sizes = d_sizes.cpu().numpy()
    for i in range(c):
        want = orc.encode_block(blocks[i].tobytes())
        got = slots[i, :sizes[i]].tobytes()
        if got != want:
            k = next((j for j in range(min(len(got), len(want))) if got[j] != want[j]), min(len(got), len(want)))
            print("block", b0 + i, "class", corpus.CLASSES[corpus.block_classes(b0 + i, 1)[0]] if only is None else only,
                  "len got/want", len(got), len(want), "first diff at", k, got[max(0,k-4):k+8].hex(), want[max(0,k-4):k+8].hex(), flush=True)
            bad += 1
            if bad > 5: sys.exit(1)
print("checked", nb, "bad", bad)

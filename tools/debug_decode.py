"""Debug helper: run the v2 decode per class, report bad units and dump them. Not a test."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
nb = int(sys.argv[1]); classes = sys.argv[2].split(","); first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
ctx = hip.Context(0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for cls in classes:
    only = None if cls == "MIX" else cls
    d_in = corpus.make_blocks_torch(torch, first, nb, dev, only=only).reshape(-1)
    d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
    d_sizes = torch.empty(nb, dtype=torch.int32, device=dev)
    d_offsets = torch.empty(nb + 1, dtype=torch.int64, device=dev)
    ctx.encode_blocks(d_in, nb * 65536, d_slots, d_sizes); ctx.sync()
    tot = int(d_sizes.to(torch.int64).sum().item())
    d_packed = torch.zeros(tot + 64, dtype=torch.uint8, device=dev)
    ctx.pack(d_slots, d_sizes, nb, d_packed, d_offsets); ctx.sync()
    d_out = torch.zeros(nb * 65536, dtype=torch.uint8, device=dev)
    d_out_off = torch.arange(nb, dtype=torch.int64, device=dev) * 65536
    d_out_cap = torch.full((nb,), 65536, dtype=torch.int32, device=dev)
    d_out_len = torch.zeros(nb, dtype=torch.int32, device=dev)
    d_status = torch.full((nb,), 77, dtype=torch.int32, device=dev)
    t0 = time.time()
    ctx.decode_blocks(d_packed, d_offsets[:nb].contiguous(), d_sizes, nb, d_out, d_out_off, d_out_cap, d_out_len, d_status)
    ctx.sync()
    dt = time.time() - t0
    st = d_status.cpu().numpy(); ol = d_out_len.cpu().numpy()
    ok_bytes = (d_out.view(nb, 65536) == d_in.view(nb, 65536)).all(dim=1).cpu().numpy()
    bad = np.nonzero((st != 0) | (ol != 65536) | (~ok_bytes))[0]
    print(cls, "time %.3f s" % dt, "bad units:", len(bad), bad[:10], "status", st[bad[:10]], "len", ol[bad[:10]], flush=True)
    offs = d_offsets.cpu().numpy(); sizes = d_sizes.cpu().numpy()
    packed = d_packed.cpu().numpy()
    for b in bad[:3]:
        open(os.path.join(ROOT, "gpurun_out", "bad_%s_%d.bin" % (cls, b)), "wb").write(packed[offs[b]:offs[b] + sizes[b]].tobytes())
        got = d_out.view(nb, 65536)[b].cpu().numpy(); want = d_in.view(nb, 65536)[b].cpu().numpy()
        diff = np.nonzero(got != want)[0]
        print("  unit", b, "first diffs at", diff[:8], "count", len(diff), flush=True)
        for d0 in sorted(set((int(x) // 16) * 16 for x in diff[:40]))[:4]:
            print("    @%d got  %s" % (d0 - 8, got[d0 - 8:d0 + 24].tobytes().hex()))
            print("    @%d want %s" % (d0 - 8, want[d0 - 8:d0 + 24].tobytes().hex()), flush=True)

"""(historic: the -DD2_WINDOW / -DD2_MINWAVES builds it compared existed at commit 72bbb48; the result --
three workgroups per CU scale -- became the ring-window instantiation of decode2_kernel.h)
Probe: does the decode kernel scale with more workgroups per CU?  Units of 32 KiB blocks decoded with
the 64 KiB window (two workgroups per CU) and with a 32 KiB window (three per CU): same work per block."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
BL = int(os.environ.get("PROBE_BLOCK", "32768"))
nb64 = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = hip.Context(0)
for cls in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["T_TEXT", "T_HTML"]):
    d_in = corpus.make_blocks_torch(torch, 0, nb64, "cuda", only=cls).reshape(-1)
    nb = nb64 * 65536 // BL
    d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device="cuda")
    d_sizes = torch.empty(nb, dtype=torch.int32, device="cuda")
    d_offsets = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
    ctx.encode_blocks(d_in, nb64 * 65536, d_slots, d_sizes, block_len=BL); ctx.sync()
    tot = int(d_sizes.to(torch.int64).sum().item())
    d_packed = torch.empty(tot + 64, dtype=torch.uint8, device="cuda")
    ctx.pack(d_slots, d_sizes, nb, d_packed, d_offsets); ctx.sync()
    d_out = torch.empty(nb * BL, dtype=torch.uint8, device="cuda")
    d_out_off = torch.arange(nb, dtype=torch.int64, device="cuda") * BL
    d_out_cap = torch.full((nb,), BL, dtype=torch.int32, device="cuda")
    d_out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_status = torch.zeros(nb, dtype=torch.int32, device="cuda")
    for it in range(3):
        ctx.timing(True)
        ctx.decode_blocks(d_packed, d_offsets[:nb].contiguous(), d_sizes, nb, d_out, d_out_off, d_out_cap, d_out_len, d_status)
        ctx.sync()
        ms, n = ctx.kernel_ms(0); ims, _ = ctx.kernel_ms(4)
        ctx.timing(False)
    assert bool(torch.equal(d_out, d_in)) and int((d_status != 0).sum().item()) == 0
    print(cls, "block", BL, "decode kernel ms %.3f (launches %d) index ms %.3f" % (ms, n, ims), "GB/s %.0f" % (nb * BL / ms / 1e6), flush=True)

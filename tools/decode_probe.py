"""Timing experiments on the decode kernel (phase ablation via SNAPPY_HIP_DBG). Not a test."""
import importlib, os, sys, time
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
    d_in = corpus.make_blocks_torch(torch, 0, nb, dev, only=None if cls == "MIX" else cls).reshape(-1)
    d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
    d_sizes = torch.empty(nb, dtype=torch.int32, device=dev)
    d_offsets = torch.empty(nb + 1, dtype=torch.int64, device=dev)
    ctx.encode_blocks(d_in, nb * 65536, d_slots, d_sizes); ctx.sync()
    tot = int(d_sizes.to(torch.int64).sum().item())
    d_packed = torch.empty(tot + 64, dtype=torch.uint8, device=dev)
    ctx.pack(d_slots, d_sizes, nb, d_packed, d_offsets); ctx.sync()
    d_out = torch.empty(nb * 65536, dtype=torch.uint8, device=dev)
    d_out_off = torch.arange(nb, dtype=torch.int64, device=dev) * 65536
    d_out_cap = torch.full((nb,), 65536, dtype=torch.int32, device=dev)
    d_out_len = torch.zeros(nb, dtype=torch.int32, device=dev)
    d_status = torch.zeros(nb, dtype=torch.int32, device=dev)
    for dbg in os.environ.get("PROBE_DBG", "0,1,2,3,7").split(","):
        os.environ["SNAPPY_HIP_DBG"] = dbg
        for it in range(2):
            ctx.timing(True)
            ctx.decode_blocks(d_packed, d_offsets[:nb].contiguous(), d_sizes, nb, d_out, d_out_off,
                              d_out_cap, d_out_len, d_status)
            ctx.sync()
            ms, n = ctx.kernel_ms(0)
            ims, _ = ctx.kernel_ms(4)
            ms8, _ = ctx.kernel_ms(8)
            ms5, _ = ctx.kernel_ms(5)
            ctx.timing(False)
        print(cls, "dbg", dbg, "ms %.3f" % ms, "index ms %.3f" % ims, "per-block us (512 concurrent) %.1f" % (ms * 1e3 * 512 / nb),
              "C/block %d" % (tot // nb), "second launch ms %.3f" % ms8, "units decoded by the index pass (running) %d" % ctx.kernel_ms(10)[1], "one-pass ms %.3f" % ms5,
              "ok" if bool((d_out == d_in).all().item()) and int(d_status.abs().sum().item()) == 0 else "WRONG", flush=True)

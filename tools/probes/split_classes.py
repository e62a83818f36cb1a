"""Probe: snappy_hip_uncompress_d of ONE raw buffer of 1024 blocks, class by class (which data needs how many looks)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
ctx = hip.Context(0)
nb = 1024
for only in (None, "T_TEXT", "T_HTML", "RS", "R", "P10", "Z", "RAMP"):
    src = corpus.make_blocks(0, nb, only=only).tobytes()
    raw = hip.encode(src)
    d_in = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    d_out = torch.empty(len(src), dtype=torch.uint8, device="cuda")
    st, w = ctx.uncompress(d_in, len(raw), d_out, len(src))
    assert (st, w) == (0, len(src)) and d_out.cpu().numpy().tobytes() == src, only
    ts = []
    for _ in range(3):
        ctx.timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.uncompress(d_in, len(raw), d_out, len(src))
        ts.append(time.perf_counter() - t0)
        ms, k = ctx.kernel_ms(7); ctx.timing(False)
    print("%-7s: %.2f ms = %.1f GB/s of output; walk rounds %d x %.3f ms" % (only or "mix", min(ts) * 1e3, len(src) / min(ts) / 1e9, k, ms), flush=True)

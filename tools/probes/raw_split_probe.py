"""Probe: uncompress of ONE raw multi-block buffer resident in HBM (snappy_hip_uncompress_d)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, torch
hip = importlib.import_module("nim-snappy_amd")
import corpus
ctx = hip.Context(0)
for nb, only in ((1024, None), (1024, "T_TEXT"), (1024, "R"), (1024, "T_TEXT+R"), (16384, None), (16384, "R")):
    mix = [50, 0, 0, 50, 0, 0, 0] if only == "T_TEXT+R" else None
    src = corpus.make_blocks(0, nb, only=None if mix else only, mix=mix).tobytes()
    raw = hip.encode(src)
    d_in = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    d_out = torch.empty(len(src), dtype=torch.uint8, device="cuda")
    st, w = ctx.uncompress(d_in, len(raw), d_out, len(src))
    assert (st, w) == (0, len(src)) and d_out.cpu().numpy().tobytes() == src
    ts = []
    for _ in range(3):
        ctx.timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.uncompress(d_in, len(raw), d_out, len(src))
        ts.append(time.perf_counter() - t0)
        ms, k = ctx.kernel_ms(7); ctx.timing(False)
    print("%5d blocks %-7s: %.2f ms = %.1f GB/s of output; split rounds %d x %.3f ms" % (nb, only or "mix", min(ts) * 1e3, len(src) / min(ts) / 1e9, k, ms), flush=True)

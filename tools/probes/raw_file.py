"""Probe: snappy_hip_uncompress_d of one golden file's raw stream (debug library: SNAPPY_HIP_STATS=1 prints the split's rounds)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import torch
hip = importlib.import_module("nim-snappy_amd")
from conftest import golden_file
name = sys.argv[1]
src = golden_file(name) * (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
raw = hip.encode(src)
ctx = hip.Context(0)
d_in = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
d_out = torch.empty(len(src), dtype=torch.uint8, device="cuda")
st, w = ctx.uncompress(d_in, len(raw), d_out, len(src))
got = d_out.cpu().numpy().tobytes()
print("status", st, "written", w, "of", len(src), "equal", got == src)
if got != src:
    bad = [i for i in range(0, len(src), 65536) if got[i:i + 65536] != src[i:i + 65536]]
    print("blocks that differ:", bad[:20], "of", (len(src) + 65535) // 65536)

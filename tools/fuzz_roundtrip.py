"""Longer structured-fuzz campaign than the test-suite runs: decode(encode(x)) == x on the device,
encoder == oracle on a sample.  usage: python tools/fuzz_roundtrip.py <blocks> <seed> [<seed> ...]"""
import importlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
hip = importlib.import_module("nim-snappy_amd")
import pyoracle as orc
from test_gpu_batch import _structured_block, _encode_pack
nb = int(sys.argv[1])
for seed in map(int, sys.argv[2:]):
    rng = random.Random(seed)
    blocks = [_structured_block(rng) for _ in range(nb)]
    flat = torch.from_numpy(np.frombuffer(b"".join(blocks), dtype=np.uint8).copy()).cuda()
    ctx = hip.Context(0)
    d_slots, d_sizes, d_offsets, d_out, total = _encode_pack(hip, torch, ctx, flat, flat.numel(), hip.UNIT_RAW)
    sizes = d_sizes.cpu().numpy(); packed = d_out.cpu().numpy(); offs = d_offsets.cpu().numpy()
    enc_bad = [i for i in range(0, nb, 5) if packed[offs[i]:offs[i] + sizes[i]].tobytes() != orc.encode(blocks[i])]
    d_out_off = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
    d_out_cap = torch.full((nb,), 65536, dtype=torch.int32, device="cuda")
    d_out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_status = torch.full((nb,), 77, dtype=torch.int32, device="cuda")
    d_dec = torch.zeros(nb * 65536, dtype=torch.uint8, device="cuda")
    ctx.decode_blocks(d_out, d_offsets[:nb].contiguous(), d_sizes, nb, d_dec, d_out_off, d_out_cap, d_out_len,
                      d_status, unit=hip.UNIT_RAW)
    ctx.sync()
    st_bad = int((d_status != 0).sum().item())
    dec = d_dec.cpu().numpy()
    dec_bad = [i for i in range(nb) if dec[i * 65536:(i + 1) * 65536].tobytes() != blocks[i]]
    print("seed", seed, "blocks", nb, "encoder mismatches", enc_bad[:5], "decode status errors", st_bad,
          "decode mismatches", dec_bad[:5], "ratio %.3f" % (total / flat.numel()), flush=True)
    ctx.close()

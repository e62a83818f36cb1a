"""Mutation fuzz at scale through the batch API: per-unit status, length and bytes of the HIP decoder
against the oracle on damaged streams.  usage: python tools/fuzz_mutations.py <units> <seed>..."""
import importlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
hip = importlib.import_module("nim-snappy_amd")
import pyoracle as orc
import corpus
from test_gpu_batch import _structured_block
nu = int(sys.argv[1])
for seed in map(int, sys.argv[2:]):
    rng = random.Random(seed)
    base_blocks = [corpus.make_blocks(rng.randrange(100000), 1)[0].tobytes() for _ in range(24)]
    base_blocks += [_structured_block(rng) for _ in range(8)]
    base_blocks += [b[:rng.choice([100, 5000, 30000])] for b in base_blocks[:6]]
    encs = [orc.encode(b) for b in base_blocks]
    units = []
    for i in range(nu):
        m = bytearray(rng.choice(encs))
        k = rng.random()
        if k < 0.6:
            for _ in range(rng.randint(1, 3)):
                m[rng.randrange(len(m))] = rng.randrange(256)
        elif k < 0.8:
            m = m[:rng.randrange(1, len(m))]
        elif k < 0.9:
            pos = rng.randrange(len(m)); m[pos:pos] = rng.randbytes(rng.randint(1, 4))
        else:
            pos = rng.randrange(len(m)); del m[pos:pos + rng.randint(1, 4)]
        units.append(bytes(m))
    want = []
    for m in units:
        n = orc.uncompressed_len(m)
        cap = 65536
        st, out = orc.uncompress(m, cap)
        want.append((st, out))
    blob = b"".join(units)
    offs = np.cumsum([0] + [len(u) for u in units])[:-1].astype(np.int64)
    lens = np.array([len(u) for u in units], dtype=np.int32)
    d_in = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda()
    d_off = torch.from_numpy(offs).cuda(); d_len = torch.from_numpy(lens).cuda()
    d_out = torch.zeros(nu * 65536, dtype=torch.uint8, device="cuda")
    d_out_off = torch.arange(nu, dtype=torch.int64, device="cuda") * 65536
    d_cap = torch.full((nu,), 65536, dtype=torch.int32, device="cuda")
    d_ol = torch.zeros(nu, dtype=torch.int32, device="cuda"); d_st = torch.full((nu,), 77, dtype=torch.int32, device="cuda")
    ctx = hip.Context(0)
    ctx.decode_blocks(d_in, d_off, d_len, nu, d_out, d_out_off, d_cap, d_ol, d_st, unit=hip.UNIT_RAW)
    ctx.sync()
    st = d_st.cpu().numpy(); ol = d_ol.cpu().numpy(); out = d_out.cpu().numpy()
    bad = []
    for i in range(nu):
        ws, wo = want[i]
        if st[i] != ws or (ws == 0 and (ol[i] != len(wo) or out[i * 65536:i * 65536 + ol[i]].tobytes() != wo)):
            bad.append((i, int(st[i]), ws, int(ol[i]), len(wo)))
    nok = sum(1 for w in want if w[0] == 0)
    print("seed", seed, "units", nu, "still-valid streams", nok, "mismatches", bad[:5], flush=True)
    ctx.close()

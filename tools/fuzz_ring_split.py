"""Longer fuzz campaign for the ring-window decoder and the raw multi-block splitter than the test-suite runs.
usage: python tools/fuzz_ring_split.py <seed> [<seed> ...]
 (a) foreign tag streams (tests/test_gpu_batch.py::_ring_stream, every style, random lengths and placements)
     through decode_blocks with and without the fused CRC -- both instantiations of the indexed decoder;
 (b) raw buffers of several blocks assembled from text, random bytes, periods, repeated strings and zeros in
     random proportions: hip.decode == source, and with one byte flipped == the oracle's verdict."""
import importlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
hip = importlib.import_module("nim-snappy_amd")
import pyoracle as orc
from test_gpu_batch import _ring_stream, _dev

text = b"".join(open(os.path.join(ROOT, "tests", "golden", "data", f), "rb").read() for f in ("alice29.txt", "html", "urls.10K"))


def ring_case(rng, ctx):
    units = []
    for i in range(rng.randint(40, 160)):
        style = rng.choice(["text", "biglit", "dense", "mid"])
        units.append(_ring_stream(rng, rng.choice([65536, 65536, rng.randint(1, 65536)]), style))
    in_off, out_off, pos, opos = [], [], 0, 0
    reps = rng.randint(1, 8)
    for r in range(reps):
        for b, p in units:
            in_off.append(pos); pos += len(b) + rng.randint(0, 9)
            opos = (opos + 15) & ~15 if rng.random() < 0.7 else opos + rng.randint(0, 15)
            out_off.append(opos); opos += len(p)
    n = len(units); nu = n * reps
    stream = np.zeros(pos + 64, np.uint8); want = np.zeros(opos, np.uint8)
    for j in range(nu):
        b, p = units[j % n]
        stream[in_off[j]:in_off[j] + len(b)] = np.frombuffer(b, np.uint8)
        want[out_off[j]:out_off[j] + len(p)] = np.frombuffer(p, np.uint8)
    d = lambda a, t: _dev(torch, np.array(a, t))
    bad = 0
    for with_crc in (False, True):
        d_len = torch.zeros(nu, dtype=torch.int32, device="cuda")
        d_st = torch.full((nu,), 77, dtype=torch.int32, device="cuda")
        d_dec = torch.zeros(opos, dtype=torch.uint8, device="cuda")
        d_crc = torch.zeros(nu, dtype=torch.int32, device="cuda") if with_crc else None
        ctx.decode_blocks(_dev(torch, stream), d(in_off, np.int64), d([len(units[j % n][0]) for j in range(nu)], np.int32), nu,
                          d_dec, d(out_off, np.int64), d([len(units[j % n][1]) for j in range(nu)], np.int32), d_len, d_st,
                          unit=hip.UNIT_BODY, d_crc=d_crc)
        ctx.sync()
        bad += int((d_st != 0).sum().item()) + int((d_dec.cpu().numpy() != want).sum())
        if with_crc:
            crcs = d_crc.cpu().numpy().view(np.uint32)
            bad += sum(int(crcs[j]) != orc.masked_crc(units[j % n][1]) for j in range(0, nu, 53))
    return nu, bad


def split_case(rng):
    parts = []
    for _ in range(rng.randint(2, 14)):
        k = rng.random(); n = rng.choice([1, 100, 5000, 70000, 200000, rng.randint(1, 400000)])
        o = rng.randrange(len(text) - 400000)
        if k < 0.3: parts.append(text[o:o + n])
        elif k < 0.5: parts.append(rng.randbytes(n))
        elif k < 0.7:
            per = rng.choice([1, 2, 3, 10, 14, 18, 22, 254, 255, 256, 300, 4097])
            parts.append((text[o:o + per] * (n // per + 1))[:n])
        elif k < 0.85:
            st = rng.randbytes(rng.randint(500, 9000)); parts.append((st * (n // len(st) + 1))[:n])
        else: parts.append(bytes(n))
    src = b"".join(parts)
    comp = orc.encode(src)
    bad = int(hip.decode(comp) != src)
    for _ in range(2):
        m = bytearray(comp); m[rng.randrange(len(m))] ^= 1 << rng.randrange(8)
        bad += int(hip.decode(bytes(m)) != orc.decode(bytes(m)))
    cut = comp[:rng.randrange(1, len(comp))]
    bad += int(hip.decode(cut) != orc.decode(cut))
    return len(src), bad


for seed in map(int, sys.argv[1:]):
    rng = random.Random(seed)
    ctx = hip.Context(0)
    nu, bad_r = ring_case(rng, ctx)
    tot, bad_s, cases = 0, 0, 0
    for _ in range(12):
        n, b = split_case(rng); tot += n; bad_s += b; cases += 1
    print("seed", seed, "ring units", nu, "ring mismatches", bad_r, "| raw buffers", cases, "bytes", tot, "mismatches", bad_s, flush=True)
    ctx.close()

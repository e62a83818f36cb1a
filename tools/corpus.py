"""Seeded synthetic many-block corpus for the parity tests and bench.py.

Every 65 536-byte block is generated from (seed, block index) alone, so any block can be
reproduced in isolation.  PRNG = splitmix64.  Classes and the default mix:

  T_TEXT 25 %  65 536-byte window of the concatenated text fixtures (alice29, asyoulik, lcet10,
               plrabn12) at a seeded offset                       -- ~13-15 k elements / block
  T_HTML 25 %  same over html, urls.10K, geo.protodata, kppkn.gtb   -- ~3-7 k elements / block
  RS     20 %  random strings of length 1000..10000, each repeated 2..4 times
               (tests/test_snappy.nim:247-253 strings, made compressible by repetition)
  R      10 %  uniform random bytes (tests/randgen.nim:21-24): incompressible
  P10    10 %  byte('a' + j mod 10)          (tests/test_snappy.nim:118-121)
  Z       5 %  zeros                         (tests/test_snappy.nim:126)
  RAMP    5 %  byte(i)                       (tests/test_framed.nim:141-144)

The text/html sources are the reference's own test data, committed as fixtures under
tests/golden/data (nothing is read from the reference checkout at run time).
"""
import os

import numpy as np

BLOCK = 65536
SEED = 0x5EED5AA9
CLASSES = ["T_TEXT", "T_HTML", "RS", "R", "P10", "Z", "RAMP"]
MIX = [25, 25, 20, 10, 10, 5, 5]  # percent, in CLASSES order
_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests",
                       "golden", "data")
_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    """splitmix64 output function, vectorised (uint64 wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream(state, count):
    """`count` successive splitmix64 outputs for each state in `state` (shape [...])."""
    with np.errstate(over="ignore"):
        k = (np.arange(1, count + 1, dtype=np.uint64) * _G)
        return _mix(np.asarray(state, dtype=np.uint64)[..., None] + k)


_sources = {}


def _source(kind):
    if kind not in _sources:
        names = {"T_TEXT": ["alice29.txt", "asyoulik.txt", "lcet10.txt", "plrabn12.txt"],
                 "T_HTML": ["html", "urls.10K", "geo.protodata", "kppkn.gtb"]}[kind]
        buf = b"".join(open(os.path.join(_GOLDEN, n), "rb").read() for n in names)
        _sources[kind] = np.frombuffer(buf, dtype=np.uint8)
    return _sources[kind]


def block_classes(first, count, seed=SEED, mix=None):
    """Class index of blocks first .. first+count-1."""
    mix = MIX if mix is None else mix
    idx = np.arange(first, first + count, dtype=np.uint64)
    r = _mix((np.uint64(seed) ^ idx) + _G) % np.uint64(100)
    edges = np.cumsum(mix)
    return np.searchsorted(edges, r.astype(np.int64), side="right").astype(np.int32)


def make_blocks(first, count, seed=SEED, mix=None, only=None):
    """uint8 array [count, 65536] with blocks first .. first+count-1 of the corpus.

    only = class name: every block is of that class (per-class benchmarks)."""
    out = np.empty((count, BLOCK), dtype=np.uint8)
    if only is not None:
        cls = np.full(count, CLASSES.index(only), dtype=np.int32)
    else:
        cls = block_classes(first, count, seed, mix)
    idx = np.arange(first, first + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        state = _mix((np.uint64(seed) ^ idx) * np.uint64(3) + _G)  # per-block stream state
    p10 = (np.arange(BLOCK) % 10 + ord("a")).astype(np.uint8)
    ramp = (np.arange(BLOCK) & 0xff).astype(np.uint8)
    for ci, name in enumerate(CLASSES):
        sel = np.nonzero(cls == ci)[0]
        if sel.size == 0:
            continue
        if name in ("T_TEXT", "T_HTML"):
            src = _source(name)
            off = (_mix(state[sel] + _G) % np.uint64(src.size - BLOCK)).astype(np.int64)
            for j, o in zip(sel, off):
                out[j] = src[o:o + BLOCK]
        elif name == "R":
            for c0 in range(0, sel.size, 1024):  # bound the temporary
                s = sel[c0:c0 + 1024]
                out[s] = _stream(state[s], BLOCK // 8).view(np.uint8).reshape(len(s), BLOCK)
        elif name == "RS":
            for j in sel:
                words = _stream(state[j], BLOCK // 8 + 64)
                rnd = words[:BLOCK // 8].view(np.uint8)
                ctl = words[BLOCK // 8:]
                pos, used, k = 0, 0, 0
                row = out[j]
                while pos < BLOCK:
                    length = 1000 + int(ctl[k % 64] % np.uint64(9001))
                    reps = 2 + int((ctl[k % 64] >> np.uint64(32)) % np.uint64(3))
                    k += 1
                    s = rnd[used:used + length]
                    used += length
                    for _ in range(reps):
                        take = min(len(s), BLOCK - pos)
                        row[pos:pos + take] = s[:take]
                        pos += take
                        if pos >= BLOCK:
                            break
        elif name == "P10":
            out[sel] = p10
        elif name == "Z":
            out[sel] = 0
        elif name == "RAMP":
            out[sel] = ramp
    return out


def class_counts(n_blocks, seed=SEED, mix=None):
    cls = block_classes(0, n_blocks, seed, mix)
    return {name: int((cls == i).sum()) for i, name in enumerate(CLASSES)}


# ---- torch mirror (device-side generation for bench.py; equality with the numpy generator is
# ---- a test: tests/test_corpus.py) -------------------------------------------------------------
def _t_const(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _t_lsr(torch, z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def _t_mix(torch, z):
    z = (z ^ _t_lsr(torch, z, 30)) * _t_const(int(_M1))
    z = (z ^ _t_lsr(torch, z, 27)) * _t_const(int(_M2))
    return z ^ _t_lsr(torch, z, 31)


def _t_umod(torch, z, m):
    """unsigned 64-bit z mod m for int64-held bit patterns (m < 2^31)"""
    lo = z & 0xFFFFFFFF
    hi = _t_lsr(torch, z, 32)
    return ((hi % m) * ((1 << 32) % m) + (lo % m)) % m


def make_blocks_torch(torch, first, count, device, seed=SEED, mix=None, only=None, chunk=1024):
    """Same bytes as make_blocks(), generated with torch ops on `device` (uint8 [count, 65536])."""
    mix = MIX if mix is None else mix
    G = _t_const(int(_G))
    out = torch.empty((count, BLOCK), dtype=torch.uint8, device=device)
    idx = torch.arange(first, first + count, dtype=torch.int64, device=device)
    if only is not None:
        cls = torch.full((count,), CLASSES.index(only), dtype=torch.int64, device=device)
    else:
        r = _t_umod(torch, _t_mix(torch, (idx ^ _t_const(seed)) + G), 100)
        edges = torch.tensor(np.cumsum(mix), dtype=torch.int64, device=device)
        cls = torch.searchsorted(edges, r, right=True)
    state = _t_mix(torch, (idx ^ _t_const(seed)) * 3 + G)
    ar = torch.arange(BLOCK, dtype=torch.int64, device=device)
    k8 = (torch.arange(1, BLOCK // 8 + 64 + 1, dtype=torch.int64, device=device)) * G
    for ci, name in enumerate(CLASSES):
        sel_all = torch.nonzero(cls == ci).flatten()
        for c0 in range(0, sel_all.numel(), chunk):
            sel = sel_all[c0:c0 + chunk]
            m = sel.numel()
            if name in ("T_TEXT", "T_HTML"):
                src = torch.from_numpy(_source(name).copy()).to(device)
                off = _t_umod(torch, _t_mix(torch, state[sel] + G), src.numel() - BLOCK)
                out[sel] = src[(off[:, None] + ar[None, :])]
            elif name == "R":
                words = _t_mix(torch, state[sel][:, None] + k8[None, :BLOCK // 8])
                out[sel] = words.contiguous().view(torch.uint8).reshape(m, BLOCK)
            elif name == "RS":
                words = _t_mix(torch, state[sel][:, None] + k8[None, :])
                rnd = words[:, :BLOCK // 8].contiguous().view(torch.uint8).reshape(m, BLOCK)
                ctl = words[:, BLOCK // 8:]                              # [m, 64]
                length = 1000 + _t_umod(torch, ctl, 9001)                # string k
                reps = 2 + _t_umod(torch, _t_lsr(torch, ctl, 32), 3)
                span = length * reps                                     # output bytes of k
                ends = torch.cumsum(span, dim=1)                         # [m, 64]
                used = torch.cumsum(length, dim=1) - length              # source start of k
                k = torch.searchsorted(ends, ar[None, :].expand(m, BLOCK).contiguous(), right=True)
                k = k.clamp(max=63)
                start = torch.gather(ends - span, 1, k)
                within = (ar[None, :] - start) % torch.gather(length, 1, k)
                out[sel] = torch.gather(rnd, 1, torch.gather(used, 1, k) + within)
            elif name == "P10":
                out[sel] = (ar % 10 + ord("a")).to(torch.uint8)
            elif name == "Z":
                out[sel] = 0
            elif name == "RAMP":
                out[sel] = (ar & 0xff).to(torch.uint8)
    return out

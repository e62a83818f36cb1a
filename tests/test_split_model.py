"""The raw-buffer splitter's procedure (csrc/split_kernels.h) restated on the CPU (tools/split_model.py): the chain it
marks is the sequential parse -- on text, incompressible data (literal chains), periods whose copies parse from a
wrong phase as well, repeated strings, and foreign streams with copy4 elements (which it must refuse or get right)."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import split_model as sm  # noqa: E402
from conftest import golden_file  # noqa: E402


def _streams(orc):
    rng = random.Random(5)
    text = golden_file("alice29.txt") + golden_file("html")
    srcs = [text[:200000], rng.randbytes(200000), bytes(150000),
            text[:70000] + rng.randbytes(140000) + text[70000:140000]]
    for period in (10, 14, 18, 254):
        srcs.append((text[5000:5000 + period] * (150000 // period + 1))[:150000])
    srcs.append(b"".join(rng.randbytes(rng.randint(1000, 9000)) * rng.randint(2, 4) for _ in range(20)))
    for src in srcs:
        comp = orc.encode(src)
        hdr = 1
        while comp[hdr - 1] & 0x80:
            hdr += 1
        yield bytes(comp[hdr:])


def test_marked_chain_is_the_sequential_parse(orc):
    for i, s in enumerate(_streams(orc)):
        want = sm.sequential_entries(s)
        got = sm.split(s, with_out=True)
        assert want is not None and got is not None, i  # (complete within the rounds enqueued without a look)
        assert got[0] == want, (i, got[1])
        assert got[2] == sm.sequential_out(s), i  # (what the prefix sum over the marked walks places the segments by)


def test_a_damaged_stream_is_refused_or_right(orc):
    rng = random.Random(6)
    text = golden_file("alice29.txt")
    comp = bytearray(orc.encode(text[:150000]))
    for _ in range(6):
        m = bytearray(comp[3:])
        m[rng.randrange(len(m))] ^= 1 << rng.randrange(8)
        want = sm.sequential_entries(bytes(m))
        got = sm.split(bytes(m))
        if got is not None:  # a chain from byte 0 to the end IS the parse: then it must be the sequential one
            assert want is not None and got[0] == want

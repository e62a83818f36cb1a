"""The framed container END TO END ON THE DEVICE: stream resident in HBM in,
bytes in HBM out, chunk walk / CRC comparison / first-failure verdict on the GPU
(snappy_hip_uncompress_framed_d, snappy_hip_compress_framed_d), against the oracle.

The reference's framed tests (tests/test_framed.nim, restated in behaviour.py) run a second time
through an adapter whose uncompress_framed / encode_framed go through the device entry points with
EXACT-SIZE device buffers (no slack in front of or behind the stream and the output)."""
import numpy as np
import pytest

import behaviour as bh
import cases
import corpus
from conftest import FRAMED_FILES, golden_file

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__
    return __graft_entry__.build()


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


class DeviceFramed:
    """The package's API with the framed calls routed through the device-resident entry points."""

    def __init__(self, hip, torch):
        self._hip, self._torch = hip, torch
        self.ctx = hip.Context(0)

    def __getattr__(self, name):
        return getattr(self._hip, name)

    def _up(self, data):
        t = self._torch
        if len(data) == 0:
            return t.empty(0, dtype=t.uint8, device="cuda")
        return t.frombuffer(bytearray(data), dtype=t.uint8).cuda()

    def uncompress_framed(self, data, cap, check_header=True, check_integrity=True):
        t = self._torch
        d_in = self._up(bytes(data))
        d_out = t.empty(cap, dtype=t.uint8, device="cuda")
        st, rd, wr = self.ctx.uncompress_framed(d_in, len(data), d_out, cap, check_header, check_integrity)
        out = bytes(d_out[:wr].cpu().numpy().tobytes()) if st == bh.OK else b""
        return st, rd, wr, out

    def decode_framed(self, data, max_size=2**63 - 1, check_integrity=True):
        n = self._hip.uncompressed_len_framed(data)
        if n is None or n > max_size:
            return b""
        st, _, _, out = self.uncompress_framed(data, n, check_integrity=check_integrity)
        return out if st == bh.OK else b""

    def encode_framed(self, data):
        t = self._torch
        cap = self._hip.max_compressed_len_framed(len(data))
        d_in = self._up(bytes(data))
        d_out = t.empty(cap, dtype=t.uint8, device="cuda")
        w = self.ctx.compress_framed(d_in, len(data), d_out, cap)
        return bytes(d_out[:w].cpu().numpy().tobytes())


@pytest.fixture(scope="module")
def dev(hip, torch_mod):
    return DeviceFramed(hip, torch_mod)


@pytest.mark.parametrize("name,target", [("alice29.txt.sz-32k", "alice29.txt"),
                                         ("alice29.txt.sz-64k", "alice29.txt"),
                                         ("house.jpg.sz", "house.jpg")])
def test_framed_golden_on_device(dev, name, target):
    bh.check_framed_golden(dev, name, target)  # incl. the partial decode + resume protocol


@pytest.mark.parametrize("name", FRAMED_FILES)
def test_framed_round_trip_on_device(dev, orc, name):
    src = golden_file(name)
    assert bh.check_framed_round_trip(dev, src) == orc.encode_framed(src)


def test_framed_edges_on_device(dev):
    bh.check_framed_edges(dev)


def _chunks(stream):
    pos, out = 10, []
    while pos < len(stream):
        dl = int.from_bytes(stream[pos + 1:pos + 4], "little")
        out.append((pos, stream[pos], dl))
        pos += 4 + dl
    return out


def test_many_chunk_stream_matches_oracle(dev, orc, torch_mod):
    """>= 4096 chunks of the seeded corpus (compressed and stored chunks), as ONE stream: the device
    encoding equals the oracle's byte for byte; then the oracle's verdict and counters for the
    intact stream, a skippable chunk in the middle, a CRC mismatch, a corrupt body, an unknown chunk
    type, a truncated tail and a mid-stream resume"""
    torch = torch_mod
    nb = 4096
    blocks = corpus.make_blocks(0, nb)
    src = blocks.tobytes()
    stream = dev.encode_framed(src)
    want = orc.encode_framed(src)
    assert stream == want
    ch = _chunks(stream)
    assert len(ch) == nb and {c[1] for c in ch} == {0, 1}  # compressed and stored chunks

    def both(data, cap, **kw):
        got = dev.uncompress_framed(data, cap, **kw)
        exp = orc.uncompress_framed(data, cap, **kw)
        assert got[0] == exp[0], (got[:3], exp[:3])
        if exp[0] == bh.OK:
            assert got[:3] == tuple(exp[:3])
            assert got[3] == bytes(exp[3])
        return got

    st, rd, wr, out = both(stream, len(src))
    assert (st, rd, wr) == (bh.OK, len(stream), len(src)) and out == src
    # a skippable chunk (0x80, not validated: snappy.nim:262-263) and a repeated stream identifier
    k = ch[1000][0]
    skip = b"\x80" + bh.le24(5) + b"hello" + cases.FRAMING_HEADER
    st, rd, wr, out = both(stream[:k] + skip + stream[k:], len(src))
    assert (st, wr) == (bh.OK, len(src)) and out == src
    # CRC mismatch in chunk 2500, corrupt body in a later compressed chunk: the first one decides
    bad = bytearray(stream)
    bad[ch[2500][0] + 5] ^= 0x40
    later = next(c for c in ch[2600:] if c[1] == 0)
    bad[later[0] + 20] ^= 0xff
    assert both(bytes(bad), len(src))[0] == bh.CRC_MISMATCH
    bad2 = bytearray(stream)
    comp = next(c for c in ch[100:] if c[1] == 0 and c[2] > 200)
    for i in range(40, 60):
        bad2[comp[0] + i] ^= 0xa5
    assert both(bytes(bad2), len(src))[0] in (bh.INVALID_INPUT, bh.CRC_MISMATCH)
    # an unknown chunk type in the middle, a stream cut inside a chunk
    assert both(stream[:k] + b"\x05" + bh.le24(0) + stream[k:], len(src))[0] == bh.UNKNOWN_CHUNK
    assert both(stream[:ch[3000][0] + 9], len(src))[0] == bh.INVALID_INPUT
    # integrity checks off: the same corrupt CRC passes
    bad3 = bytearray(stream)
    bad3[ch[2500][0] + 5] ^= 0x40
    st, rd, wr, out = both(bytes(bad3), len(src), check_integrity=False)
    assert st == bh.OK and out == src
    # the output fills in the middle of the stream: ok((read, written)), then resume without header
    cap = 65536 * 1500 + 17
    st, rd, wr, out = both(stream, cap)
    assert st == bh.OK and wr == 65536 * 1500 and rd == ch[1500][0] and out == src[:wr]
    st, rd2, wr2, out2 = both(stream[rd:], len(src) - wr, check_header=False)
    assert (st, rd2, wr2) == (bh.OK, len(stream) - rd, len(src) - wr) and out2 == src[wr:]


def test_tiny_chunks_overflow_the_first_list(dev, orc):
    """a stream of very many tiny chunks needs the second, worst-case sizing of the chunk lists"""
    H = cases.FRAMING_HEADER
    one = b"\x01" + bh.le24(5) + orc.masked_crc(b"x").to_bytes(4, "little") + b"x"
    n = 9000
    stream = H + one * n
    st, rd, wr, out = dev.uncompress_framed(stream, n)
    assert (st, rd, wr) == (bh.OK, len(stream), n) and out == b"x" * n


def test_random_framed_streams_match_oracle(dev, orc):
    """streams assembled chunk by chunk from every chunk class (compressed, stored, skippable, padding,
    repeated identifiers, unknown), random truncations, random corruptions and random output
    capacities: status and, when ok, both counters and the bytes equal the oracle's -- through the
    parallel chunk walk where it applies (>= 4 MiB) and the serial one elsewhere"""
    import random
    rng = random.Random(77)
    H = cases.FRAMING_HEADER
    text = golden_file("alice29.txt") + golden_file("html")

    def chunk(kind, payload=b""):
        if kind == 0:
            comp = orc.encode(payload)
            return b"\x00" + bh.le24(len(comp) + 4) + orc.masked_crc(payload).to_bytes(4, "little") + comp
        if kind == 1:
            return b"\x01" + bh.le24(len(payload) + 4) + orc.masked_crc(payload).to_bytes(4, "little") + payload
        return bytes([kind]) + bh.le24(len(payload)) + payload

    for it in range(28):
        big = it % 4 == 0  # a stream long enough for the parallel walk
        parts, plain = [H], bytearray()
        for _ in range(rng.randint(120, 200) if big else rng.randint(1, 40)):
            r = rng.random()
            n = rng.choice([0, 1, 17, 500, 20000, 65536]) if rng.random() < 0.3 else rng.randint(1, 65536)
            o = rng.randrange(len(text) - 65536)
            data = text[o:o + n] if rng.random() < 0.7 else rng.randbytes(n)
            if r < 0.55:
                parts.append(chunk(0, data)); plain += data
            elif r < 0.85:
                parts.append(chunk(1, data)); plain += data
            elif r < 0.93:
                parts.append(chunk(rng.randint(0x80, 0xfe), rng.randbytes(rng.randint(0, 300))))
            elif r < 0.97:
                parts.append(H)
            elif not big:
                parts.append(chunk(rng.randint(2, 0x7f), rng.randbytes(rng.randint(0, 40))))
        stream = b"".join(parts)
        variants = [(stream, len(plain))]
        if len(stream) > 30:
            cut = rng.randrange(10, len(stream))
            variants.append((stream[:cut], len(plain)))
            bad = bytearray(stream)
            bad[rng.randrange(10, len(stream))] ^= 1 << rng.randrange(8)
            variants.append((bytes(bad), len(plain)))
            variants.append((stream, rng.randrange(0, len(plain) + 1)))
        for data, cap in variants:
            got = dev.uncompress_framed(data, cap)
            exp = orc.uncompress_framed(data, cap)
            assert got[0] == exp[0], (it, got[:3], exp[:3])
            if exp[0] == bh.OK:
                assert got[:3] == tuple(exp[:3]) and got[3] == bytes(exp[3]), it


def test_stream_ending_in_a_short_padding_chunk_flush_against_its_allocation(hip, orc):
    """A skippable / padding chunk with fewer than 4 data bytes may END the stream (`fe 00 00 00`): the parallel
    chunk walk (streams >= 4 MiB) must not read a CRC field behind such a header.  The stream is padded so that
    its length is a multiple of 2 MiB and sits in a hipMalloc of exactly its size (no torch allocator)."""
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    rt.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    rt.hipFree.argtypes = [ctypes.c_void_p]
    src = corpus.make_blocks(0, 160).tobytes()  # 10 MiB: text, html, random (stored chunks), runs
    body = orc.encode_framed(src)
    gran = 2 << 20
    fill = (-(len(body) + 4 + 4)) % gran  # a padding chunk of `fill` bytes, then the 4-byte empty one
    stream = body + b"\xfe" + bh.le24(fill) + bytes(fill) + b"\xfe\x00\x00\x00"
    assert len(stream) % gran == 0
    ctx = hip.Context(0)
    for tail in (stream, stream[:-4] + b"\x80\x03\x00\x00abc"):
        # second variant: a skippable chunk with 3 data bytes ends the stream (length no longer a multiple)
        p_in, p_out = ctypes.c_void_p(), ctypes.c_void_p()
        assert rt.hipMalloc(ctypes.byref(p_in), len(tail)) == 0 and rt.hipMalloc(ctypes.byref(p_out), len(src)) == 0
        assert rt.hipMemcpy(p_in, tail, len(tail), 1) == 0
        got = ctx.uncompress_framed(p_in.value, len(tail), p_out.value, len(src))
        exp = orc.uncompress_framed(tail, len(src))
        assert got == (exp[0], exp[1], exp[2]) == (bh.OK, len(tail), len(src))
        back = ctypes.create_string_buffer(len(src))
        assert rt.hipMemcpy(back, p_out, len(src), 2) == 0 and back.raw == src
        rt.hipFree(p_in)
        rt.hipFree(p_out)
    ctx.close()

"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the golden
fixtures.  Bar: bit-exact -- this is byte / integer work.  Needs a real MI355X."""
import hashlib
import random

import numpy as np
import pytest

import behaviour as bh
import cases
from conftest import DATA_FILES, FRAMED_FILES, golden_file

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__
    return __graft_entry__.build()


# ---- the reference's own test-suite, run against the HIP path ----------------------------------
def test_reference_block_suite(hip, orc, vectors):
    bh.check_empty(hip)
    bh.check_handwritten(hip)
    bh.check_compresses(hip)
    bh.check_bad_data(hip)
    bh.check_baddata_files(hip, vectors)
    bh.check_literal61_quirk(hip)
    bh.check_caller_buffer_rules(hip, orc)
    bh.check_golden_rawsnappy(hip, vectors)
    bh.check_encoder_reproduces_golden_rawsnappy(hip)
    bh.check_lone_copy_units(hip)
    for w in cases.WITNESSES:
        assert bh.check_round_trip(hip, w) == orc.encode(w)


@pytest.mark.parametrize("name", DATA_FILES)
def test_data_files_bit_exact(hip, orc, vectors, name):
    src = golden_file(name)
    enc = bh.check_round_trip(hip, src)
    assert enc == orc.encode(src)  # byte-identical to the reference encoder's restatement
    assert (len(enc), bh.sha(enc)) == (vectors["files"][name]["oracle_nim_len"],
                                       vectors["files"][name]["oracle_nim_sha256"])
    # foreign (C++-semantics) stream of the same data decodes identically too
    assert hip.decode(orc.encode(src, flags=3)) == src


def test_synthetic_patterns_bit_exact(hip, orc):
    """tests/test_snappy.nim:110-134"""
    for y in cases.repeat_cases():
        assert bh.check_round_trip(hip, y) == orc.encode(y)
    i = 1
    while i < 20000:
        buf = cases.mod10(i)
        assert bh.check_round_trip(hip, buf) == orc.encode(buf)
        i += 23 * 7  # every 7th size of the reference's sweep keeps the run short
    for n in cases.block_boundary_lengths():
        if n % 65536 in (65531, 0, 5):  # edges of each boundary window
            for buf in (bytes(n), cases.mod10(n)):
                assert bh.check_round_trip(hip, buf) == orc.encode(buf)


def test_every_small_length(hip, orc):
    """lengths around minNonLiteralBlockSize and the table-size steps (encoder.nim:27-34)"""
    text = golden_file("alice29.txt")
    for n in list(range(1, 40)) + [255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 4095, 4096,
                                   4097, 16383, 16384, 16385, 32768]:
        src = text[1000:1000 + n]
        assert hip.encode(src) == orc.encode(src), n
        assert hip.encode_block(src) == orc.encode_block(src), n
        assert hip.encode_frame(src) == orc.encode_frame(src), n


def test_match_runs_into_block_end(hip, orc):
    """findMatchLength is bounded by the block's end (encoder.nim:130-182), also when the copy-loop
    probe sits exactly at ip = n - 15 (encoder.nim:362 lets it) and 15 bytes match to the end"""
    rng = random.Random(7)
    head = bytes(rng.randrange(256) for _ in range(300))
    for tail in range(12, 40):
        for gap in (0, 1, 3, 17):
            # ... X[0:tail+20] ... filler ... X again, cut so that the block ends `tail` bytes into
            # the second match region after an earlier copy
            x = bytes(rng.randrange(256) for _ in range(tail + 24))
            src = head + x + bytes(rng.randrange(256) for _ in range(50 + gap)) + x[:8] + b"#" + x[:tail]
            assert hip.encode_block(src) == orc.encode_block(src), (tail, gap)
            src2 = head + x + x[:20] + bytes([rng.randrange(256)]) * gap + x[:tail]
            assert hip.encode_block(src2) == orc.encode_block(src2), (tail, gap)


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _literal(data):
    n = len(data) - 1
    if n < 60:
        return bytes([n << 2]) + data
    ll = (n.bit_length() + 7) // 8
    return bytes([(59 + ll) << 2]) + n.to_bytes(ll, "little") + data


def test_raw_multi_block_streams(hip, orc):
    """uncompress() of raw buffers that decode to several 64 KiB blocks: streams of a 64 KiB-block
    encoder are split and decoded in parallel; foreign streams whose elements or copies cross a
    64 KiB output boundary must give the same bytes through the serial path (decoder.nim:20-155
    works on the whole buffer, snappy.nim:92-110)"""
    rng = random.Random(11)
    text = golden_file("alice29.txt") + golden_file("html")
    # 1. from the encoder itself (elements never cross a block boundary)
    for n in (65537, 131072, 200000, len(text)):
        comp = orc.encode(text[:n])
        assert hip.decode(comp) == text[:n]
    # 2. a literal that straddles the boundary
    a = bytes(rng.randrange(256) for _ in range(70000))
    b = bytes(rng.randrange(256) for _ in range(3000))
    s2 = _varint(len(a) + len(b) + 64) + _literal(a) + _literal(b) + bytes([(63 << 2) | 2, 0x10, 0x00])
    assert hip.decode(s2) == orc.decode(s2) != b""
    # 3. block-aligned elements, but a copy in the second block that reads from the first one
    c = bytes(rng.randrange(256) for _ in range(65536))
    s3 = _varint(65536 + 64 + 100) + _literal(c) + bytes([(63 << 2) | 2, 0xff, 0xff]) + _literal(b[:100])
    assert hip.decode(s3) == orc.decode(s3) != b""
    # 4. the same kinds, damaged: identical verdicts
    for bad in (orc.encode(text[:200000])[:-3], s2[:-1], s3[:40000], s3[:-101] + bytes([0xfc])):
        assert hip.uncompress(bad, 400000)[0] == orc.uncompress(bad, 400000)[0] != bh.OK


def test_input_longer_than_the_format_allows(hip):
    """snappy.nim:41-42: len > 2^32 - 1 is invalidInput -- decided before the input is touched"""
    import ctypes
    w = ctypes.c_size_t(77)
    out = ctypes.create_string_buffer(64)
    for n in (1 << 32, (1 << 32) + 5, 1 << 40):
        assert hip.lib.snappy_hip_compress(None, n, out, 64, ctypes.byref(w)) == bh.INVALID_INPUT
    assert hip.lib.snappy_hip_max_compressed_len(0xFFFFFFFF) == 32 + 0xFFFFFFFF + 0xFFFFFFFF // 6


def test_copy_offset_65535(hip, orc):
    """65 535 is a legal copy offset for exactly one element: a copy at output position 65 535 of a full block
    (decoder.nim:112 wants offset <= op).  The indexed decoder's element list keeps 16-bit offsets: such a unit must
    still give the reference's bytes -- many short elements (the ring decoder's units), a short stream that is no
    period, copy2 and copy4 forms, and the same element one position early (offset > op: invalidInput)."""
    rng = random.Random(65535)
    # (a) 1 093 literals of 60 bytes, then the copy
    plain = rng.randbytes(65535)
    lits = b"".join(_literal(plain[i:i + 60]) for i in range(0, 65535, 60))
    # (b) one literal and a stretch of copies with changing offsets: stream < 4 KiB, > 832 elements, not a period
    out_b = bytearray(rng.randbytes(100))
    body_b = bytearray(_literal(bytes(out_b)))
    k = 0
    while len(out_b) < 65535:
        ln = min(64, 65535 - len(out_b))
        off = 100 - (k % 7)
        body_b += bytes([((ln - 1) << 2) | 2]) + off.to_bytes(2, "little")
        for _ in range(ln):
            out_b.append(out_b[-off])
        k += 1
    for body, pl in ((lits, plain), (bytes(body_b), bytes(out_b))):
        for tail in (bytes([0x02, 0xff, 0xff]), bytes([0x03, 0xff, 0xff, 0x00, 0x00])):
            want = pl + pl[:1]
            assert orc.decode_all_tags(body + tail, 65536) == (0, want)
            assert hip.decode_all_tags(body + tail, 65536) == (0, want)
            assert hip.decode(_varint(65536) + body + tail) == want
        # one byte less in front of it: op = 65 534 < offset
        short = _literal(pl[:59]) + body[61:] if body is lits else None
        if short is not None:
            bad = short + bytes([0x02, 0xff, 0xff])
            assert orc.decode_all_tags(bad, 65536)[0] == bh.INVALID_INPUT
            assert hip.decode_all_tags(bad, 65536)[0] == bh.INVALID_INPUT


def test_random_strings(hip, orc):
    """tests/test_snappy.nim:247-253"""
    for s in bh.random_strings(0x5EED, count=40):
        assert bh.check_round_trip(hip, s) == orc.encode(s)


def test_block_level_entry_points(hip, orc):
    src = golden_file("html")[:65536]
    body = hip.encode_block(src)
    assert body == orc.encode_block(src)
    st, out = hip.decode_all_tags(body, 65536)
    assert st == 0 and out == src
    st, out = hip.decode_all_tags(body, 100000)  # larger window: written < cap is fine
    assert st == 0 and out == src
    st, _ = hip.decode_all_tags(body, 65535)
    assert st == bh.INVALID_INPUT  # decoder.nim:77-79 / :127-128: not bufferTooSmall
    assert hip.decode_all_tags(b"", 0) == (0, b"")
    assert hip.decode_all_tags(b"\x00a", 0)[0] == bh.BUFFER_TOO_SMALL
    assert hip.encode_frame(src) == orc.encode_frame(src)


def test_crc_parity(hip, orc, vectors):
    pat = bytes((i * 131 + 7) & 0xff for i in range(70000))
    for k in vectors["crc_kats"]:
        if "pattern" in k:
            data = pat[:k["len"]]
        elif "hex" in k:
            data = bytes.fromhex(k["hex"])
        else:
            data = bytes(k["zeros"])
        assert hip.masked_crc(data) == k["masked"], k
    rng = random.Random(5)
    for n in [1021, 1023, 1024, 1025, 1027, 2047, 2048, 2049, 200000, 1 << 20]:
        data = rng.randbytes(n)
        assert hip.masked_crc(data) == orc.masked_crc(data), n


# ---- framed --------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,target", [("alice29.txt.sz-32k", "alice29.txt"),
                                         ("alice29.txt.sz-64k", "alice29.txt"),
                                         ("house.jpg.sz", "house.jpg")])
def test_framed_golden(hip, name, target):
    bh.check_framed_golden(hip, name, target)


@pytest.mark.parametrize("name", FRAMED_FILES)
def test_framed_round_trip_bit_exact(hip, orc, name):
    src = golden_file(name)
    assert bh.check_framed_round_trip(hip, src) == orc.encode_framed(src)


def test_framed_edges(hip):
    bh.check_framed_edges(hip)


def test_framed_error_order(hip, orc):
    """first failing chunk in stream order decides, exactly as the serial loop"""
    a, b = cases.mod10(70000), golden_file("html")
    good = orc.encode_framed(a + b)
    H = cases.FRAMING_HEADER
    # corrupt the CRC of the 2nd chunk and the body of the 3rd: crcMismatch wins
    pos = 10
    chunks = []
    while pos < len(good):
        dl = int.from_bytes(good[pos + 1:pos + 4], "little")
        chunks.append((pos, dl))
        pos += 4 + dl
    assert len(chunks) >= 3
    bad = bytearray(good)
    bad[chunks[1][0] + 4] ^= 0xff
    bad[chunks[2][0] + 12] ^= 0xff
    for data in (bytes(bad), good[:chunks[2][0] + 9], good + b"\x02\x00\x00\x00"):
        want = orc.uncompress_framed(data, len(a) + len(b))
        got = hip.uncompress_framed(data, len(a) + len(b))
        assert got[0] == want[0]
        if want[0] == 0:
            assert got == want
    assert hip.uncompress_framed(bytes(bad), len(a) + len(b))[0] == bh.CRC_MISMATCH


# ---- malformed input: same verdict as the oracle on mutated streams --------------------------------
def test_mutation_fuzz_matches_oracle(hip, orc):
    rng = random.Random(1234)
    seeds = [orc.encode(golden_file("html")[:20000]), orc.encode(cases.mod10(3000)),
             orc.encode(golden_file("alice29.txt")[:30000]), orc.encode(rng.randbytes(500))]
    for base in seeds:
        for _ in range(40):
            m = bytearray(base)
            for _ in range(rng.randint(1, 3)):
                m[rng.randrange(len(m))] = rng.randrange(256)
            if rng.random() < 0.2:
                m = m[:rng.randrange(1, len(m))]
            m = bytes(m)
            n = orc.uncompressed_len(m)
            if n is None or n > 1 << 20:
                assert hip.decode(m) == b""
                continue
            assert hip.uncompress(m, n) == orc.uncompress(m, n)


def _random_stream(rng, target, far=False):
    """A valid tag stream built element by element: every tag form the decoder knows
    (decoder.nim:42-109), including the ones the reference's encoder never emits -- literals
    with 3 and 4 length bytes, copy4, copy2 of length 1..3, self-overlapping copies."""
    out = bytearray()
    body = bytearray()
    while len(out) < target:
        kind = rng.random()
        if kind < 0.3 or len(out) == 0:
            n = rng.choice([1, 2, 3, 5, 17, 59, 60, 61, 70, 255, 256, 257, 300, 2000])
            data = rng.randbytes(n)
            form = rng.random()
            m = n - 1
            if m < 60:
                body += bytes([m << 2])
            else:  # (length bytes need >= 61 stream bytes after the tag, decoder.nim:54-57: n >= 61 here)
                ll = max((m.bit_length() + 7) // 8, 1)
                ll = min(4, ll + (1 if form > 0.8 else 0))  # non-minimal length bytes too
                body += bytes([(59 + ll) << 2]) + m.to_bytes(ll, "little")
            body += data
            out += data
        else:
            off = rng.randint(1, min(len(out), 65535 if not far else len(out)))
            if rng.random() < 0.3:
                off = rng.randint(1, min(len(out), 12))  # short offsets: overlapping copies
            form = rng.random()
            if form < 0.4 and off < 2048:
                ln = rng.randint(4, 11)
                body += bytes([((off >> 8) << 5) | ((ln - 4) << 2) | 1, off & 0xff])
            elif form < 0.9 and off < 65536:
                ln = rng.randint(1, 64)
                body += bytes([((ln - 1) << 2) | 2]) + off.to_bytes(2, "little")
            else:
                ln = rng.randint(1, 64)
                body += bytes([((ln - 1) << 2) | 3]) + off.to_bytes(4, "little")
            for _ in range(ln):
                out.append(out[-off])
    return bytes(body), bytes(out)


def test_random_valid_streams_match_oracle(hip, orc):
    """streams of a foreign encoder: all element forms, through the block path (<= 64 KiB) and the
    raw multi-block path"""
    rng = random.Random(2024)
    for i in range(60):
        target = rng.choice([50, 700, 5000, 30000, 65000])
        body, plain = _random_stream(rng, target)
        if len(plain) > 65536:
            continue
        assert orc.decode_all_tags(body, len(plain)) == (0, plain)
        assert hip.decode_all_tags(body, len(plain)) == (0, plain), i
        raw = _varint(len(plain)) + body
        assert hip.decode(raw) == plain, i
    for i in range(6):  # several blocks, elements placed without regard to 64 KiB boundaries
        body, plain = _random_stream(rng, rng.choice([70000, 150000, 300000]), far=True)
        raw = _varint(len(plain)) + body
        assert orc.decode(raw) == plain
        assert hip.decode(raw) == plain, i


def test_long_same_offset_runs_inside_a_block(hip, orc):
    """long matches (runs of same-offset copies, encoder.nim:97-112) that start in the middle of a
    block, right after unrelated data: the decoder's run extension may only reach back into the
    run itself (found by examples/roundtrip.c)"""
    base = b"the quick brown fox jumps over the lazy dog "
    for period, seg in ((44, 7000), (10, 3000), (1, 5000), (200, 9000), (255, 4000), (257, 4000), (300, 6000)):
        unit = (base * 8)[:period]
        src = bytes((unit[i % period] + (i // seg)) & 255 for i in range(200000))
        comp = orc.encode(src)
        assert hip.encode(src) == comp
        assert hip.decode(comp) == src, (period, seg)
        body = orc.encode_block(src[:65536])
        assert hip.decode_all_tags(body, 65536) == (0, src[:65536]), (period, seg)


def test_copies_longer_than_a_round_at_every_lane(hip, orc):
    """a match of 65..1100 bytes that starts at any of a round's 64 positions: the encoder emits the
    64- / 60-byte elements of emitCopy (encoder.nim:97-125) from the idle lanes behind the match, or
    -- more elements than lanes left -- through its slow drain; either way the oracle's bytes"""
    rng = random.Random(65)
    text = golden_file("alice29.txt")
    for trial in range(24):
        parts = [text[:3000 + trial]]
        for k in range(40):
            L = rng.choice((65, 66, 67, 68, 69, 100, 127, 128, 129, 131, 190, 192, 196, 260, 700, 1100))
            at = rng.randrange(0, len(parts[0]) - L)
            parts.append(parts[0][at:at + L])                       # a repeat: one long match
            parts.append(rng.randbytes(rng.randrange(1, 70)))         # literals: shifts the next match's lane
        src = b"".join(parts)[:65536]
        assert hip.encode(src) == orc.encode(src), trial


def test_stream_adapters(hip):
    """the batching stream front-ends (snappy/faststreams.nim, snappy/streams.nim restated in
    nim-snappy_amd/streams.py) over the HIP codec: same checks as with the oracle backend"""
    import test_streams
    test_streams.check_adapters(hip, 2)
    test_streams.check_adapters(hip, 256)
    test_streams.check_adapter_errors(hip, 2)
    test_streams.check_adapter_errors(hip, 256)


def test_long_foreign_block_streams_take_the_one_pass_kernel(hip, orc):
    """a block whose tag stream is longer than the indexed decoder takes (> 80 KiB for <= 64 KiB of
    output: only a foreign encoder writes that) is handed to the one-pass kernel: same result"""
    rng = random.Random(3)
    plain = rng.randbytes(50000)
    body = b"".join(bytes([0x00, b]) for b in plain)  # 1-byte literals: 2 stream bytes per byte
    assert len(body) > 81920
    assert orc.decode_all_tags(body, len(plain)) == (0, plain)
    assert hip.decode_all_tags(body, len(plain)) == (0, plain)
    raw = _varint(len(plain)) + body
    assert hip.decode(raw) == plain
    assert hip.uncompress(raw[:-1], len(plain))[0] == orc.uncompress(raw[:-1], len(plain))[0] != bh.OK


def test_host_calls_from_several_threads(hip, orc):
    """the reference's API is re-entrant (all `func`, no globals): concurrent host-buffer calls run on
    contexts of their own (a pool) and give the oracle's bytes; large inputs go in several batches on
    worker threads (more than 2048 blocks) and must come out in stream order"""
    import threading
    srcs = [golden_file("alice29.txt") * 3, golden_file("html") * 5, cases.mod10(300000),
            golden_file("urls.10K"), golden_file("kppkn.gtb") * 2, bytes(range(256)) * 2000]
    got = [None] * len(srcs)

    def work(i):
        for _ in range(3):
            enc = hip.encode(srcs[i])
            fr = hip.encode_framed(srcs[i])
            assert hip.decode(enc) == srcs[i] and hip.decode_framed(fr) == srcs[i]
            got[i] = (enc, fr)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(srcs))]
    [t.start() for t in th]
    [t.join() for t in th]
    for i, s in enumerate(srcs):
        assert got[i] == (orc.encode(s), orc.encode_framed(s)), i
    big = (golden_file("alice29.txt") + golden_file("html") + golden_file("fireworks.jpeg")) * 400  # ~150 MiB, 3 batches
    assert hip.encode_framed(big) == orc.encode_framed(big)
    assert hip.decode_framed(hip.encode_framed(big)) == big
    assert hip.encode(big) == orc.encode(big)


def test_random_multi_block_raw_buffers(hip, orc):
    """raw buffers of several blocks through the speculative split (csrc/split_kernels.h) and its
    fallbacks: buffers of the block encoder (text, mixed with incompressible and repetitive blocks),
    the same with one byte flipped or the tail cut (verdict equality), and foreign streams whose
    elements ignore the block boundaries -- always the oracle's bytes or the oracle's failure"""
    rng = random.Random(99)
    text = golden_file("alice29.txt") + golden_file("html") + golden_file("urls.10K")
    srcs = [text[:300000], text[:131072] + rng.randbytes(70000) + text[:200000] + bytes(90000),
            rng.randbytes(200000), (text[1000:9000] * 40)[:400000],
            b"".join(rng.randbytes(rng.randint(1000, 9000)) * rng.randint(2, 4) for _ in range(30))]
    for i, src in enumerate(srcs):
        comp = orc.encode(src)
        assert hip.decode(comp) == src, i
        for _ in range(4):
            bad = bytearray(comp)
            bad[rng.randrange(5, len(bad))] ^= 1 << rng.randrange(8)
            assert hip.decode(bytes(bad)) == orc.decode(bytes(bad)), i
        cut = comp[:rng.randrange(len(comp) // 2, len(comp))]
        assert hip.decode(cut) == orc.decode(cut) == b"", i
    for i in range(5):
        body, plain = _random_stream(rng, rng.choice([200000, 500000]), far=True)
        raw = _varint(len(plain)) + body
        assert hip.decode(raw) == orc.decode(raw) == plain, i
    # long literals back to back (every incompressible block is one): the walk that lands on one follows
    # the chain (split_kernels.h); alone, in stretches between text, and with a bit flipped in a length byte
    big = [rng.randbytes(6 << 20),
           b"".join(text[:rng.randint(1, 200000)] + rng.randbytes(rng.randint(1, 400000)) for _ in range(12)),
           rng.randbytes(65536 * 3 + 17) + text[:70000] + rng.randbytes(65536 * 20)]
    # periods: the copies of a period are an element stream of period 3 ("fe 0a 00" ...) that parses as a
    # chain of copy2 elements from its second byte too -- two chains side by side that never fall into step
    # (periods 10, 14, 18: the offset's low byte is a copy2 tag); candidates + marking must pick the real one
    for period in (10, 14, 18, 254, 300):
        big.append((text[5000:5000 + period] * (400000 // period + 1))[:400000])
    big.append(b"".join((text[k * 100:k * 100 + 14] * 6000) + text[:30000] for k in range(6)))
    for i, src in enumerate(big):
        comp = orc.encode(src)
        assert hip.decode(comp) == src, i
        at = comp.find(b"\xf4\xff\xff", len(comp) // 2)
        if at > 0:
            bad = bytearray(comp)
            bad[at + 2] ^= 0x10  # the literal is 4096 bytes shorter than the encoder said
            assert hip.decode(bytes(bad)) == orc.decode(bytes(bad)), i

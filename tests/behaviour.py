"""Backend-independent restatement of the reference's test-suite.

Every check takes a backend `be` exposing the reference's in-memory API names
(encode/decode/compress/uncompress/encode_framed/decode_framed/uncompress_framed/
uncompressed_len_framed/masked_crc/...).  The CPU oracle (oracle/pyoracle.py) and the HIP
product (nim-snappy_amd) both provide it, so the same cases pin the oracle against the
reference's golden vectors (-m "not gpu") and the HIP path against both (-m gpu).
"""
import hashlib
import random

import cases
from conftest import golden_file

OK, BUFFER_TOO_SMALL, INVALID_INPUT, CRC_MISMATCH, UNKNOWN_CHUNK = range(5)


def sha(b):
    return hashlib.sha256(b).hexdigest()


# ---- tests/test_snappy.nim ---------------------------------------------------------------
def check_round_trip(be, src):
    """tests/test_snappy.nim:44-69 (the Nim-decoder half; the C++ half lives in the vectors)."""
    enc = be.encode(src)
    assert len(enc) >= 1
    assert be.decode(enc) == src
    return enc


def check_empty(be):
    """tests/test_snappy.nim:163-165"""
    assert be.encode(b"") == b"\x00"
    assert be.decode(b"\x00") == b""


def check_handwritten(be):
    """tests/test_snappy.nim:136-148"""
    for enc, plain in cases.HANDWRITTEN:
        assert be.decode(enc) == plain


def check_bad_data(be):
    """tests/test_snappy.nim:156-218: decode(...) must return empty"""
    for bad in cases.BAD_DATA:
        assert be.decode(bad) == b"", bad
    # error class through the caller-buffer API (a6/a8 of SURVEY.md 8a)
    st, _ = be.uncompress(b"\x05\x00a", 5)
    assert st == INVALID_INPUT
    st, _ = be.uncompress(b"\x05\x00a", 4)
    assert st == BUFFER_TOO_SMALL
    st, _ = be.uncompress(b"", 10)
    assert st == INVALID_INPUT


def check_lone_copy_units(be):
    """a unit that is one copy and nothing else: invalidInput (decoder.nim:112), as raw buffer and as tag stream"""
    for raw in cases.LONE_COPY_UNITS:
        assert be.decode(raw) == b"", raw
        st, out = be.uncompress(raw, raw[0])
        assert st == INVALID_INPUT, raw
        for cap in (raw[0], 64, 65536):
            st, out = be.decode_all_tags(raw[1:], cap)
            assert st == INVALID_INPUT, (raw, cap)


def check_encoder_reproduces_golden_rawsnappy(be):
    """The one file of encoder output the reference holds (tests/test_snappy.nim:71-83 decodes it): re-encoding its
    plaintext gives the file back byte for byte (9 871 bytes; a 14 KB block, where encoder.nim:36-37's fixed shift and
    libsnappy's size-dependent one pick the same slots)."""
    raw = golden_file("Mark.Twain-Tom.Sawyer.txt.rawsnappy")
    assert be.encode(be.decode(raw)) == raw


def check_compresses(be):
    """tests/test_snappy.nim:150-154"""
    assert len(be.encode(bytes(1024))) < 512


def check_golden_rawsnappy(be, vectors):
    """tests/test_snappy.nim:71-83,108: Nim decode == C++ decode of the Go-encoded golden."""
    raw = golden_file("Mark.Twain-Tom.Sawyer.txt.rawsnappy")
    dec = be.decode(raw)
    assert len(dec) == vectors["rawsnappy"]["decoded_len"]
    assert sha(dec) == vectors["rawsnappy"]["decoded_sha256"]


def check_baddata_files(be, vectors):
    """baddata{1,2,3}.snappy: referenced by no reference test; libsnappy 1.1.8 rejects them."""
    for name, v in vectors["baddata"].items():
        assert not v["libsnappy_accepts"]
        assert be.decode(golden_file(name)) == b"", name


def check_literal61_quirk(be):
    """decoder.nim:54-57: an extended-length literal needs >= 61 bytes after the tag even when
    the literal itself is short -- stricter than the format (libsnappy accepts this stream)."""
    stream = b"\x05" + b"\xf0\x04" + b"abcde"      # literal, 1 length byte (=5), 5 bytes
    assert be.decode(stream) == b""
    padded = b"\x42" + b"\xf0\x04" + b"abcde" + b"\xf0\x3c" + bytes(range(61))
    assert be.decode(padded) == b"abcde" + bytes(range(61))


def check_caller_buffer_rules(be, orc):
    """snappy.nim:41-45,96-97"""
    src = cases.mod10(1000)
    cap = orc.max_compressed_len(len(src))
    st, out = be.compress(src, cap - 1)
    assert st == BUFFER_TOO_SMALL  # even though the data would fit
    st, out = be.compress(src, cap)
    assert st == OK
    st, dec = be.uncompress(out, len(src) - 1)
    assert st == BUFFER_TOO_SMALL
    st, dec = be.uncompress(out, len(src) + 100)
    assert st == OK and dec == src
    st, _ = be.uncompress(out + b"\x00", len(src))  # trailing garbage: one more literal
    assert st == INVALID_INPUT
    st, _ = be.uncompress(out[:-1], len(src))
    assert st == INVALID_INPUT
    assert be.decode(out, max_size=len(src) - 1) == b""  # snappy.nim:121


def random_strings(seed, count=100, lo=1000, hi=10000):
    """tests/test_snappy.nim:247-253 distribution (randgen.nim:21-24), seeded here."""
    rng = random.Random(seed)
    return [rng.randbytes(rng.randint(lo, hi)) for _ in range(count)]


# ---- tests/test_framed.nim ---------------------------------------------------------------
def check_framed_golden(be, name, target):
    """tests/test_framed.nim:9-59"""
    src = golden_file(name)
    expected = golden_file(target)
    assert be.decode_framed(src) == expected
    assert be.uncompressed_len_framed(src) == len(expected)
    st, rd, wr, out = be.uncompress_framed(src, len(expected))
    assert (st, rd, wr) == (OK, len(src), len(expected)) and out == expected
    # partial read into a buffer one byte short, then resume without header check
    st, rd, wr, out = be.uncompress_framed(src, len(expected) - 1)
    assert st == OK and rd < len(src) and wr < len(expected)
    assert out == expected[:wr]
    st, rd2, wr2, out2 = be.uncompress_framed(src[rd:], len(expected) - wr, check_header=False)
    assert (st, rd2, wr2) == (OK, len(src) - rd, len(expected) - wr)
    assert out2 == expected[wr:]


def check_framed_round_trip(be, data):
    """tests/test_framed.nim:61-81"""
    enc = be.encode_framed(data)
    assert be.uncompressed_len_framed(enc) == len(data)
    assert be.decode_framed(enc) == data
    return enc


def check_valid_framed(be, payload, expected, check_integrity=True):
    """tests/test_framed.nim:95-108"""
    assert be.decode_framed(payload, check_integrity=check_integrity) == expected
    st, rd, wr, out = be.uncompress_framed(payload, len(expected), check_integrity=check_integrity)
    assert (st, rd, wr) == (OK, len(payload), len(expected)) and out == expected
    assert be.uncompressed_len_framed(payload) == len(expected)


def check_invalid_framed(be, payload, ulen):
    """tests/test_framed.nim:83-93"""
    st, _, _, _ = be.uncompress_framed(payload, ulen)
    assert st != OK
    assert be.decode_framed(payload) == b""
    assert be.uncompressed_len_framed(payload) is None


def le24(n):
    return n.to_bytes(4, "little")[:3]


def check_framed_edges(be):
    """tests/test_framed.nim:137-219"""
    H = cases.FRAMING_HEADER
    check_valid_framed(be, H, b"")  # just a header
    buf = cases.ramp(128 * 1024)
    for n in cases.FRAMED_SIZES:
        assert be.decode_framed(be.encode_framed(buf[:n])) == buf[:n]
    # full uncompressed / compressed chunks
    data = bytes(65536)
    comp = be.encode(data)
    crc = be.masked_crc(data).to_bytes(4, "little")
    framed = H + b"\x01" + le24(len(data) + 4) + crc + data
    framed_c = H + b"\x00" + le24(len(comp) + 4) + crc + comp
    check_valid_framed(be, framed, data)
    check_valid_framed(be, framed_c, data)
    # checkIntegrity = false with a zero CRC
    z = bytes(4)
    check_valid_framed(be, H + b"\x01" + le24(len(data) + 4) + z + data, data, False)
    check_valid_framed(be, H + b"\x00" + le24(len(comp) + 4) + z + comp, data, False)
    st, _, _, _ = be.uncompress_framed(H + b"\x01" + le24(len(data) + 4) + z + data, len(data))
    assert st == CRC_MISMATCH
    st, _, _, _ = be.uncompress_framed(H + b"\x00" + le24(len(comp) + 4) + z + comp, len(data))
    assert st == CRC_MISMATCH
    # invalid header
    check_invalid_framed(be, bytes([3, 2, 1, 0]), 0)
    check_invalid_framed(be, bytes([0, 0, 0, 0, 42]), 0)
    # overlong frames (65537 bytes) for both chunk kinds
    data = bytes(65537)
    comp = be.encode(data)
    crc = be.masked_crc(data).to_bytes(4, "little")
    check_invalid_framed(be, H + b"\x01" + le24(len(data) + 4) + crc + data, len(data))
    check_invalid_framed(be, H + b"\x00" + le24(len(comp) + 4) + crc + comp, len(data))
    # chunk classes (snappy.nim:259-263): 0x02..0x7f unknown, 0x80..0xff skipped unverified
    st, _, _, _ = be.uncompress_framed(H + b"\x02" + le24(0), 0)
    assert st == UNKNOWN_CHUNK
    st, rd, wr, _ = be.uncompress_framed(H + b"\x80" + le24(3) + b"xyz" + H, 0)
    assert (st, rd, wr) == (OK, 10 + 7 + 10, 0)

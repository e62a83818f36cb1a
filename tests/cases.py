"""Inputs and expected outputs restated from the reference's own tests (data, not code).

tests/test_snappy.nim:136-148  hand-written streams (decoder KATs)
tests/test_snappy.nim:168-218  must-fail strings (decode(...) == empty)
tests/test_snappy.nim:221-245  quick-check witnesses
tests/test_framed.nim:137-219  framed edge cases
"""

HANDWRITTEN = [
    (bytes([27, 0b00001000, 1, 2, 3, 0b00000010, 3, 0, 0b01011000] + list(range(4, 27))),
     bytes([1, 2, 3, 1] + list(range(4, 27)))),
    (bytes([28, 0b00001000, 1, 2, 3, 0b00000010, 3, 0, 0b01011100] + list(range(4, 28))),
     bytes([1, 2, 3, 1] + list(range(4, 28)))),
]

BAD_DATA = [
    b"\x05\x00a",                                      # fewer bytes than the header reports
    b"\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\x00",  # varint overflows u64
    b"\x80\x80\x80\x80\x10",                          # varint fits u64, overflows u32
    b"\x02\x00hi",                                     # literal too small -> dangling copy1
    b"\x02\xechi",                                     # literal length too big
    b"\x02\xf0hi",                                     # 1 extra length byte, src too short
    b"\x02\xf0hi\x00\x00\x00",                        # ... src too short for the literal
    b"\x02\x00a\x01",                                  # copy1 stops at the tag
    b"\x11\x00a\x3e",                                  # copy2 stops at the tag
    b"\x11\x00a\x3e\x01",                              # copy2 stops inside the offset
    b"\x11\x00a\x3f",                                  # copy4 ...
    b"\x11\x00a\x3f\x00",
    b"\x11\x00a\x3f\x00\x00",
    b"\x11\x00a\x3f\x00\x00\x00",
    b"\x11\x00a\x01\x00",                              # offset zero
    b"\x11\x00a\x01\xff",                              # offset too big
    b"\x05\x00a\x1d\x01",                              # length too big
    b"\x11\x00\x00\xfc\xfe\xff\xff\xff",              # 4-byte literal length, huge
    b"\x11\x00\x00\xfc\xff\xff\xff\xff",              # 4-byte literal length wraps to 0
]

# Not from the reference's list -- units whose ONLY element is a copy (decoder.nim:112: op <= offset - 1 with op = 0 is
# invalidInput).  The decoder's "one element writes every byte" shortcut must not take them for a literal.
LONE_COPY_UNITS = [
    b"\x04\x01\x01",                  # copy1, length 4, offset 1
    b"\x0b\x1d\x01",                  # copy1, length 11, offset 1
    b"\x40\xfe\x01\x00",              # copy2, length 64, offset 1
    b"\x01\x02\x01\x00",              # copy2, length 1
    b"\x04\x0f\x01\x00\x00\x00",      # copy4, length 4, offset 1
    b"\x40\xff\x40\x00\x00\x00",      # copy4, length 64, offset 64
]

RANDOM1 = bytes([
    0, 0, 0, 0, 1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 4, 0, 0, 0, 5, 0, 0, 1, 1,
    0, 0, 1, 2, 0, 0, 2, 1, 0, 0, 2, 2, 0, 0, 0, 6, 0, 0, 3, 1, 0, 0, 0, 7, 0,
    0, 1, 3, 0, 0, 0, 8, 0, 0, 2, 3, 0, 0, 0, 9, 0, 0, 1, 4, 0, 0, 1, 0, 0, 3,
    0, 0, 1, 0, 1, 0, 0, 0, 10, 0, 0, 0, 0, 2, 4, 0, 0, 2, 0, 0, 3, 0, 1, 0, 0,
    1, 5, 0, 0, 6, 0, 0, 0, 0, 11, 0, 0, 1, 6, 0, 0, 1, 7, 0, 0, 0, 12, 0, 0,
    3, 2, 0, 0, 0, 13, 0, 0, 2, 5, 0, 0, 0, 3, 3, 0, 0, 0, 1, 8, 0, 0, 1, 0,
    1, 0, 0, 0, 4, 1, 0, 0, 0, 0, 14, 0, 0, 0, 1, 9, 0, 0, 0, 1, 10, 0, 0, 0,
    0, 1, 11, 0, 0, 0, 1, 0, 2, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 5, 1, 0, 0, 0, 1,
    2, 1, 0, 0, 0, 0, 0, 2, 6, 0, 0, 0, 0, 0, 1, 12, 0, 0, 0, 0, 0, 3, 4, 0, 0,
    0, 0, 0, 7, 0, 0, 0, 0, 0, 1, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0])
RANDOM2 = bytes([10, 2, 14, 13, 0, 8, 2, 10, 2, 14, 13, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])
RANDOM3 = bytes([0, 0, 0, 4, 1, 4, 0, 0, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])
RANDOM4 = bytes([
    0, 0, 0, 0, 1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 4, 0, 0, 0, 5, 0, 0, 1, 1,
    0, 0, 1, 2, 0, 0, 1, 3, 0, 0, 1, 4, 0, 0, 2, 1, 0, 0, 0, 4, 0, 1, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0])
WITNESSES = [RANDOM1, RANDOM2, RANDOM3, RANDOM4]

FRAMING_HEADER = bytes([0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59])

# tests/test_framed.nim:140-158 ("buffer sizes"), input = byte(i) ramp
FRAMED_SIZES = [0, 1, 10, 16, 17, 18, 65535, 65536, 65537, 128 * 1024]


def mod10(n):
    return bytes((j % 10) + ord("a") for j in range(n))


def ramp(n):
    return bytes(i & 0xff for i in range(n))


def repeat_cases():
    """tests/test_snappy.nim:111-114"""
    return [b"aaaa" + b"b" * i + b"aaaabbbb" for i in range(1, 33)]


def block_boundary_lengths():
    """tests/test_snappy.nim:123-134"""
    return [n for m in range(1, 6) for n in range(m * 65536 - 5, m * 65536 + 6)]

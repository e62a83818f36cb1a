"""Stream adapters (nim-snappy_amd/streams.py): the host logic restated from snappy/faststreams.nim
and snappy/streams.nim.  The CPU tests drive it with the oracle as the codec backend (checker
role); tests/test_gpu_parity.py::test_stream_adapters runs the same checks on the HIP codec."""
import importlib
import io
import random

import pytest

import pyoracle as orc
from conftest import golden_file

streams = importlib.import_module("nim-snappy_amd.streams")


def check_adapters(be, batch_blocks):
    rng = random.Random(5)
    text = golden_file("alice29.txt") + golden_file("html")
    for n in (0, 1, 17, 65535, 65536, 65537, 200000, len(text)):
        src = text[:n]
        # tests/test_snappy.nim:56-57: the adapters write what the in-memory API returns
        out = io.BytesIO()
        streams.compress(io.BytesIO(src), len(src), out, batch_blocks=batch_blocks, be=be)
        assert out.getvalue() == orc.encode(src)
        out = io.BytesIO()
        streams.compress_framed(io.BytesIO(src), out, batch_blocks=batch_blocks, be=be)
        framed = out.getvalue()
        assert framed == orc.encode_framed(src)
        back = io.BytesIO()
        streams.uncompress_framed(io.BytesIO(framed), back, batch_blocks=batch_blocks, be=be)
        assert back.getvalue() == src
    # a stream with stored chunks, skippable chunks and a repeated stream header in the middle
    rnd = rng.randbytes(70000)
    f1, f2 = orc.encode_framed(rnd), orc.encode_framed(text[:100000])
    mixed = f1 + bytes([0x80, 3, 0, 0]) + b"pad" + f2  # f2 starts with a 0xff header chunk: skipped
    back = io.BytesIO()
    streams.uncompress_framed(io.BytesIO(mixed), back, batch_blocks=batch_blocks, be=be)
    assert back.getvalue() == rnd + text[:100000]


def check_adapter_errors(be, batch_blocks):
    text = golden_file("alice29.txt")[:150000]
    framed = orc.encode_framed(text)
    run = lambda data, **kw: streams.uncompress_framed(io.BytesIO(data), io.BytesIO(), be=be,
                                                       batch_blocks=batch_blocks, **kw)
    with pytest.raises(streams.UnexpectedEofError):
        run(framed[:5])                                   # faststreams.nim:93-94
    with pytest.raises(streams.MalformedSnappyData):
        run(b"\x00" + framed[1:])                         # :96-97
    with pytest.raises(streams.UnexpectedEofError):
        run(framed[:-10])                                 # :106-107
    with pytest.raises(streams.MalformedSnappyData):
        run(framed + b"\x01")                             # :139-140 trailing bytes
    with pytest.raises(streams.MalformedSnappyData):
        run(framed[:10] + bytes([0x02, 1, 0, 0, 0]))      # :130-134 reserved unskippable
    with pytest.raises(streams.MalformedSnappyData):
        run(framed[:10] + bytes([0x00, 3, 0, 0, 1, 2, 3]))  # :110-111 too short for a CRC
    # a damaged CRC in the last chunk: everything before it has been written (the reference writes
    # chunk by chunk), and checkIntegrity = false lets it pass
    bad = bytearray(framed)
    last = 10
    while last + 4 + (bad[last + 1] | bad[last + 2] << 8 | bad[last + 3] << 16) < len(bad):
        last += 4 + (bad[last + 1] | bad[last + 2] << 8 | bad[last + 3] << 16)
    bad[last + 4] ^= 0xff
    out = io.BytesIO()
    with pytest.raises(streams.MalformedSnappyData):
        streams.uncompress_framed(io.BytesIO(bytes(bad)), out, be=be, batch_blocks=batch_blocks)
    assert out.getvalue() == text[:131072]
    out = io.BytesIO()
    streams.uncompress_framed(io.BytesIO(bytes(bad)), out, check_integrity=False, be=be,
                              batch_blocks=batch_blocks)
    assert out.getvalue() == text
    with pytest.raises(streams.InputTooLarge):
        streams.compress(io.BytesIO(b""), 1 << 32, io.BytesIO(), be=be)  # streams.nim:19-21


@pytest.mark.parametrize("batch_blocks", [1, 2, 256])
def test_adapters_with_oracle_backend(batch_blocks):
    check_adapters(orc, batch_blocks)


def test_adapter_errors_with_oracle_backend():
    check_adapter_errors(orc, 2)
    check_adapter_errors(orc, 256)

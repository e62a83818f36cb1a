"""Stream adapters over the HIP codec: the batching front-ends of SURVEY.md 8(f) rank 2.

Mirrors the reference's adapters -- ``snappy/faststreams.nim`` (``compress`` :20-55,
``compressFramed`` :61-83, ``uncompressFramed`` :89-147) and ``snappy/streams.nim``
(``compress`` :9-41) -- over Python file-like objects (``read(n)`` / ``write(b)`` /
``flush()``).  The reference calls the codec once per 64 KiB piece; on a GPU that would pay a
PCIe round trip per piece, so these adapters accumulate ``batch_blocks`` pieces and hand them to
the device in one call.  The bytes written are identical to the in-memory API's
(tests/test_snappy.nim:56-57 checks exactly that for the reference's adapters).

Errors are the reference's exception classes (``snappy/exceptions.nim``).
"""
import io

MAX_BLOCK_LEN = 65536
MAX_UNCOMPRESSED_LEN = 0xFFFFFFFF
MAX_COMPRESSED_FRAME_DATA_LEN = 32 + MAX_BLOCK_LEN + MAX_BLOCK_LEN // 6  # codec.nim:217
FRAMING_HEADER = bytes([0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59])  # codec.nim:33-34
CHUNK_COMPRESSED, CHUNK_UNCOMPRESSED = 0x00, 0x01


class SnappyError(Exception):  # exceptions.nim:3-4
    pass


class SnappyDecodingError(SnappyError):
    pass


class SnappyEncodingError(SnappyError):
    pass


class UnexpectedEofError(SnappyDecodingError):
    pass


class MalformedSnappyData(SnappyDecodingError):
    pass


class InputTooLarge(SnappyEncodingError):
    pass


def _backend(be):
    if be is not None:
        return be
    import importlib
    return importlib.import_module(__package__)  # the HIP codec (fails loudly without its library)


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _read_exact(stream, n):
    """Up to n bytes; fewer only at the end of the stream."""
    parts = []
    got = 0
    while got < n:
        b = stream.read(n - got)
        if not b:
            break
        parts.append(b)
        got += len(b)
    return b"".join(parts)


def compress(input, input_len, output, batch_blocks=256, be=None):
    """``compress(input: Stream, inputLen: int, output: Stream)``, streams.nim:9-41 /
    faststreams.nim:20-55: ``varint(inputLen)`` followed by the block bodies of the 65 536-byte
    slices.  Like streams.nim, stops silently when the input ends early."""
    be = _backend(be)
    if input_len < 0 or input_len > MAX_UNCOMPRESSED_LEN:
        raise InputTooLarge("Input too large to be compressed with Snappy")
    output.write(_varint(input_len))
    read = 0
    while read < input_len:
        want = min(batch_blocks * MAX_BLOCK_LEN, input_len - read)
        data = _read_exact(input, want)
        if not data:
            break
        enc = be.encode(data)  # varint(len(data)) + the same block bodies: drop the batch's own header
        hdr = len(_varint(len(data)))
        output.write(enc[hdr:])
        read += len(data)
        if len(data) < want:
            break
    if hasattr(output, "flush"):
        output.flush()


def compress_framed(input, output, batch_blocks=256, be=None):
    """``compressFramed(input, output)``, faststreams.nim:61-83."""
    be = _backend(be)
    output.write(FRAMING_HEADER)
    while True:
        data = _read_exact(input, batch_blocks * MAX_BLOCK_LEN)
        if not data:
            break
        enc = be.encode_framed(data)
        output.write(enc[len(FRAMING_HEADER):])  # the batch's chunks without its own stream header
        if len(data) < batch_blocks * MAX_BLOCK_LEN:
            break
    if hasattr(output, "flush"):
        output.flush()


def uncompress_framed(input, output, check_integrity=True, batch_blocks=256, be=None):
    """``uncompressFramed(input, output, checkIntegrity)``, faststreams.nim:89-147: same checks in
    the same order, same exception classes.  Chunks are collected and decoded ``batch_blocks`` at a
    time; when a batch fails it is replayed chunk by chunk so that, like the reference, everything
    before the offending chunk has been written when the exception is raised."""
    be = _backend(be)
    head = _read_exact(input, len(FRAMING_HEADER))
    if len(head) < len(FRAMING_HEADER):
        raise UnexpectedEofError("Failed to read stream header")
    if head != FRAMING_HEADER:
        raise MalformedSnappyData("Invalid header value")

    batch = []  # whole chunks (4-byte header included) waiting for the device

    def run(chunks):
        if not chunks:
            return
        blob = b"".join(chunks)
        st, _read, _written, out = be.uncompress_framed(blob, len(chunks) * MAX_BLOCK_LEN, check_header=False,
                                                         check_integrity=check_integrity)
        if st == 0:
            output.write(out)
            return
        if len(chunks) == 1:
            ctype = chunks[0][0]
            if st == 3:  # crcMismatch
                raise MalformedSnappyData("Content CRC checksum failed")
            if ctype == CHUNK_COMPRESSED:
                raise MalformedSnappyData("Failed to decompress content")
            raise MalformedSnappyData("Invalid frame")
        for c in chunks:  # replay: write what precedes the bad chunk, then raise from it
            run([c])

    while True:
        h = _read_exact(input, 4)
        if len(h) < 4:
            trailing = h
            break
        cid = h[0]
        data_len = h[1] | (h[2] << 8) | (h[3] << 16)
        if data_len > MAX_COMPRESSED_FRAME_DATA_LEN:
            run(batch)
            raise MalformedSnappyData("Invalid frame length: %d" % data_len)
        body = _read_exact(input, data_len)
        if len(body) < data_len:
            run(batch)
            raise UnexpectedEofError("Failed to read the entire snappy frame")
        if cid in (CHUNK_COMPRESSED, CHUNK_UNCOMPRESSED):
            if data_len < 4:
                run(batch)
                raise MalformedSnappyData("Frame size too low to contain CRC checksum")
            if cid == CHUNK_UNCOMPRESSED and data_len - 4 > MAX_BLOCK_LEN:
                run(batch)
                raise MalformedSnappyData("Invalid frame length: %d" % data_len)
            batch.append(h + body)
            if len(batch) >= batch_blocks:
                run(batch)
                batch = []
        elif cid < 0x80:
            # reserved unskippable chunk (0x02-0x7f): the spec says it is an error
            run(batch)
            raise MalformedSnappyData("Invalid chunk type %02x" % cid)
        # else: reserved skippable chunk (0x80-0xfe) or a repeated stream header (0xff): skipped
    run(batch)
    if trailing:
        raise MalformedSnappyData("Input contains unknown trailing bytes")
    if hasattr(output, "flush"):
        output.flush()


def compress_bytes(data, batch_blocks=256, be=None):
    """``compress(input: openArray[byte], output)``, faststreams.nim:57-59, into a bytes object."""
    out = io.BytesIO()
    compress(io.BytesIO(data), len(data), out, batch_blocks=batch_blocks, be=be)
    return out.getvalue()

"""ctypes front-end of the CPU oracle (oracle/snappy_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (nim-snappy_amd) never does.  The Python-level helpers mirror the
reference's in-memory API (snappy.nim): encode/decode return b"" on error exactly where the
Nim wrappers return an empty seq.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

OK, BUFFER_TOO_SMALL, INVALID_INPUT, CRC_MISMATCH, UNKNOWN_CHUNK = range(5)
ENC_CPP_GE_LIMIT, ENC_CPP_SHIFT = 1, 2
MAX_UNCOMPRESSED_LEN = 0xFFFFFFFF


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when the reference checkout is present)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(
            os.path.join(_HERE, "snappy_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/snappy/crc32c.c") and (
            force or not os.path.exists(os.path.join(_HERE, "_ref", "libref_crc32c.so"))):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, stdout=subprocess.DEVNULL)


def _load():
    build()
    lib = ctypes.CDLL(_LIB)
    u8p, sz, szp = ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)
    lib.sor_max_compressed_len.restype = ctypes.c_uint64
    lib.sor_max_compressed_len.argtypes = [ctypes.c_uint32]
    lib.sor_max_compressed_len_framed.restype = ctypes.c_uint64
    lib.sor_max_compressed_len_framed.argtypes = [ctypes.c_int64]
    lib.sor_uncompressed_len.argtypes = [u8p, sz, ctypes.POINTER(ctypes.c_uint64)]
    lib.sor_uncompressed_len_framed.argtypes = [u8p, sz, ctypes.POINTER(ctypes.c_uint64)]
    for name in ("sor_masked_crc32c", "sor_crc32c"):
        getattr(lib, name).restype = ctypes.c_uint32
        getattr(lib, name).argtypes = [u8p, sz]
    for name in ("sor_encode_block", "sor_encode_frame"):
        getattr(lib, name).restype = sz
        getattr(lib, name).argtypes = [u8p, sz, ctypes.c_void_p]
    lib.sor_encode_block_ex.restype = sz
    lib.sor_encode_block_ex.argtypes = [u8p, sz, ctypes.c_void_p, ctypes.c_uint]
    lib.sor_decode_all_tags.argtypes = [u8p, sz, ctypes.c_void_p, sz, szp]
    for name in ("sor_compress", "sor_uncompress", "sor_compress_framed"):
        getattr(lib, name).argtypes = [u8p, sz, ctypes.c_void_p, sz, szp]
    lib.sor_compress_ex.argtypes = [u8p, sz, ctypes.c_void_p, sz, szp, ctypes.c_uint]
    lib.sor_uncompress_framed.argtypes = [u8p, sz, ctypes.c_void_p, sz, ctypes.c_int,
                                          ctypes.c_int, szp, szp]
    lib.sor_compress_blocks.restype = None
    lib.sor_compress_blocks.argtypes = [ctypes.c_void_p, sz, sz, ctypes.c_void_p, sz,
                                        ctypes.c_void_p]
    lib.sor_uncompress_blocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, sz,
                                          ctypes.c_void_p, sz]
    lib.sor_compress_blocks_mt.restype = None
    lib.sor_compress_blocks_mt.argtypes = [ctypes.c_void_p, sz, sz, ctypes.c_void_p, sz,
                                           ctypes.c_void_p, ctypes.c_int]
    lib.sor_encode_frames_mt.restype = None
    lib.sor_encode_frames_mt.argtypes = [ctypes.c_void_p, sz, sz, ctypes.c_void_p, sz,
                                         ctypes.c_void_p, ctypes.c_int]
    lib.sor_uncompress_blocks_mt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, sz,
                                             ctypes.c_void_p, sz, ctypes.c_int]
    return lib


lib = _load()


def max_compressed_len(n):
    return lib.sor_max_compressed_len(n)


def max_compressed_len_framed(n):
    return lib.sor_max_compressed_len_framed(n)


def masked_crc(data):
    return lib.sor_masked_crc32c(bytes(data), len(data))


def crc32c(data):
    return lib.sor_crc32c(bytes(data), len(data))


def uncompressed_len(data):
    """codec.nim:129 -- returns None where the reference returns err()."""
    v = ctypes.c_uint64()
    st = lib.sor_uncompressed_len(bytes(data), len(data), ctypes.byref(v))
    return v.value if st == OK else None


def uncompressed_len_framed(data):
    v = ctypes.c_uint64()
    st = lib.sor_uncompressed_len_framed(bytes(data), len(data), ctypes.byref(v))
    return v.value if st == OK else None


def encode_block(data, flags=0):
    data = bytes(data)
    out = ctypes.create_string_buffer(max_compressed_len(len(data)))
    n = lib.sor_encode_block_ex(data, len(data), out, flags)
    return out.raw[:n]


def encode_frame(data):
    data = bytes(data)
    out = ctypes.create_string_buffer(max_compressed_len(len(data)) + 8)
    n = lib.sor_encode_frame(data, len(data), out)
    return out.raw[:n]


def decode_all_tags(data, cap):
    data = bytes(data)
    out = ctypes.create_string_buffer(max(cap, 1))
    w = ctypes.c_size_t()
    st = lib.sor_decode_all_tags(data, len(data), out, cap, ctypes.byref(w))
    return st, out.raw[:w.value]


def compress(data, cap=None, flags=0):
    """snappy.nim:27 -- returns (status, bytes)."""
    data = bytes(data)
    if cap is None:
        cap = max_compressed_len(len(data))
    out = ctypes.create_string_buffer(max(cap, 1))
    w = ctypes.c_size_t()
    st = lib.sor_compress_ex(data, len(data), out, cap, ctypes.byref(w), flags)
    return st, out.raw[:w.value]


def uncompress(data, cap):
    """snappy.nim:84 -- returns (status, bytes)."""
    data = bytes(data)
    out = ctypes.create_string_buffer(max(cap, 1))
    w = ctypes.c_size_t()
    st = lib.sor_uncompress(data, len(data), out, cap, ctypes.byref(w))
    return st, out.raw[:w.value]


def encode(data, flags=0):
    """snappy.nim:66"""
    st, out = compress(data, flags=flags)
    return out if st == OK else b""


def decode(data, max_size=MAX_UNCOMPRESSED_LEN):
    """snappy.nim:112 -- b"" on any error, including declared size > max_size."""
    n = uncompressed_len(data)
    if n is None or n > max_size:
        return b""
    st, out = uncompress(data, n)
    return out if st == OK else b""


def compress_framed(data, cap=None):
    data = bytes(data)
    if cap is None:
        cap = max_compressed_len_framed(len(data))
    out = ctypes.create_string_buffer(max(cap, 1))
    w = ctypes.c_size_t()
    st = lib.sor_compress_framed(data, len(data), out, cap, ctypes.byref(w))
    return st, out.raw[:w.value]


def encode_framed(data):
    """snappy.nim:157"""
    st, out = compress_framed(data)
    assert st == OK
    return out


def uncompress_framed(data, cap, check_header=True, check_integrity=True):
    """snappy.nim:169 -- returns (status, read, written, bytes)."""
    data = bytes(data)
    out = ctypes.create_string_buffer(max(cap, 1))
    r, w = ctypes.c_size_t(), ctypes.c_size_t()
    st = lib.sor_uncompress_framed(data, len(data), out, cap, int(check_header),
                                   int(check_integrity), ctypes.byref(r), ctypes.byref(w))
    return st, r.value, w.value, out.raw[:w.value]


def decode_framed(data, max_size=2**63 - 1, check_integrity=True):
    """snappy.nim:269 -- b"" on any error."""
    n = uncompressed_len_framed(data)
    if n is None or n > max_size:
        return b""
    st, _, _, out = uncompress_framed(data, n, check_integrity=check_integrity)
    return out if st == OK else b""

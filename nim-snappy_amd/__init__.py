"""nim-snappy_amd -- MI355X-native Snappy block / framed codec behind nim-snappy's API.

Host-side mirror of the reference's in-memory API (snappy.nim) over the C ABI of
include/snappy_hip.h.  Same names, argument meaning and error behaviour as the reference:

    encode / decode / compress / uncompress            snappy.nim:27-128
    encode_framed / decode_framed / compress_framed /
    uncompress_framed                                   snappy.nim:130-290
    encode_block / encode_frame / decode_all_tags       snappy/encoder.nim:184,385 decoder.nim:20
    masked_crc / max_compressed_len(_framed) /
    uncompressed_len(_framed)                           snappy/codec.nim

All codec work runs in the HIP kernels of csrc/ (gfx950).  There is no CPU fallback: if the
shared library is missing the import fails, and without a GPU every codec call raises
DeviceError.  The package never imports anything from oracle/.

The directory name contains a hyphen, so import it with
    importlib.import_module("nim-snappy_amd")
"""
import ctypes
import os

try:  # PyTorch-ROCm is plumbing only (device memory / streams / torch.distributed in bench.py);
    import torch  # importing it first makes this process use ONE HIP runtime (same SONAME)
except Exception:  # pragma: no cover
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsnappy_hip.so")
# (test hook: a variant build of the SAME sources -- fault injection, A/B experiments -- in place of the shipped library;
# tests/test_gpu_faults.py, tools/ab.sh)
LIB_OVERRIDDEN = bool(os.environ.get("SNAPPY_HIP_LIBRARY"))
if LIB_OVERRIDDEN:
    import sys as _sys
    LIB_PATH = os.environ["SNAPPY_HIP_LIBRARY"]
    print("nim-snappy_amd: SNAPPY_HIP_LIBRARY is set -- loading %s instead of the shipped library" % LIB_PATH,
          file=_sys.stderr)

OK, BUFFER_TOO_SMALL, INVALID_INPUT, CRC_MISMATCH, UNKNOWN_CHUNK = range(5)
DEVICE_ERROR = 100
UNIT_BODY, UNIT_RAW, UNIT_FRAME = 0, 1, 2
MAX_UNCOMPRESSED_LEN = 0xFFFFFFFF
MAX_BLOCK_LEN = 65536
SLOT_STRIDE = 76800

#: every symbol include/snappy_hip.h declares
ABI_SYMBOLS = [
    "snappy_hip_max_compressed_len", "snappy_hip_max_compressed_len_framed",
    "snappy_hip_uncompressed_len", "snappy_hip_uncompressed_len_framed",
    "snappy_hip_compress", "snappy_hip_uncompress", "snappy_hip_compress_framed",
    "snappy_hip_uncompress_framed", "snappy_hip_masked_crc32c", "snappy_hip_encode_block",
    "snappy_hip_encode_frame", "snappy_hip_decode_all_tags", "snappy_hip_ctx_create",
    "snappy_hip_ctx_destroy", "snappy_hip_ctx_sync", "snappy_hip_last_error",
    "snappy_hip_encode_blocks_d", "snappy_hip_pack_d", "snappy_hip_decode_blocks_d",
    "snappy_hip_crc32c_d", "snappy_hip_ctx_timing", "snappy_hip_ctx_kernel_ms",
    "snappy_hip_compress_framed_d", "snappy_hip_uncompress_framed_d", "snappy_hip_uncompress_d",
    "snappy_hip_compress_shards", "snappy_hip_compress_shards_staged", "snappy_hip_release_pool",
    "snappy_hip_ctx_launch_order",
]


class DeviceError(RuntimeError):
    """The HIP path could not run (no GPU, HIP failure).  Never replaced by a CPU result."""


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "nim-snappy_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
        "g.build()'` (hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

_u8p, _sz = ctypes.c_char_p, ctypes.c_size_t
_szp = ctypes.POINTER(ctypes.c_size_t)
_vp = ctypes.c_void_p
lib.snappy_hip_max_compressed_len.restype = ctypes.c_uint64
lib.snappy_hip_max_compressed_len.argtypes = [ctypes.c_uint32]
lib.snappy_hip_max_compressed_len_framed.restype = ctypes.c_uint64
lib.snappy_hip_max_compressed_len_framed.argtypes = [ctypes.c_int64]
lib.snappy_hip_uncompressed_len.argtypes = [_u8p, _sz, ctypes.POINTER(ctypes.c_uint64)]
lib.snappy_hip_uncompressed_len_framed.argtypes = [_u8p, _sz, ctypes.POINTER(ctypes.c_uint64)]
for _n in ("snappy_hip_compress", "snappy_hip_uncompress", "snappy_hip_compress_framed",
           "snappy_hip_encode_block", "snappy_hip_encode_frame", "snappy_hip_decode_all_tags"):
    getattr(lib, _n).argtypes = [_u8p, _sz, _vp, _sz, _szp]
lib.snappy_hip_uncompress_framed.argtypes = [_u8p, _sz, _vp, _sz, ctypes.c_int, ctypes.c_int,
                                             _szp, _szp]
lib.snappy_hip_masked_crc32c.restype = ctypes.c_uint32
lib.snappy_hip_masked_crc32c.argtypes = [_u8p, _sz, ctypes.POINTER(ctypes.c_int)]
lib.snappy_hip_last_error.restype = ctypes.c_char_p
lib.snappy_hip_ctx_create.argtypes = [ctypes.POINTER(_vp), ctypes.c_int]
lib.snappy_hip_ctx_destroy.argtypes = [_vp]
lib.snappy_hip_ctx_destroy.restype = None
lib.snappy_hip_ctx_sync.argtypes = [_vp, _vp]
lib.snappy_hip_encode_blocks_d.argtypes = [_vp, _vp, ctypes.c_uint64, ctypes.c_uint32,
                                           ctypes.c_int, _vp, ctypes.c_uint32, _vp, _vp]
lib.snappy_hip_pack_d.argtypes = [_vp, _vp, ctypes.c_uint32, _vp, ctypes.c_uint64,
                                  ctypes.c_uint64, _vp, _vp, _vp]
lib.snappy_hip_decode_blocks_d.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_uint64, ctypes.c_int,
                                           _vp, _vp, _vp, _vp, _vp, _vp, _vp]
lib.snappy_hip_crc32c_d.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_uint64, _vp, _vp]
lib.snappy_hip_compress_framed_d.argtypes = [_vp, _vp, ctypes.c_uint64, _vp, ctypes.c_uint64,
                                             ctypes.POINTER(ctypes.c_uint64), _vp]
lib.snappy_hip_uncompress_framed_d.argtypes = [_vp, _vp, ctypes.c_uint64, _vp, ctypes.c_uint64,
                                               ctypes.c_int, ctypes.c_int,
                                               ctypes.POINTER(ctypes.c_uint64),
                                               ctypes.POINTER(ctypes.c_uint64), _vp]
lib.snappy_hip_compress_shards.argtypes = [ctypes.POINTER(_vp), ctypes.c_int, ctypes.POINTER(_vp),
                                           ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, _vp,
                                           ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
                                           ctypes.POINTER(ctypes.c_uint64)]
lib.snappy_hip_compress_shards_staged.argtypes = [ctypes.POINTER(_vp), ctypes.c_int, ctypes.POINTER(_vp),
                                                  ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64, ctypes.c_int,
                                                  _vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
                                                  ctypes.POINTER(ctypes.c_uint64)]
lib.snappy_hip_uncompress_d.argtypes = [_vp, _vp, ctypes.c_uint64, _vp, ctypes.c_uint64,
                                        ctypes.POINTER(ctypes.c_uint64), _vp]
lib.snappy_hip_release_pool.restype = None
lib.snappy_hip_release_pool.argtypes = []
lib.snappy_hip_ctx_timing.argtypes = [_vp, ctypes.c_int]
lib.snappy_hip_ctx_launch_order.argtypes = [_vp, ctypes.c_int]
lib.snappy_hip_ctx_kernel_ms.restype = ctypes.c_double
lib.snappy_hip_ctx_kernel_ms.argtypes = [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]


def release_pool():
    """Frees the idle pooled contexts of the host-buffer calls (their device workspace)."""
    lib.snappy_hip_release_pool()


def last_error():
    return (lib.snappy_hip_last_error() or b"").decode(errors="replace")


def _check_device(st):
    if st == DEVICE_ERROR:
        raise DeviceError("HIP path unavailable: " + last_error())
    return st


# ---- host-side scalar helpers (codec.nim) --------------------------------------------------------
def max_compressed_len(n):
    return lib.snappy_hip_max_compressed_len(n)


def max_compressed_len_framed(n):
    return lib.snappy_hip_max_compressed_len_framed(n)


def uncompressed_len(data):
    """codec.nim:129 -- None where the reference returns err()."""
    v = ctypes.c_uint64()
    st = lib.snappy_hip_uncompressed_len(bytes(data), len(data), ctypes.byref(v))
    return v.value if st == OK else None


def uncompressed_len_framed(data):
    """codec.nim:178"""
    v = ctypes.c_uint64()
    st = lib.snappy_hip_uncompressed_len_framed(bytes(data), len(data), ctypes.byref(v))
    return v.value if st == OK else None


# ---- host-buffer codec API (snappy.nim) -------------------------------------------------------
def _call5(fn, data, cap):
    data = bytes(data)
    out = ctypes.create_string_buffer(max(cap, 1))
    w = ctypes.c_size_t()
    st = _check_device(fn(data, len(data), out, cap, ctypes.byref(w)))
    return st, out.raw[:w.value]


def masked_crc(data):
    """codec.nim:71"""
    data = bytes(data)
    st = ctypes.c_int()
    v = lib.snappy_hip_masked_crc32c(data, len(data), ctypes.byref(st))
    _check_device(st.value)
    if st.value != OK:
        raise ValueError("masked_crc: status %d" % st.value)
    return v


def compress(data, cap=None):
    """snappy.nim:27 -- (status, bytes)"""
    if cap is None:
        cap = max_compressed_len(len(data))
    return _call5(lib.snappy_hip_compress, data, cap)


def uncompress(data, cap):
    """snappy.nim:84 -- (status, bytes)"""
    return _call5(lib.snappy_hip_uncompress, data, cap)


def encode(data):
    """snappy.nim:66 -- b"" on failure"""
    st, out = compress(data)
    return out if st == OK else b""


def decode(data, max_size=MAX_UNCOMPRESSED_LEN):
    """snappy.nim:112 -- b"" on any error, including declared size > max_size"""
    n = uncompressed_len(data)
    if n is None or n > max_size:
        return b""
    st, out = uncompress(data, n)
    return out if st == OK else b""


def encode_block(data):
    """encoder.nim:184"""
    st, out = _call5(lib.snappy_hip_encode_block, data, max_compressed_len(len(data)))
    if st != OK:
        raise ValueError("encode_block: status %d" % st)
    return out


def encode_frame(data):
    """encoder.nim:385"""
    st, out = _call5(lib.snappy_hip_encode_frame, data, max_compressed_len(len(data)) + 8)
    if st != OK:
        raise ValueError("encode_frame: status %d" % st)
    return out


def decode_all_tags(data, cap):
    """decoder.nim:20 -- (status, bytes)"""
    return _call5(lib.snappy_hip_decode_all_tags, data, cap)


def compress_framed(data, cap=None):
    """snappy.nim:130 -- (status, bytes)"""
    if cap is None:
        cap = max_compressed_len_framed(len(data))
    return _call5(lib.snappy_hip_compress_framed, data, cap)


def encode_framed(data):
    """snappy.nim:157"""
    st, out = compress_framed(data)
    if st != OK:
        raise ValueError("encode_framed: status %d" % st)
    return out


def uncompress_framed(data, cap, check_header=True, check_integrity=True):
    """snappy.nim:169 -- (status, read, written, bytes)"""
    data = bytes(data)
    out = ctypes.create_string_buffer(max(cap, 1))
    r, w = ctypes.c_size_t(), ctypes.c_size_t()
    st = _check_device(lib.snappy_hip_uncompress_framed(
        data, len(data), out, cap, int(check_header), int(check_integrity), ctypes.byref(r),
        ctypes.byref(w)))
    return st, r.value, w.value, out.raw[:w.value]


def decode_framed(data, max_size=2**63 - 1, check_integrity=True):
    """snappy.nim:269 -- b"" on any error"""
    n = uncompressed_len_framed(data)
    if n is None or n > max_size:
        return b""
    st, _, _, out = uncompress_framed(data, n, check_integrity=check_integrity)
    return out if st == OK else b""


# ---- device-resident batch API ---------------------------------------------------------------
def _ptr(t):
    """torch tensor / int / None -> device pointer"""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def _after_torch(stream, *tensors):
    """The context launches on its OWN stream.  Work that PyTorch has queued on its current stream
    for these tensors (torch.zeros fills, copies, ...) is not ordered against that, so a fill could
    land after a kernel's result.  When the caller did not pass a stream of their own, wait for
    PyTorch's current stream first (a few microseconds on an idle stream)."""
    if stream is not None:
        return
    for t in tensors:
        if t is not None and not isinstance(t, int):
            import torch
            torch.cuda.current_stream(t.device).synchronize()
            return


def compress_shards(ctxs, d_ins, lens, out_ptr, cap, framed=True, stage_blocks=0):
    """snappy_hip_compress_shards(_staged): shard k (device tensor d_ins[k], lens[k] bytes, on ctxs[k]'s GPU)
    is encoded there, and all shards land in ONE host buffer (out_ptr: address, ideally page-locked) at
    their scanned offsets.  stage_blocks = S > 0: d_ins[k] holds context k's STAGES back to back -- stage j of
    context k is the global block range [(j n + k) S, (j n + k + 1) S) -- and a context downloads a stage while
    it encodes the next.  Returns (written, offsets[n + 1])."""
    n = len(ctxs)
    for c, t in zip(ctxs, d_ins):
        _after_torch(None, t)
    hs = (_vp * n)(*[c._h for c in ctxs])
    ps = (_vp * n)(*[_ptr(t) for t in d_ins])
    ls = (ctypes.c_uint64 * n)(*[int(x) for x in lens])
    offs = (ctypes.c_uint64 * (n + 1))()
    w = ctypes.c_uint64()
    st = _check_device(lib.snappy_hip_compress_shards_staged(hs, n, ps, ls, int(stage_blocks), int(framed),
                                                             out_ptr, cap, ctypes.byref(w), offs))
    if st != OK:
        raise ValueError("compress_shards: status %d" % st)
    return w.value, list(offs)


class Context:
    """One GPU: stream, CRC / probe tables.  Device pointers are torch tensors (or ints)."""

    def __init__(self, device=0):
        h = _vp()
        _check_device(lib.snappy_hip_ctx_create(ctypes.byref(h), int(device)))
        self._h = h
        self.device = int(device)

    def close(self):
        if self._h:
            lib.snappy_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def sync(self, stream=None):
        _check_device(lib.snappy_hip_ctx_sync(self._h, stream))

    def encode_blocks(self, d_in, total_len, d_slots, d_sizes, unit=UNIT_RAW,
                      block_len=MAX_BLOCK_LEN, slot_stride=SLOT_STRIDE, stream=None):
        _after_torch(stream, d_in, d_slots, d_sizes)
        st = _check_device(lib.snappy_hip_encode_blocks_d(
            self._h, _ptr(d_in), total_len, block_len, unit, _ptr(d_slots), slot_stride,
            _ptr(d_sizes), stream))
        if st != OK:
            raise ValueError("encode_blocks: status %d" % st)

    def pack(self, d_slots, d_sizes, n_blocks, d_out, d_offsets, base=0,
             slot_stride=SLOT_STRIDE, stream=None):
        _after_torch(stream, d_slots, d_sizes, d_out, d_offsets)
        st = _check_device(lib.snappy_hip_pack_d(
            self._h, _ptr(d_slots), slot_stride, _ptr(d_sizes), n_blocks, base, _ptr(d_out),
            _ptr(d_offsets), stream))
        if st != OK:
            raise ValueError("pack: status %d" % st)

    def decode_blocks(self, d_in, d_in_off, d_in_len, n_units, d_out, d_out_off, d_out_cap,
                      d_out_len, d_status, unit=UNIT_RAW, d_crc=None, stream=None):
        _after_torch(stream, d_in, d_out)
        st = _check_device(lib.snappy_hip_decode_blocks_d(
            self._h, _ptr(d_in), _ptr(d_in_off), _ptr(d_in_len), n_units, unit, _ptr(d_out),
            _ptr(d_out_off), _ptr(d_out_cap), _ptr(d_out_len), _ptr(d_status), _ptr(d_crc),
            stream))
        if st != OK:
            raise ValueError("decode_blocks: status %d" % st)

    def crc32c(self, d_in, d_off, d_len, n_units, d_crc, stream=None):
        _after_torch(stream, d_in, d_crc)
        st = _check_device(lib.snappy_hip_crc32c_d(
            self._h, _ptr(d_in), _ptr(d_off), _ptr(d_len), n_units, _ptr(d_crc), stream))
        if st != OK:
            raise ValueError("crc32c: status %d" % st)

    def compress_framed(self, d_in, n, d_out, cap, stream=None):
        """compressFramed of d_in[0:n] (device) into d_out (device); returns the stream's length."""
        _after_torch(stream, d_in, d_out)
        w = ctypes.c_uint64()
        st = _check_device(lib.snappy_hip_compress_framed_d(
            self._h, _ptr(d_in), n, _ptr(d_out), cap, ctypes.byref(w), stream))
        if st != OK:
            raise ValueError("compress_framed: status %d" % st)
        return w.value

    def uncompress_framed(self, d_in, n, d_out, cap, check_header=True, check_integrity=True,
                          stream=None):
        """uncompressFramed of the stream d_in[0:n] (device) into d_out (device): (status, read,
        written), the reference's Result[(read, written), FrameError]."""
        _after_torch(stream, d_in, d_out)
        r, w = ctypes.c_uint64(), ctypes.c_uint64()
        st = _check_device(lib.snappy_hip_uncompress_framed_d(
            self._h, _ptr(d_in), n, _ptr(d_out), cap, int(check_header), int(check_integrity),
            ctypes.byref(r), ctypes.byref(w), stream))
        return st, r.value, w.value

    def uncompress(self, d_in, n, d_out, cap, stream=None):
        """uncompress of ONE raw buffer d_in[0:n] (device) into d_out (device): (status, written)."""
        _after_torch(stream, d_in, d_out)
        w = ctypes.c_uint64()
        st = _check_device(lib.snappy_hip_uncompress_d(self._h, _ptr(d_in), n, _ptr(d_out), cap,
                                                       ctypes.byref(w), stream))
        return st, w.value

    def launch_order(self, enable):
        """large batches in sorted launch order (default) or in the caller's order"""
        lib.snappy_hip_ctx_launch_order(self._h, int(enable))

    def timing(self, enable):
        lib.snappy_hip_ctx_timing(self._h, int(enable))

    def kernel_ms(self, which):
        """(average ms per launch, launches) for 0 decode / 1 encode / 2 crc / 3 pack"""
        n = ctypes.c_uint64()
        ms = lib.snappy_hip_ctx_kernel_ms(self._h, which, ctypes.byref(n))
        return ms, n.value

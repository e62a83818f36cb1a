#!/usr/bin/env python3
"""Generate tests/golden/vectors.json in the BUILD CONTAINER (never on the GPU box).

Sources of truth used here and nowhere else at run time:
  * /opt/conda/lib/libsnappy.so.1.1.8  -- Google C++ snappy, the reference's own `cppLib`
    differential oracle (tests/cpp_snappy.nim:8-11 binds the same four snappy-c.h functions);
  * oracle/_ref/libref_crc32c.so       -- the reference's snappy/crc32c.c compiled as-is
    (oracle/Makefile target `ref`).
The data files under tests/golden/data and tests/golden/stream_compressed are verbatim copies
of the reference's test fixtures (tests/data, tests/stream_compressed).

What is recorded (data only -- inputs are the fixture files / seeded generators below):
  files[name]        sha256/len of the fixture, libsnappy 1.1.8 compressed len+sha256, the
                     reference-CRC (masked) of every 64 KiB slice and of the whole file, and
                     a regression self-pin of the Nim-semantics oracle output
  rawsnappy          libsnappy decode of Mark.Twain-Tom.Sawyer.txt.rawsnappy (len, sha256)
  baddata            libsnappy verdict on baddata{1,2,3}.snappy
  framed             per golden .sz stream: chunk table and reference-CRC check of each chunk
  crc_kats           masked CRCs from the reference build for small / edge-length inputs
  synthetic          libsnappy 1.1.8 compressed len+sha256 of the synthetic inputs of
                     tests/test_snappy.nim:110-134 (pins oracle flags=3 == C++ on them too)
"""
import ctypes
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as orc  # noqa: E402

S = ctypes.CDLL("/opt/conda/lib/libsnappy.so.1.1.8")
R = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_crc32c.so"))
R.masked_crc32c.restype = ctypes.c_uint32
R.masked_crc32c.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
S.snappy_max_compressed_length.restype = ctypes.c_size_t
S.snappy_max_compressed_length.argtypes = [ctypes.c_size_t]


def sha(b):
    return hashlib.sha256(b).hexdigest()


def ref_crc(b):
    return R.masked_crc32c(bytes(b), len(b))


def cpp_encode(d):
    cap = S.snappy_max_compressed_length(len(d))
    out = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(cap)
    assert S.snappy_compress(bytes(d), ctypes.c_size_t(len(d)), out, ctypes.byref(n)) == 0
    return out.raw[:n.value]


def cpp_decode(c):
    n = ctypes.c_size_t()
    if S.snappy_uncompressed_length(bytes(c), ctypes.c_size_t(len(c)), ctypes.byref(n)) != 0:
        return None
    out = ctypes.create_string_buffer(max(n.value, 1))
    if S.snappy_uncompress(bytes(c), ctypes.c_size_t(len(c)), out, ctypes.byref(n)) != 0:
        return None
    return out.raw[:n.value]


def synthetic_inputs():
    """tests/test_snappy.nim:110-134 -- name -> bytes (kept small enough to list)."""
    out = {}
    for i in range(1, 33):
        out["repeat_%d" % i] = b"aaaa" + b"b" * i + b"aaaabbbb"
    i = 1
    while i < 20000:
        out["mod10_%d" % i] = bytes((j % 10) + ord("a") for j in range(i))
        i += 23
    for m in range(1, 6):
        for n in range(m * 65536 - 5, m * 65536 + 6):
            out["zeros_%d" % n] = bytes(n)
            out["mod10_%d" % n] = bytes((j % 10) + ord("a") for j in range(n))
    return out


def main():
    data_dir = os.path.join(ROOT, "tests", "golden", "data")
    vec = {"generator": "tools/gen_golden.py", "libsnappy": "1.1.8 (/opt/conda/lib)",
           "ref_crc": "status-im/nim-snappy snappy/crc32c.c compiled with gcc (oracle/_ref)"}

    files = {}
    for f in sorted(glob.glob(os.path.join(data_dir, "*"))):
        name = os.path.basename(f)
        if name.endswith((".snappy", ".rawsnappy")) or name == "COPYING":
            continue
        d = open(f, "rb").read()
        c = cpp_encode(d)
        assert cpp_decode(c) == d
        nim = orc.encode(d)
        assert cpp_decode(nim) == d, name  # tests/test_snappy.nim:60
        files[name] = {
            "len": len(d), "sha256": sha(d),
            "libsnappy_len": len(c), "libsnappy_sha256": sha(c),
            "oracle_cppflags_equals_libsnappy": orc.encode(d, flags=3) == c,
            "oracle_nim_len": len(nim), "oracle_nim_sha256": sha(nim),
            "oracle_nim_equals_libsnappy": nim == c,
            "ref_masked_crc_whole": ref_crc(d),
            "ref_masked_crc_64k": [ref_crc(d[i:i + 65536]) for i in range(0, len(d), 65536)],
        }
    vec["files"] = files

    raw = open(os.path.join(data_dir, "Mark.Twain-Tom.Sawyer.txt.rawsnappy"), "rb").read()
    dec = cpp_decode(raw)
    vec["rawsnappy"] = {"len": len(raw), "decoded_len": len(dec), "decoded_sha256": sha(dec),
                        "libsnappy_reencode_equals_golden": cpp_encode(dec) == raw}

    vec["baddata"] = {}
    for n in ("baddata1.snappy", "baddata2.snappy", "baddata3.snappy"):
        b = open(os.path.join(data_dir, n), "rb").read()
        vec["baddata"][n] = {"len": len(b), "libsnappy_accepts": cpp_decode(b) is not None}

    framed = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "stream_compressed", "*"))):
        b = open(f, "rb").read()
        pos, chunks = 10, []
        assert b[:10] == bytes([0xff, 6, 0, 0, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59])
        while pos < len(b):
            cid = b[pos]
            dl = int.from_bytes(b[pos + 1:pos + 4], "little")
            crc = int.from_bytes(b[pos + 4:pos + 8], "little")
            body = b[pos + 8:pos + 4 + dl]
            plain = cpp_decode(body) if cid == 0 else body
            chunks.append({"id": cid, "data_len": dl, "crc": crc, "uncompressed_len": len(plain),
                           "ref_crc_ok": ref_crc(plain) == crc})
            pos += 4 + dl
        framed[os.path.basename(f)] = chunks
    vec["framed"] = framed

    kats = []
    pat = bytes((i * 131 + 7) & 0xff for i in range(70000))
    for n in list(range(0, 41)) + [63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 65535, 65536, 65537]:
        kats.append({"pattern": "(i*131+7)&255", "len": n, "masked": ref_crc(pat[:n])})
    kats.append({"hex": "313233343536373839", "masked": ref_crc(b"123456789")})
    kats.append({"zeros": 32, "masked": ref_crc(bytes(32))})
    kats.append({"zeros": 65536, "masked": ref_crc(bytes(65536))})
    vec["crc_kats"] = kats

    syn = {}
    for name, d in synthetic_inputs().items():
        c = cpp_encode(d)
        nim = orc.encode(d)
        assert cpp_decode(nim) == d, name
        assert orc.encode(d, flags=3) == c, name
        syn[name] = {"libsnappy_len": len(c), "libsnappy_sha256": sha(c),
                     "oracle_nim_equals_libsnappy": nim == c}
    vec["synthetic"] = syn

    out = os.path.join(ROOT, "tests", "golden", "vectors.json")
    with open(out, "w") as fh:
        json.dump(vec, fh, indent=1, sort_keys=True)
    print("wrote", out, os.path.getsize(out), "bytes")
    diff = [k for k, v in files.items() if not v["oracle_nim_equals_libsnappy"]]
    print("files where Nim semantics differ from libsnappy 1.1.8:", diff)
    print("synthetic inputs where they differ:",
          [k for k, v in syn.items() if not v["oracle_nim_equals_libsnappy"]][:20])
    assert all(v["oracle_cppflags_equals_libsnappy"] for v in files.values())


if __name__ == "__main__":
    main()

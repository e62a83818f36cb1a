"""Device-resident batch API (the path bench.py measures) against the oracle, on the seeded
synthetic corpus of SURVEY.md 8d, plus the full-size size-independent properties."""
import hashlib
import os
import random

import numpy as np
import pytest

import cases

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


def _dev(torch, arr):
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


def _encode_pack(hip, torch, ctx, d_in, total_len, unit, base=0):
    nb = (total_len + 65535) // 65536
    d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device="cuda")
    d_sizes = torch.empty(nb, dtype=torch.int32, device="cuda")
    d_offsets = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
    ctx.encode_blocks(d_in, total_len, d_slots, d_sizes, unit=unit)
    ctx.sync()
    total = int(d_sizes.to(torch.int64).sum().item()) + base
    d_out = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
    ctx.pack(d_slots, d_sizes, nb, d_out, d_offsets, base=base)
    ctx.sync()
    assert int(d_offsets[-1].item()) == total
    return d_slots, d_sizes, d_offsets, d_out, total


def test_corpus_sample_bit_exact(hip, orc, torch_mod):
    """512 mixed blocks: per-block GPU encoding == oracle, packed stream == oracle concat,
    GPU decode of it == source, statuses ok, CRCs == oracle."""
    import corpus
    torch = torch_mod
    nb = 512
    blocks = corpus.make_blocks(0, nb)
    flat = blocks.reshape(-1)
    ctx = hip.Context(0)
    d_in = _dev(torch, flat)
    d_slots, d_sizes, d_offsets, d_out, total = _encode_pack(hip, torch, ctx, d_in, flat.size,
                                                             hip.UNIT_RAW)
    sizes = d_sizes.cpu().numpy()
    packed = d_out[:total].cpu().numpy().tobytes()
    expect = [orc.encode(blocks[i].tobytes()) for i in range(nb)]
    assert [int(s) for s in sizes] == [len(e) for e in expect]
    assert packed == b"".join(expect)

    d_in_len = d_sizes
    d_in_off = d_offsets[:nb].contiguous()
    d_out_off = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
    d_out_cap = torch.full((nb,), 65536, dtype=torch.int32, device="cuda")
    d_out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_status = torch.full((nb,), 77, dtype=torch.int32, device="cuda")
    d_crc = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_dec = torch.zeros(nb * 65536, dtype=torch.uint8, device="cuda")
    ctx.decode_blocks(d_out, d_in_off, d_in_len, nb, d_dec, d_out_off, d_out_cap, d_out_len,
                      d_status, unit=hip.UNIT_RAW, d_crc=d_crc)
    ctx.sync()
    assert (d_status.cpu().numpy() == 0).all()
    assert (d_out_len.cpu().numpy() == 65536).all()
    assert (d_dec.cpu().numpy() == flat).all()
    crcs = d_crc.cpu().numpy().view(np.uint32)
    assert [int(c) for c in crcs] == [orc.masked_crc(blocks[i].tobytes()) for i in range(nb)]
    # the same without the CRC buffer: the ring-window instantiation of the indexed decoder goes first
    d_dec.zero_()
    d_status.fill_(77)
    d_out_len.zero_()
    ctx.decode_blocks(d_out, d_in_off, d_in_len, nb, d_dec, d_out_off, d_out_cap, d_out_len, d_status, unit=hip.UNIT_RAW)
    ctx.sync()
    assert (d_status.cpu().numpy() == 0).all() and (d_out_len.cpu().numpy() == 65536).all()
    assert (d_dec.cpu().numpy() == flat).all()
    ctx.close()


def test_framed_batch_bit_exact(hip, orc, torch_mod):
    import corpus
    torch = torch_mod
    nb = 96
    blocks = corpus.make_blocks(1000, nb)
    flat = blocks.reshape(-1)[:nb * 65536 - 12345]  # ragged last frame
    ctx = hip.Context(0)
    d_in = _dev(torch, flat)
    _, _, _, d_out, total = _encode_pack(hip, torch, ctx, d_in, flat.size, hip.UNIT_FRAME, base=10)
    got = d_out[:total].cpu().numpy().tobytes()
    want = orc.encode_framed(flat.tobytes())
    assert got[10:] == want[10:]  # the 10-byte stream identifier is written by the host API
    ctx.close()


def test_ragged_and_body_units(hip, orc, torch_mod):
    """short last block, BODY units, tiny total"""
    torch = torch_mod
    ctx = hip.Context(0)
    text = np.frombuffer(open(os.path.join(os.path.dirname(__file__), "golden", "data",
                                           "lcet10.txt"), "rb").read(), dtype=np.uint8)
    for total in (1, 16, 17, 65536, 65537, 3 * 65536 + 777):
        src = text[:total]
        d_in = _dev(torch, src)
        d_slots, d_sizes, d_offsets, d_out, tot = _encode_pack(hip, torch, ctx, d_in, total,
                                                               hip.UNIT_BODY)
        want = b"".join(orc.encode_block(src[i:i + 65536].tobytes())
                        for i in range(0, total, 65536))
        assert d_out[:tot].cpu().numpy().tobytes() == want
    ctx.close()


@pytest.mark.parametrize("nb", [256, 1100])
def test_corrupt_units_match_oracle(hip, orc, torch_mod, nb):
    """per-unit status of damaged streams == oracle's uncompress verdict (256 units: the index pass's team of four waves a
    unit; 1 100: one wave a unit -- index_kernel.h)"""
    import corpus
    torch = torch_mod
    rng = random.Random(99)
    blocks = corpus.make_blocks(5000, nb)
    units, caps = [], []
    for i in range(nb):
        e = bytearray(orc.encode(blocks[i].tobytes()))
        kind = rng.randrange(5)
        if kind == 0:
            pass
        elif kind == 1:
            e[rng.randrange(len(e))] ^= 1 << rng.randrange(8)
        elif kind == 2:
            e = e[:rng.randrange(1, len(e))]
        elif kind == 3:
            e += bytes([rng.randrange(256)])
        else:
            for _ in range(4):
                e[rng.randrange(len(e))] = rng.randrange(256)
        units.append(bytes(e))
        caps.append(65536 if rng.random() < 0.9 else rng.randrange(0, 65536))
    offs = np.zeros(nb, dtype=np.int64)
    offs[1:] = np.cumsum([len(u) for u in units])[:-1]
    ctx = hip.Context(0)
    d_in = _dev(torch, np.frombuffer(b"".join(units) + bytes(64), dtype=np.uint8))
    d_in_off = _dev(torch, offs)
    d_in_len = _dev(torch, np.array([len(u) for u in units], dtype=np.int32))
    d_out_off = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
    d_out_cap = _dev(torch, np.array(caps, dtype=np.int32))
    d_out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_status = torch.full((nb,), 77, dtype=torch.int32, device="cuda")
    d_dec = torch.zeros(nb * 65536, dtype=torch.uint8, device="cuda")
    ctx.decode_blocks(d_in, d_in_off, d_in_len, nb, d_dec, d_out_off, d_out_cap, d_out_len,
                      d_status, unit=hip.UNIT_RAW)
    ctx.sync()
    status = d_status.cpu().numpy()
    lens = d_out_len.cpu().numpy()
    dec = d_dec.cpu().numpy()
    seen = set()
    for i in range(nb):
        st, out = orc.uncompress(units[i], caps[i])
        assert int(status[i]) == st, (i, int(status[i]), st)
        seen.add(st)
        if st == 0:
            assert int(lens[i]) == len(out)
            assert dec[i * 65536:i * 65536 + len(out)].tobytes() == out
    assert {0, 1, 2} <= seen
    ctx.close()


def test_large_raw_stream_takes_stream_kernel(hip, orc):
    """uncompress of a multi-block raw buffer (snappy.nim:84-110): no block delimiters, so it
    runs on the whole-stream kernel; result identical to the oracle."""
    src = open(os.path.join(os.path.dirname(__file__), "golden", "data", "kppkn.gtb"), "rb").read()
    enc = orc.encode(src)
    assert hip.decode(enc) == src
    # foreign stream with back-references across 64 KiB boundaries (copy4, decoder.nim:103-109)
    lit = bytes(range(256)) * 300  # 76 800 bytes
    stream = bytearray()
    n = len(lit) + 64
    v = n
    while v >= 0x80:
        stream.append((v & 0x7f) | 0x80)
        v >>= 7
    stream.append(v)
    stream += bytes([62 << 2]) + (len(lit) - 1).to_bytes(3, "little")  # 3 length bytes
    stream += lit
    stream += bytes([(63 << 2) | 3]) + (70000).to_bytes(4, "little")  # copy4 len 64 off 70000
    want = orc.decode(bytes(stream))
    assert len(want) == n
    assert hip.decode(bytes(stream)) == want


def test_dense_copy_stream_takes_one_pass_kernel(hip, orc):
    """a chunk with more copy elements than the indexed decoder's list holds (only a foreign
    encoder produces this) is handed to the one-pass kernel; same result as the oracle"""
    n_copies = 3000
    out_len = 4 + 4 * n_copies
    stream = bytearray()
    v = out_len
    while v >= 0x80:
        stream.append((v & 0x7f) | 0x80)
        v >>= 7
    stream.append(v)
    stream += bytes([3 << 2]) + b"abcd"
    stream += bytes([0x01, 0x04]) * n_copies  # copy1: len 4, offset 4
    want = orc.decode(bytes(stream))
    assert want == b"abcd" * (n_copies + 1)
    assert hip.decode(bytes(stream)) == want
    # short copies (len 1..3, copy2 only) and a self-overlapping run, also foreign
    s2 = bytearray([20, 3 << 2]) + b"wxyz" + bytes([(0 << 2) | 2, 4, 0, (1 << 2) | 2, 2, 0,
                                                      (2 << 2) | 2, 1, 0, (9 << 2) | 2, 3, 0])
    assert hip.decode(bytes(s2)) == orc.decode(bytes(s2)) != b""


def test_full_size_round_trip_properties(hip, orc, torch_mod):
    """BASELINE size (65 536 x 64 KiB = 4 GiB; SNAPPY_HIP_TEST_BLOCKS overrides): compress ->
    pack -> decompress on the device restores every byte, all statuses ok, packed offsets
    monotone and consistent with the sizes, framed CRC-of-output equals CRC-of-input; and every packed
    raw unit and the whole framed stream equal the oracle's encoding byte for byte."""
    import corpus
    torch = torch_mod
    nb = int(os.environ.get("SNAPPY_HIP_TEST_BLOCKS", "65536"))
    ctx = hip.Context(0)
    d_in = torch.empty(nb * 65536, dtype=torch.uint8, device="cuda")
    step = 2048
    for b0 in range(0, nb, step):
        c = min(step, nb - b0)
        d_in[b0 * 65536:(b0 + c) * 65536] = corpus.make_blocks_torch(torch, b0, c, "cuda").reshape(-1)
    chk = corpus.make_blocks(nb - 8, 8).reshape(-1)  # (the device generator gives the numpy generator's bytes)
    assert np.array_equal(d_in[(nb - 8) * 65536:].cpu().numpy(), chk)
    d_slots, d_sizes, d_offsets, d_out, total = _encode_pack(hip, torch, ctx, d_in, nb * 65536,
                                                             hip.UNIT_RAW)
    offs = d_offsets.cpu().numpy()
    sizes = d_sizes.cpu().numpy().astype(np.int64)
    assert offs[0] == 0 and (np.diff(offs) == sizes).all()
    assert sizes.max() <= 65536 + 32 + 65536 // 6 and sizes.min() >= 4
    d_out_off = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
    d_out_cap = torch.full((nb,), 65536, dtype=torch.int32, device="cuda")
    d_out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_status = torch.full((nb,), 77, dtype=torch.int32, device="cuda")
    d_crc = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_crc_in = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_dec = torch.zeros(nb * 65536, dtype=torch.uint8, device="cuda")
    ctx.decode_blocks(d_out, d_offsets[:nb].contiguous(), d_sizes, nb, d_dec, d_out_off,
                      d_out_cap, d_out_len, d_status, unit=hip.UNIT_RAW, d_crc=d_crc)
    ctx.crc32c(d_in, d_out_off, d_out_cap, nb, d_crc_in)
    ctx.sync()
    assert int((d_status != 0).sum().item()) == 0
    assert int((d_out_len != 65536).sum().item()) == 0
    assert bool(torch.equal(d_dec, d_in))
    assert bool(torch.equal(d_crc, d_crc_in))
    del d_dec, d_slots
    # BASELINE configs[2] and [3] at FULL size, byte for byte: the oracle encodes every block on all host threads
    # (ranges of 8 192 blocks bound the host memory) -- sizes, the packed raw units and the ONE framed stream
    import hashlib
    nthr = os.cpu_count() or 1
    fcap = hip.max_compressed_len_framed(nb * 65536)
    d_fstream = torch.empty(fcap, dtype=torch.uint8, device="cuda")
    flen = ctx.compress_framed(d_in, nb * 65536, d_fstream, fcap)
    assert bytes(d_fstream[:10].cpu().numpy().tobytes()) == cases.FRAMING_HEADER
    slot = hip.SLOT_STRIDE
    f_at = 10
    sha_gpu, sha_orc = hashlib.sha256(), hashlib.sha256()
    for b0 in range(0, nb, 8192):
        c = min(8192, nb - b0)
        src = d_in[b0 * 65536:(b0 + c) * 65536].cpu().numpy()
        buf = np.empty(c * slot, dtype=np.uint8)
        csz = np.empty(c, dtype=np.uint32)
        orc.lib.sor_compress_blocks_mt(src.ctypes.data, src.size, 65536, buf.ctypes.data, slot, csz.ctypes.data, nthr)
        assert np.array_equal(csz.astype(np.int64), sizes[b0:b0 + c]), b0
        want = np.concatenate([buf[i * slot:i * slot + int(csz[i])] for i in range(c)])
        got = d_out[int(offs[b0]):int(offs[b0 + c])].cpu().numpy()
        assert np.array_equal(got, want), b0
        sha_gpu.update(got.tobytes())
        sha_orc.update(want.tobytes())
        orc.lib.sor_encode_frames_mt(src.ctypes.data, src.size, 65536, buf.ctypes.data, slot, csz.ctypes.data, nthr)
        want = np.concatenate([buf[i * slot:i * slot + int(csz[i])] for i in range(c)])
        assert f_at + want.size <= flen, b0
        assert np.array_equal(d_fstream[f_at:f_at + want.size].cpu().numpy(), want), b0
        f_at += want.size
    assert f_at == flen
    assert sha_gpu.hexdigest() == sha_orc.hexdigest()
    # ... and uncompressFramed (snappy.nim:169-267) of that ONE stream at full size: (read, written) = (all, all), every
    # byte back, with the CRCs checked; a stream with one CRC byte flipped in its last chunk is a crcMismatch, and an
    # output buffer one byte short stops at the last chunk's header (the resume contract, test_framed.nim:38-59)
    del d_out
    d_back = torch.empty(nb * 65536, dtype=torch.uint8, device="cuda")
    assert ctx.uncompress_framed(d_fstream, flen, d_back, nb * 65536) == (0, flen, nb * 65536)
    assert bool(torch.equal(d_back, d_in))
    last_hdr = flen - int(csz[c - 1])  # (the last chunk: 4 bytes of header + its declared length)
    st, rd, wr = ctx.uncompress_framed(d_fstream, flen, d_back, nb * 65536 - 1)
    assert (st, rd, wr) == (0, last_hdr, (nb - 1) * 65536), (st, rd, wr, last_hdr)
    d_fstream[last_hdr + 4] ^= 0x40
    assert ctx.uncompress_framed(d_fstream, flen, d_back, nb * 65536)[0] == 3  # crcMismatch
    ctx.close()


def _structured_block(rng, n=65536):
    """LZ-friendly fuzz: random bytes, repeats of earlier slices at all kinds of distances and
    lengths, periodic runs, constant and ramp stretches -- the shapes that drive the decoder's
    special paths (long literals, same-offset runs, self-overlapping and near copies)."""
    out = bytearray()
    while len(out) < n:
        k = rng.random()
        if k < 0.25 or len(out) < 8:
            out += rng.randbytes(rng.choice([1, 2, 3, 7, 20, 61, 300, 2000]))
        elif k < 0.65:
            off = rng.choice([1, 2, 3, 4, 7, 8, 10, 44, 63, 64, 65, 255, 256, 257, 1000, 2047, 2048,
                              rng.randint(1, len(out))])
            off = min(off, len(out))
            ln = rng.choice([4, 5, 11, 12, 17, 63, 64, 65, 67, 68, 69, 130, 300, 1000, 5000])
            for _ in range(ln):
                out.append(out[-off])
        elif k < 0.85:
            p = rng.choice([1, 2, 3, 5, 10, 44, 100, 255, 256, 257, 300, 777])
            unit = rng.randbytes(p)
            ln = rng.choice([50, 300, 1000, 4000, 20000])
            out += (unit * (ln // p + 1))[:ln]
        elif k < 0.93:
            out += bytes(rng.choice([100, 1000, 9000]))
        else:
            s0 = rng.randrange(256)
            out += bytes((s0 + i) & 255 for i in range(rng.choice([300, 3000])))
    return bytes(out[:n])


def test_structured_fuzz_round_trip(hip, orc, torch_mod):
    """decode(encode(x)) == x and encode(x) == oracle(x) over structured random blocks"""
    torch = torch_mod
    rng = random.Random(99)
    nb = 384
    blocks = [_structured_block(rng) for _ in range(nb)]
    flat = np.frombuffer(b"".join(blocks), dtype=np.uint8)
    ctx = hip.Context(0)
    d_in = _dev(torch, flat)
    d_slots, d_sizes, d_offsets, d_out, total = _encode_pack(hip, torch, ctx, d_in, flat.size, hip.UNIT_RAW)
    sizes = d_sizes.cpu().numpy()
    packed = d_out.cpu().numpy()
    offs = d_offsets.cpu().numpy()
    for i in range(0, nb, 3):  # the GPU encoder against the oracle
        want = orc.encode(blocks[i])
        assert packed[offs[i]:offs[i] + sizes[i]].tobytes() == want, i
    d_out_off = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
    d_out_cap = torch.full((nb,), 65536, dtype=torch.int32, device="cuda")
    d_out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_status = torch.full((nb,), 77, dtype=torch.int32, device="cuda")
    d_dec = torch.zeros(nb * 65536, dtype=torch.uint8, device="cuda")
    ctx.decode_blocks(d_out, d_offsets[:nb].contiguous(), d_sizes, nb, d_dec, d_out_off, d_out_cap,
                      d_out_len, d_status, unit=hip.UNIT_RAW)
    ctx.sync()
    assert int((d_status != 0).sum().item()) == 0
    dec = d_dec.cpu().numpy()
    bad = [i for i in range(nb) if dec[i * 65536:(i + 1) * 65536].tobytes() != blocks[i]]
    assert not bad, bad[:8]
    ctx.close()


def test_bench_contract_and_two_rank_path(hip):
    """bench.py prints ONE JSON line with the contract's keys; its N > 1 path (block-range shards,
    barrier, MAX over ranks) is exercised with two ranks that share this box's GPU over gloo
    (BENCH_SHARE_DEVICE / BENCH_DIST_BACKEND are test hooks: RCCL refuses two ranks on one device)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--blocks", "1024", "--steps", "2", "--warmup", "1", "--no-cpu"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True,
                         text=True, timeout=600)
    lines = [ln for ln in one.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, one.stdout + one.stderr
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["dtype"] == "u8" and j["vs_baseline"] is None and j["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_step", "traffic_step",
                "traffic_over_algorithmic")) <= set(j["roofline"])
    assert 0 < j["roofline"]["frac_step"] < 1
    # the side numbers live INSIDE `roofline` (the driver's record keeps that dict whole): value + fraction of the HBM peak
    for k in ("decompress_step", "compress", "round_trip", "framed", "framed_compress"):
        assert j["roofline"][k]["value"] > 0 and 0 < j["roofline"][k]["frac"] < 1, k
    assert j["roofline"]["framed"]["stream_bytes"] > 10 and j["roofline"]["framed"]["calls"] == 2
    assert abs(j["roofline"]["decompress_step"]["value"] - j["value"]) < 0.01 * j["value"]
    assert j["framed_decompress_calls"] == 2 and j["framed_decompress_best_GBps"] >= j["framed_decompress_GBps"] > 0
    assert j["library"] == {"path": os.path.join("nim-snappy_amd", "libsnappy_hip.so"), "overridden": False}
    # a variant library is refused unless asked for
    ref = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, SNAPPY_HIP_LIBRARY=os.path.join(root, "nim-snappy_amd", "libsnappy_hip.so")))
    assert ref.returncode == 2 and "SNAPPY_HIP_LIBRARY" in ref.stderr
    assert j["roofline_compress"]["frac"] > 0 and j["sharded_compress"]["n_shards"] == 1
    assert j["sharded_compress"]["equals_single_gpu_sha256"] is True
    # --gpus 2 with NO launcher on the command line: bench.py starts its two ranks itself
    env = dict(os.environ, BENCH_SHARE_DEVICE="1", BENCH_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                         capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    j2 = json.loads(lines[0])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "weak" and j2["value"] > 0
    sc = j2["sharded_compress"]
    assert sc["n_shards"] == 2 and len(sc["shard_bytes"]) == 2 and sc["equals_single_gpu_sha256"] is True
    # the same fixed total from one and from two shards: the same stream
    assert sc["total_blocks"] == j["sharded_compress"]["total_blocks"]
    assert sc["stream_sha256_tree64MiB"] == j["sharded_compress"]["stream_sha256_tree64MiB"]
    assert sc["stream_bytes"] == j["sharded_compress"]["stream_bytes"]
    # under an external launcher the flag must agree with WORLD_SIZE
    env3 = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                         capture_output=True, text=True, timeout=300, env=env3)
    assert bad.returncode != 0


def test_bench_eight_ranks_share_the_gpu(hip):
    """the shape of the driver's 8-GPU run on a one-GPU box: eight ranks (gloo, all on this GPU), 128 blocks each; the
    sharded compress of the same fixed total (1 024 blocks, stages going round eight ranks) gives the stream one rank
    gives, digest for digest"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--blocks", "128", "--steps", "1", "--warmup", "1", "--no-cpu"]
    env = dict(os.environ, BENCH_SHARE_DEVICE="1", BENCH_DIST_BACKEND="gloo", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    outs = {}
    for n in (1, 8):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n)] + common,
                           capture_output=True, text=True, timeout=1200, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[n] = json.loads(lines[0])
    assert outs[8]["n_gpus"] == 8 and outs[8]["value"] > 0
    a, b = outs[1]["sharded_compress"], outs[8]["sharded_compress"]
    assert b["n_shards"] == 8 and len(b["shard_bytes"]) == 8 and b["equals_single_gpu_sha256"] is True
    assert a["total_blocks"] == b["total_blocks"] == 1024
    assert a["stream_sha256_tree64MiB"] == b["stream_sha256_tree64MiB"] and a["stream_bytes"] == b["stream_bytes"]


def test_sharded_compress_tool_two_ranks(hip):
    """tools/sharded_compress.py (BASELINE configs[4]): the block-range shards of two ranks, laid end to
    end at the scanned offsets, are the same bytes as one rank's encoding of the whole input"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "sharded_compress.py")
    one = subprocess.run([sys.executable, tool, "--blocks-per-gpu", "512", "--verify"], capture_output=True,
                         text=True, timeout=600)
    j1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    env = dict(os.environ, BENCH_SHARE_DEVICE="1", BENCH_DIST_BACKEND="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", tool,
                          "--blocks-per-gpu", "256", "--verify"], capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert lines, two.stdout[-1500:] + two.stderr[-1500:]
    j2 = json.loads(lines[-1])
    assert j1["verified"] is True and j2["verified"] is True
    assert j2["n_gpus"] == 2 and len(j2["shard_bytes"]) == 2
    assert j2["uncompressed_bytes"] == j1["uncompressed_bytes"]
    assert j2["compressed_bytes"] == j1["compressed_bytes"] and sum(j2["shard_bytes"]) == j1["shard_bytes"][0]


def test_decode_crc_ragged_units(hip, orc, torch_mod):
    """The CRC that comes out of the decode kernel's flush, for unit lengths of every residue mod 4
    and mod 1024, misaligned output offsets, an empty unit and a unit that errors: each == oracle
    masked CRC of what the unit produced (maskedCrc, codec.nim:71-75)."""
    import corpus
    torch = torch_mod
    rng = np.random.default_rng(4242)
    text = corpus.make_blocks(0, 4, only="T_TEXT").reshape(-1)
    lens = [1, 2, 3, 4, 5, 7, 8, 1023, 1024, 1025, 1027, 2048, 4097, 30001, 65533, 65534, 65535, 65536]
    lens += [int(x) for x in rng.integers(1, 65537, size=14)]
    plains = []
    for i, n in enumerate(lens):
        s = int(rng.integers(0, text.size - n))
        plains.append(text[s:s + n].tobytes() if i % 3 else bytes(rng.integers(0, 256, n, dtype=np.uint8)))
    units = [orc.encode(p) for p in plains]
    units.append(orc.encode(b""))                      # empty: varint 0 only
    plains.append(b"")
    bad = bytearray(orc.encode(plains[9]))
    bad[3] = 0xff; bad[4] = 0xff                        # breaks the element chain
    units.append(bytes(bad))
    plains.append(None)
    nu = len(units)
    blob = b"".join(units)
    in_off = np.cumsum([0] + [len(u) for u in units[:-1]]).astype(np.int64)
    in_len = np.array([len(u) for u in units], dtype=np.int32)
    # outputs packed back to back (so most start misaligned)
    caps = np.array([len(p) if p is not None else len(plains[9]) for p in plains], dtype=np.int32)
    out_off = np.cumsum([0] + [int(c) + 3 for c in caps[:-1]]).astype(np.int64)
    ctx = hip.Context(0)
    d_in = _dev(torch, np.frombuffer(blob + b"\0" * 64, dtype=np.uint8))
    d_out = torch.zeros(int(out_off[-1]) + int(caps[-1]) + 64, dtype=torch.uint8, device="cuda")
    d_out_len = torch.zeros(nu, dtype=torch.int32, device="cuda")
    d_status = torch.full((nu,), 77, dtype=torch.int32, device="cuda")
    d_crc = torch.zeros(nu, dtype=torch.int32, device="cuda")
    ctx.decode_blocks(d_in, _dev(torch, in_off), _dev(torch, in_len), nu, d_out, _dev(torch, out_off),
                      _dev(torch, caps), d_out_len, d_status, unit=hip.UNIT_RAW, d_crc=d_crc)
    ctx.sync()
    st = d_status.cpu().numpy()
    ol = d_out_len.cpu().numpy()
    crcs = d_crc.cpu().numpy().view(np.uint32)
    out = d_out.cpu().numpy()
    for i, p in enumerate(plains):
        if p is None:
            assert st[i] != 0
            continue
        assert st[i] == 0 and ol[i] == len(p), (i, st[i], ol[i], len(p))
        assert out[out_off[i]:out_off[i] + len(p)].tobytes() == p, i
        assert int(crcs[i]) == orc.masked_crc(p), (i, len(p))
    ctx.close()


def test_launch_order_large_batch(hip, orc, torch_mod):
    """3 000 units of very different compressed lengths (the library launches them sorted by length,
    crc_pack_kernels.h): every unit lands in its own output range, statuses and lengths are per
    unit, CRCs == oracle; the same batch with the ordering switched off gives the same bytes."""
    import corpus
    torch = torch_mod
    rng = np.random.default_rng(77)
    nb = 3000
    blocks = corpus.make_blocks(5000, nb)
    flat = blocks.reshape(-1)
    ctx = hip.Context(0)
    d_in = _dev(torch, flat)
    d_slots, d_sizes, d_offsets, d_packed, total = _encode_pack(hip, torch, ctx, d_in, flat.size, hip.UNIT_RAW)
    stream = bytearray(d_packed[:total].cpu().numpy().tobytes())
    in_off = d_offsets.cpu().numpy()[:nb].astype(np.int64).copy()
    in_len = d_sizes.cpu().numpy().astype(np.int32).copy()
    expect = [blocks[i].tobytes() for i in range(nb)]
    for i in range(64):  # the first 64 units: ragged lengths, encoded by the oracle
        plain = blocks[i][:int(rng.integers(1, 65537))].tobytes()
        u = orc.encode(plain)
        in_off[i] = len(stream)
        in_len[i] = len(u)
        stream += u
        expect[i] = plain
    out_off = np.arange(nb, dtype=np.int64) * 65536
    cap = np.full(nb, 65536, dtype=np.int32)
    d_stream = _dev(torch, np.frombuffer(bytes(stream) + b"\0" * 64, dtype=np.uint8))
    results = []
    for no_order in (False, True):
        ctx.launch_order(not no_order)  # (snappy_hip_ctx_launch_order: sorted launch order / the caller's order)
        try:
            d_out = torch.zeros(nb * 65536, dtype=torch.uint8, device="cuda")
            d_out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
            d_status = torch.full((nb,), 77, dtype=torch.int32, device="cuda")
            d_crc = torch.zeros(nb, dtype=torch.int32, device="cuda")
            ctx.decode_blocks(d_stream, _dev(torch, in_off), _dev(torch, in_len), nb, d_out, _dev(torch, out_off),
                              _dev(torch, cap), d_out_len, d_status, unit=hip.UNIT_RAW, d_crc=d_crc)
            ctx.sync()
        finally:
            ctx.launch_order(True)
        assert (d_status.cpu().numpy() == 0).all()
        ol = d_out_len.cpu().numpy()
        out = d_out.cpu().numpy()
        crcs = d_crc.cpu().numpy().view(np.uint32)
        assert [int(x) for x in ol] == [len(e) for e in expect]
        for i in list(range(64)) + list(range(64, nb, 97)):
            assert out[i * 65536:i * 65536 + ol[i]].tobytes() == expect[i], i
            assert int(crcs[i]) == orc.masked_crc(expect[i]), i
        results.append((out.copy(), crcs.copy()))
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    # and without the CRC buffer (ring-window instantiation first, the whole-block one for what it passes on)
    d_out = torch.zeros(nb * 65536, dtype=torch.uint8, device="cuda")
    d_out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_status = torch.full((nb,), 77, dtype=torch.int32, device="cuda")
    ctx.decode_blocks(d_stream, _dev(torch, in_off), _dev(torch, in_len), nb, d_out, _dev(torch, out_off),
                      _dev(torch, cap), d_out_len, d_status, unit=hip.UNIT_RAW)
    ctx.sync()
    assert (d_status.cpu().numpy() == 0).all()
    assert np.array_equal(d_out.cpu().numpy(), results[0][0])
    ctx.close()


def test_units_flush_against_exact_size_allocations(hip, orc):
    """include/snappy_hip.h: no padding is required -- a unit may start at the first byte and end at
    the last byte of a hipMalloc of exactly its size (the kernels' 16-byte loads are aligned and stay
    inside the lines that hold the unit's ends).  Input, slots, packed stream and output are
    hipMalloc'ed at their exact sizes through the HIP runtime, no torch allocator in between."""
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    rt.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    rt.hipFree.argtypes = [ctypes.c_void_p]

    def dmalloc(n):
        p = ctypes.c_void_p()
        assert rt.hipMalloc(ctypes.byref(p), max(n, 1)) == 0
        return p.value

    def up(data):
        p = dmalloc(len(data))
        assert rt.hipMemcpy(p, bytes(data), len(data), 1) == 0
        return p

    def down(p, n):
        b = ctypes.create_string_buffer(max(n, 1))
        assert rt.hipMemcpy(b, p, n, 2) == 0
        return b.raw[:n]

    ctx = hip.Context(0)
    src = golden_text = open(os.path.join(os.path.dirname(__file__), "golden", "data", "alice29.txt"), "rb").read()
    for n in (152089, 65536, 65537, 4099, 17, 1):
        data = src[:n]
        nb = (n + 65535) // 65536
        d_in = up(data)
        d_slots = dmalloc(nb * hip.SLOT_STRIDE)
        d_sizes = dmalloc(nb * 4)
        ctx.encode_blocks(d_in, n, d_slots, d_sizes)
        ctx.sync()
        sizes = np.frombuffer(down(d_sizes, nb * 4), dtype=np.uint32)
        total = int(sizes.sum())
        d_packed = dmalloc(total)
        d_offsets = dmalloc((nb + 1) * 8)
        ctx.pack(d_slots, d_sizes, nb, d_packed, d_offsets)
        ctx.sync()
        packed = down(d_packed, total)
        want = b"".join(orc.encode(data[i:i + 65536]) for i in range(0, n, 65536))
        assert packed == want, n
        offs = np.frombuffer(down(d_offsets, (nb + 1) * 8), dtype=np.int64)
        d_out = dmalloc(n)
        oo = np.arange(nb, dtype=np.int64) * 65536
        oc = np.minimum(65536, n - oo).astype(np.uint32)
        d_oo, d_oc = up(oo.tobytes()), up(oc.tobytes())
        d_ol, d_st = dmalloc(nb * 4), dmalloc(nb * 4)
        ctx.decode_blocks(d_packed, d_offsets, d_sizes, nb, d_out, d_oo, d_oc, d_ol, d_st)
        ctx.sync()
        assert down(d_out, n) == data, n
        assert not np.frombuffer(down(d_st, nb * 4), dtype=np.uint32).any()
        # the framed stream and the raw buffer, resident, exact sizes
        fr = orc.encode_framed(data)
        d_fr, d_fo = up(fr), dmalloc(n)
        assert ctx.uncompress_framed(d_fr, len(fr), d_fo, n) == (0, len(fr), n) and down(d_fo, n) == data
        raw = orc.encode(data)
        d_raw, d_ro = up(raw), dmalloc(n)
        assert ctx.uncompress(d_raw, len(raw), d_ro, n) == (0, n) and down(d_ro, n) == data
        for p in (d_in, d_slots, d_sizes, d_packed, d_offsets, d_out, d_oo, d_oc, d_ol, d_st, d_fr, d_fo, d_raw, d_ro):
            rt.hipFree(p)
    ctx.close()


def test_shards_from_one_process_land_in_one_host_buffer(hip, orc, torch_mod):
    """BASELINE configs[4], single-process form (snappy_hip_compress_shards): two contexts, block ranges
    encoded separately, the shard totals scanned on the host, every shard downloaded to its offset in
    ONE page-locked buffer -- byte-identical to the oracle's encoding of the whole input, framed and
    raw, with a ragged tail in the last shard"""
    torch = torch_mod
    import json
    import subprocess
    import sys
    import corpus
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    nb = 96
    with open(os.path.join(ROOT, "tests", "golden", "data", "alice29.txt"), "rb") as fh:
        tail = fh.read()[:12345]
    src = corpus.make_blocks(0, nb).tobytes() + tail
    ctxs = [hip.Context(0), hip.Context(0)]
    cut = 64 * 65536
    d_ins = [_dev(torch, np.frombuffer(src[:cut], dtype=np.uint8)), _dev(torch, np.frombuffer(src[cut:], dtype=np.uint8))]
    lens = [cut, len(src) - cut]
    for framed, want in ((True, orc.encode_framed(src)), (False, orc.encode(src))):
        cap = hip.max_compressed_len_framed(len(src)) if framed else hip.max_compressed_len(len(src))
        out = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
        written, offs = hip.compress_shards(ctxs, d_ins, lens, out.data_ptr(), cap, framed=framed)
        assert out[:written].numpy().tobytes() == want
        assert offs[0] in (10, 4) and offs[-1] == written and offs[0] < offs[1] < offs[2]
    # one context cannot serve two shards at once (it owns the scratch buffers its calls use)
    with pytest.raises(ValueError):
        hip.compress_shards([ctxs[0], ctxs[0]], d_ins, lens, out.data_ptr(), cap, framed=False)
    # the same in stages (snappy_hip_compress_shards_staged): stage j of context k is the global block range
    # [(2 j + k) S, (2 j + k + 1) S), a context's stages lie back to back in its input, and a context downloads a
    # stage while it encodes the next -- byte-identical again, whole stages, a short last stage, a ragged tail
    B = 65536
    for S in (16, 40, 7):
        parts = [bytearray(), bytearray()]
        for i, at in enumerate(range(0, len(src), S * B)):
            parts[i % 2] += src[at:at + S * B]
        d_st = [_dev(torch, np.frombuffer(bytes(q), dtype=np.uint8)) for q in parts]
        for framed, want in ((True, orc.encode_framed(src)), (False, orc.encode(src))):
            cap = hip.max_compressed_len_framed(len(src)) if framed else hip.max_compressed_len(len(src))
            out = torch.zeros(cap, dtype=torch.uint8, pin_memory=True)
            written, offs = hip.compress_shards(ctxs, d_st, [len(q) for q in parts], out.data_ptr(), cap, framed=framed,
                                                stage_blocks=S)
            assert out[:written].numpy().tobytes() == want, (S, framed)
            assert offs[-1] == written
        with pytest.raises(ValueError):  # lengths that are not what the stage layout gives the contexts
            hip.compress_shards(ctxs, d_ins, lens, out.data_ptr(), cap, framed=False, stage_blocks=S)
    # EIGHT contexts (one per GPU of a node; here all on this box's GPU): stages of 5 blocks go round them, the n stage
    # sizes are exchanged under the mutex, every context downloads beside its next stage -- the oracle's bytes
    ctx8 = ctxs + [hip.Context(0) for _ in range(6)]
    for S in (5, 1):
        parts = [bytearray() for _ in range(8)]
        for i, at in enumerate(range(0, len(src), S * B)):
            parts[i % 8] += src[at:at + S * B]
        d_st = [_dev(torch, np.frombuffer(bytes(q) if q else b"\0", dtype=np.uint8)) for q in parts]
        cap = hip.max_compressed_len_framed(len(src))
        out = torch.zeros(cap, dtype=torch.uint8, pin_memory=True)
        written, offs = hip.compress_shards(ctx8, d_st, [len(q) for q in parts], out.data_ptr(), cap, framed=True, stage_blocks=S)
        assert out[:written].numpy().tobytes() == orc.encode_framed(src), S
        assert len(offs) == 9 and offs[-1] == written
    for c in ctx8[2:]:
        c.close()
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shards_one_process.py"), "--gpus", "2", "--same-gpu",
                         "--total-gib", "0.25", "--check", "--reps", "1"], capture_output=True, text=True, timeout=600)
    assert rc.returncode == 0, rc.stderr[-2000:]
    assert json.loads(rc.stdout.strip().splitlines()[-1])["equals_oracle"] is True
    for c in ctxs:
        c.close()


def _ring_stream(rng, target, style):
    """One block's tag stream for the ring-window decoder (decode2_kernel.h): short literals and copies at
    a text-like ratio (so that three steps' output fits the ring), with copies whose sources lie further
    back than the ring holds, long literals, runs of same-offset copies, and -- style 'dense' -- stretches
    of pure copies that make two steps' output outgrow the ring in the middle of the block (the unit is passed on),
    style 'mid' -- stretches at ~7 KiB of output a step: three steps outgrow a ring of 16 KiB, two do not (catch-up
    flushes)."""
    out, body = bytearray(), bytearray()

    def lit(n):
        data = rng.randbytes(n)
        m = n - 1
        if m < 60:
            body.append(m << 2)
        else:
            ll = max((m.bit_length() + 7) // 8, 1)
            body.extend(bytes([(59 + ll) << 2]) + m.to_bytes(ll, "little"))
        body.extend(data)
        out.extend(data)

    def copy(off, ln):
        if 4 <= ln <= 11 and off < 2048 and rng.random() < 0.5:
            body.extend(bytes([((off >> 8) << 5) | ((ln - 4) << 2) | 1, off & 0xff]))
        else:
            body.extend(bytes([((ln - 1) << 2) | 2]) + off.to_bytes(2, "little"))
        for _ in range(ln):
            out.append(out[-off])

    lit(rng.randint(1, 60))
    while len(out) < target:
        r = rng.random()
        room = target - len(out)
        if style == "dense" and 20000 < len(out) < 45000:
            copy(rng.randint(1, min(len(out), 65535)), min(room, 64))
        elif style == "mid" and 12000 < len(out) < 50000:  # ~7 KiB of output a step: three steps outgrow 16 KiB, two do not
            copy(rng.randint(1, min(len(out), 65535)), min(room, rng.randint(12, 20)))
            if rng.random() < 0.5:
                lit(min(target - len(out), rng.randint(1, 3)) or 1) if len(out) < target else None
        elif r < 0.45:
            lit(min(room, rng.randint(1, 40)))
        elif r < 0.47:
            lit(min(room, rng.choice([61, 100, 700, 3000, 9000])))
        elif r < 0.472 and style == "biglit":
            lit(min(room, rng.randint(15000, 40000)))
        elif r < 0.49:  # a run: one offset, copy after copy (also offsets from before the ring)
            off = rng.choice([1, 2, 7, 44, 300, 5000, min(len(out), 40000)])
            off = min(off, len(out))
            for _ in range(rng.randint(3, 120)):
                if len(out) >= target:
                    break
                copy(off, min(target - len(out), 64))
        else:
            far = rng.random() < 0.25
            off = rng.randint(min(len(out), 14000), min(len(out), 65535)) if far else rng.randint(1, min(len(out), 3000))
            copy(off, min(room, rng.randint(4, 40)))
    return bytes(body), bytes(out)


def test_ring_window_decoder(hip, orc, torch_mod):
    """the indexed decoder's two instantiations (ring of the last 16 KiB first, whole block for the units it
    passes on), without and with the CRC coming out of the decode kernels (the ring instantiation checksums
    the rows its flush completes): foreign streams whose copies reach behind the ring, wrap it, run across
    its end; outputs at unaligned addresses; output lengths of every residue mod 4 and mod 1024"""
    torch = torch_mod
    rng = random.Random(99)
    units = []
    for i in range(240):
        style = ("text", "biglit", "dense", "mid")[i % 4]
        target = 65536 if i % 3 else rng.randint(33000, 65536)
        body, plain = _ring_stream(rng, target, style)
        units.append((body, plain))
    for b, p in units[:12]:
        assert orc.decode_all_tags(b, len(p)) == (0, p)
    n = len(units)
    # more units than one wave of workgroups: copies of the same streams, different placements
    reps = 6
    in_off, out_off, pos, opos = [], [], 0, 0
    for r in range(reps):
        for k, (b, p) in enumerate(units):
            in_off.append(pos)
            pos += len(b) + rng.randint(0, 5)
            opos += 0 if (k + r) % 5 == 0 else rng.randint(1, 15)  # (16-byte aligned or not)
            if (k + r) % 5 == 0:
                opos = (opos + 15) & ~15
            out_off.append(opos)
            opos += len(p)
    stream = np.zeros(pos + 64, np.uint8)
    for j, o in enumerate(in_off):
        b = units[j % n][0]
        stream[o:o + len(b)] = np.frombuffer(b, np.uint8)
    nu = n * reps
    ctx = hip.Context(0)
    d_stream = _dev(torch, stream)
    d_in_off = _dev(torch, np.array(in_off, np.int64))
    d_in_len = _dev(torch, np.array([len(units[j % n][0]) for j in range(nu)], np.int32))
    d_out_off = _dev(torch, np.array(out_off, np.int64))
    d_out_cap = _dev(torch, np.array([len(units[j % n][1]) for j in range(nu)], np.int32))
    want = np.zeros(opos, np.uint8)
    for j, o in enumerate(out_off):
        p = units[j % n][1]
        want[o:o + len(p)] = np.frombuffer(p, np.uint8)
    for with_crc in (False, True):
        d_out_len = torch.zeros(nu, dtype=torch.int32, device="cuda")
        d_status = torch.full((nu,), 77, dtype=torch.int32, device="cuda")
        d_dec = torch.zeros(opos, dtype=torch.uint8, device="cuda")
        d_crc = torch.zeros(nu, dtype=torch.int32, device="cuda") if with_crc else None
        ctx.decode_blocks(d_stream, d_in_off, d_in_len, nu, d_dec, d_out_off, d_out_cap, d_out_len, d_status,
                          unit=hip.UNIT_BODY, d_crc=d_crc)
        ctx.sync()
        st = d_status.cpu().numpy()
        assert (st == 0).all(), (with_crc, np.nonzero(st)[0][:10], st[st != 0][:10])
        got = d_dec.cpu().numpy()
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (with_crc, bad[:10], np.searchsorted(np.array(out_off), bad[:3], side="right") - 1)
        if with_crc:
            crcs = d_crc.cpu().numpy().view(np.uint32)
            want_crc = [orc.masked_crc(p) for _, p in units]
            bad_crc = [j for j in range(nu) if int(crcs[j]) != want_crc[j % n]]
            assert not bad_crc, bad_crc[:10]


def test_pool_release_and_caller_stream(hip, orc, torch_mod):
    """snappy_hip_release_pool() frees the idle pooled contexts (calls after it make new ones); and
    snappy_hip_uncompress_d on a CALLER's stream still splits a multi-block raw buffer on the device (the
    result is the oracle's either way; the split is seen in the raw-split timer slot)"""
    import corpus
    torch = torch_mod
    src = corpus.make_blocks(3, 40).tobytes()
    enc = hip.encode(src)
    assert enc == orc.encode(src)
    hip.release_pool()
    hip.release_pool()  # (nothing idle: a no-op)
    assert hip.decode(enc) == src and hip.encode(src[:70000]) == orc.encode(src[:70000])
    ctx = hip.Context(0)
    d_raw = _dev(torch, np.frombuffer(enc, dtype=np.uint8))
    d_out = torch.zeros(len(src), dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    ctx.timing(True)
    assert ctx.uncompress(d_raw, len(enc), d_out, len(src), stream=st.cuda_stream) == (0, len(src))
    st.synchronize()
    ms, launches = ctx.kernel_ms(7)  # raw-buffer split rounds
    ctx.timing(False)
    assert launches > 0, "the caller's stream took the serial whole-stream kernel"
    assert d_out.cpu().numpy().tobytes() == src
    ctx.close()


def test_one_context_on_two_streams_back_to_back(hip, orc, torch_mod):
    """include/snappy_hip.h: one stream per context at a time.  A caller that hands a context to another stream without a
    synchronisation in between used to get two launches sharing the context's workspace -- the encoder's work-queue counter
    among it: skipped blocks, stale sizes.  Since round 6 a call on another stream than the call before it first waits for
    that call's work (an event), so the results are the oracle's whatever the streams."""
    import corpus
    torch = torch_mod
    nb = 2304  # (above the encoder's second-wave threshold: the queue and the tables in global memory are in use)
    a = corpus.make_blocks(11, nb)
    b = corpus.make_blocks(5011, nb)
    ctx = hip.Context(0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d = []
    for arr in (a, b):
        d_in = _dev(torch, arr.reshape(-1))
        d.append((d_in, torch.zeros(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device="cuda"),
                  torch.zeros(nb, dtype=torch.int32, device="cuda")))
    torch.cuda.synchronize()
    for rep in range(3):  # enqueued back to back, nothing waited for in between
        ctx.encode_blocks(d[0][0], nb * 65536, d[0][1], d[0][2], stream=s1.cuda_stream)
        ctx.encode_blocks(d[1][0], nb * 65536, d[1][1], d[1][2], stream=s2.cuda_stream)
    s1.synchronize()
    s2.synchronize()
    for arr, (d_in, d_slots, d_sizes) in zip((a, b), d):
        sizes = d_sizes.cpu().numpy()
        slots = d_slots.cpu().numpy().reshape(nb, hip.SLOT_STRIDE)
        for i in list(range(0, nb, 97)) + [nb - 1]:
            assert slots[i, :sizes[i]].tobytes() == orc.encode(arr[i].tobytes()), i
        assert int((sizes <= 0).sum()) == 0
    ctx.close()


def _sparse_stream(rng, style, target=65536):
    """One block's tag stream of FEW, LONG elements (sparse_kernel.h): long literals with stretches of copies behind
    them -- the repeated-strings pattern -- plus everything that kernel treats specially: literals of every length
    class (1..60, length bytes, above and below its whole-wave threshold), copy1 / copy2 / copy4, copies that overlap
    themselves (offset < length), copies whose source spans several elements, chains of copies through copies
    (levels), more long literals than its list holds, and -- style 'deep' -- more levels than it goes."""
    out, body = bytearray(), bytearray()

    def lit(n):
        data = rng.randbytes(n)
        m = n - 1
        if m < 60:
            body.append(m << 2)
        else:
            ll = max((m.bit_length() + 7) // 8, 1)
            body.extend(bytes([(59 + ll) << 2]) + m.to_bytes(ll, "little"))
        body.extend(data)
        out.extend(data)

    def copy(off, ln, form=None):
        form = form or (1 if 4 <= ln <= 11 and off < 2048 and rng.random() < 0.5 else 2)
        if form == 1:
            body.extend(bytes([((off >> 8) << 5) | ((ln - 4) << 2) | 1, off & 0xff]))
        elif form == 2:
            body.extend(bytes([((ln - 1) << 2) | 2]) + off.to_bytes(2, "little"))
        else:
            body.extend(bytes([((ln - 1) << 2) | 3]) + off.to_bytes(4, "little"))
        for _ in range(ln):
            out.append(out[-off])

    def run(off, total):
        while total > 0 and len(out) < target:
            ln = min(total, 64, target - len(out))
            copy(off, ln, 2)
            total -= ln

    while len(out) < target:
        room = target - len(out)
        r = rng.random()
        if style == "deep" and len(out) > 3000:  # every copy reads the copy in front of it: as many levels as copies
            copy(64, min(room, 64), 2)
            continue
        if style == "manylits" and len(out) < 40000:
            lit(min(room, rng.randint(256, 300)))  # ~140 literals above the whole-wave threshold (its list holds 64)
            continue
        if r < 0.5 or not out:
            n = min(room, rng.choice([1, 17, 60, 61, 200, 255, 256, 257, 1000, 4000, 9000]))
            lit(n)
            if len(out) < target and rng.random() < 0.8:  # ... repeated behind itself, one to three times
                run(min(n, len(out)), min(target - len(out), n * rng.randint(1, 3)))
        elif r < 0.6:
            copy(rng.randint(1, min(len(out), 63)), min(room, rng.randint(2, 64)), 2)          # overlaps itself (or not)
        elif r < 0.7:
            copy(rng.randint(1, min(len(out), 65535)), min(room, rng.randint(1, 64)), 3)       # copy4
        elif r < 0.8:
            copy(rng.randint(1, min(len(out), 2047)), min(room, rng.randint(4, 11)), 1)        # copy1
        else:
            off = rng.randint(1, min(len(out), 65535))
            run(off, min(room, rng.randint(64, 3000)))                                         # a stretch at any distance
    return bytes(body), bytes(out)


def test_batches_on_both_sides_of_the_team_index_pass(hip, orc, torch_mod):
    """decode_blocks of 1 024 units goes through the index pass's team of four waves a unit (index_kernel.h, TEAM), of 1 025
    through one wave a unit: the same units, the same bytes, statuses, lengths and CRCs either way (damaged and ragged
    units in small batches: the mutation tests of this file, which all run through the team now)"""
    import corpus
    torch = torch_mod
    nb = 1025
    blocks = corpus.make_blocks(7, nb)
    flat = blocks.reshape(-1)
    ctx = hip.Context(0)
    d_in = _dev(torch, flat)
    d_slots, d_sizes, d_offsets, d_packed, total = _encode_pack(hip, torch, ctx, d_in, flat.size, hip.UNIT_RAW)
    d_in_off = d_offsets[:nb].contiguous()
    d_out_off = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
    d_out_cap = torch.full((nb,), 65536, dtype=torch.int32, device="cuda")
    results = []
    for n in (1024, 1025):
        d_out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
        d_status = torch.full((nb,), 77, dtype=torch.int32, device="cuda")
        d_crc = torch.zeros(nb, dtype=torch.int32, device="cuda")
        d_dec = torch.zeros(nb * 65536, dtype=torch.uint8, device="cuda")
        before = ctx.kernel_ms(10)[1]
        ctx.decode_blocks(d_packed, d_in_off, d_sizes, n, d_dec, d_out_off, d_out_cap, d_out_len, d_status, unit=hip.UNIT_RAW,
                          d_crc=d_crc)
        ctx.sync()
        assert not d_status[:n].cpu().numpy().any()
        assert (d_out_len[:n].cpu().numpy() == 65536).all()
        assert torch.equal(d_dec[:n * 65536], d_in[:n * 65536])
        results.append((d_crc[:1024].cpu().numpy().copy(), ctx.kernel_ms(10)[1] - before))
    assert (results[0][0] == results[1][0]).all()
    for j in range(0, 1024, 97):
        assert int(results[0][0][j]) & 0xffffffff == orc.masked_crc(blocks[j].tobytes()), j
    # (both write the one-literal and one-period units themselves; only one wave a unit also the units of few long elements)
    assert 0 < results[0][1] < results[1][1], results
    ctx.close()


def test_sparse_units_decoder(hip, orc, torch_mod):
    """units of few, long elements are decoded element-parallel by the index pass's own waves (sparse_kernel.h: units of 2
    to 832 elements whose stream is longer than 4 KiB): the repeated-strings pattern and every element form that
    kernel treats specially, outputs at unaligned addresses, with and without the CRC; what it does not do (more
    levels than it goes) comes out right through the indexed decoder; a bad offset is refused like the oracle refuses it"""
    torch = torch_mod
    rng = random.Random(4242)
    units = []
    for i in range(96):
        style = ("rs", "rs", "manylits", "deep", "rs", "rs")[i % 6]
        target = 65536 if i % 4 else rng.randint(9000, 65536)
        units.append(_sparse_stream(rng, style, target))
    # bad offsets: a copy that reaches in front of the output (decoder.nim:112), at the start and in the middle
    bad_units = []
    for k in range(6):
        b, p = units[k]
        bb = bytearray(b)
        if k % 2 == 0:
            bb = bytearray(bytes([(63 << 2) | 2, 0xff, 0xff])) + bb  # copy2 of 64 bytes at output position 0
        else:
            bb += bytes([(3 << 2) | 2, 0, 0])  # offset 0
        bad_units.append(bytes(bb))
    n_el = []
    for b, p in units:
        assert orc.decode_all_tags(b, len(p)) == (0, p)
    # (eleven times over: a batch of more than 1 024 units -- a smaller one goes through the index pass's team of four
    # waves a unit, index_kernel.h TEAM, which leaves units of few long elements to the indexed decoder)
    all_units = [(b, p, True) for b, p in units] * 11 + [(b, b"", False) for b in bad_units]
    nu = len(all_units)
    assert nu > 1024
    in_off, out_off, pos, opos = [], [], 0, 0
    for k, (b, p, ok) in enumerate(all_units):
        in_off.append(pos)
        pos += len(b) + rng.randint(0, 7)
        opos += 0 if k % 3 == 0 else rng.randint(1, 15)
        if k % 3 == 0:
            opos = (opos + 15) & ~15
        out_off.append(opos)
        opos += 65536
    stream = np.zeros(pos + 64, np.uint8)
    for j, o in enumerate(in_off):
        stream[o:o + len(all_units[j][0])] = np.frombuffer(all_units[j][0], np.uint8)
    ctx = hip.Context(0)
    d_stream = _dev(torch, stream)
    d_in_off = _dev(torch, np.array(in_off, np.int64))
    d_in_len = _dev(torch, np.array([len(u[0]) for u in all_units], np.int32))
    d_out_off = _dev(torch, np.array(out_off, np.int64))
    d_out_cap = _dev(torch, np.array([65536] * nu, np.int32))
    for with_crc in (False, True):
        d_out_len = torch.zeros(nu, dtype=torch.int32, device="cuda")
        d_status = torch.full((nu,), 77, dtype=torch.int32, device="cuda")
        d_dec = torch.zeros(opos, dtype=torch.uint8, device="cuda")
        d_crc = torch.zeros(nu, dtype=torch.int32, device="cuda") if with_crc else None
        before = ctx.kernel_ms(10)[1]  # (a running count: units the index pass decoded itself)
        ctx.decode_blocks(d_stream, d_in_off, d_in_len, nu, d_dec, d_out_off, d_out_cap, d_out_len, d_status,
                          unit=hip.UNIT_BODY, d_crc=d_crc)
        ctx.sync()
        assert ctx.kernel_ms(10)[1] - before >= 32 * 11  # (most of the 'rs' and 'manylits' units)
        st = d_status.cpu().numpy()
        ol = d_out_len.cpu().numpy()
        got = d_dec.cpu().numpy()
        crcs = d_crc.cpu().numpy() if with_crc else None
        for j, (b, p, ok) in enumerate(all_units):
            want_st, want_out = orc.decode_all_tags(b, 65536) if (j < 96 or not ok) else (0, p)
            assert int(st[j]) == want_st, (j, int(st[j]), want_st)
            if ok:
                assert want_st == 0 and int(ol[j]) == len(p)
                assert got[out_off[j]:out_off[j] + len(p)].tobytes() == p, (with_crc, j)
                if with_crc and j % 7 == 0:
                    assert int(crcs[j]) & 0xffffffff == orc.masked_crc(p), j
            else:
                assert want_st == hip.INVALID_INPUT and int(ol[j]) == 0
    # the first 96 units alone: a small batch (the team of four waves a unit; these units through the indexed decoder)
    d_out_len = torch.zeros(96, dtype=torch.int32, device="cuda")
    d_status = torch.full((96,), 77, dtype=torch.int32, device="cuda")
    d_dec = torch.zeros(opos, dtype=torch.uint8, device="cuda")
    ctx.decode_blocks(d_stream, d_in_off, d_in_len, 96, d_dec, d_out_off, d_out_cap, d_out_len, d_status, unit=hip.UNIT_BODY)
    ctx.sync()
    got = d_dec.cpu().numpy()
    assert not d_status.cpu().numpy().any()
    for j in range(96):
        p_j = all_units[j][1]
        assert int(d_out_len[j].item()) == len(p_j) and got[out_off[j]:out_off[j] + len(p_j)].tobytes() == p_j, j
    ctx.close()


def _period_stream(lit, off, total, tail_form=2):
    """A literal, then copies of ONE offset up to `total` output bytes (what encodeBlock makes of a period): copy2
    elements of 64 bytes and a last shorter one (tail_form 1: as a copy1 when it fits that form)."""
    body, out = bytearray(), bytearray(lit)
    m = len(lit) - 1
    if m < 60:
        body.append(m << 2)
    else:
        ll = max((m.bit_length() + 7) // 8, 1)
        body.extend(bytes([(59 + ll) << 2]) + m.to_bytes(ll, "little"))
    body.extend(lit)
    while len(out) < total:
        ln = min(64, total - len(out))
        if ln < 4:  # (no copy2 shorter than... as a literal that carries the period on: then the unit is no pure period)
            body.append((ln - 1) << 2)
            for _ in range(ln):
                body.append(out[-off])
                out.append(out[-off])
            continue
        if tail_form == 1 and ln <= 11 and off < 2048:
            body.extend(bytes([((off >> 8) << 5) | ((ln - 4) << 2) | 1, off & 0xff]))
        else:
            body.extend(bytes([((ln - 1) << 2) | 2]) + off.to_bytes(2, "little"))
        for _ in range(ln):
            out.append(out[-off])
    return bytes(body), bytes(out)


@pytest.mark.gpu
def test_units_the_index_pass_writes_itself(hip, orc, torch_mod):
    """The two fast paths of the index pass (sparse_kernel.h: early_literal_unit, early_period_unit) and everything
    that must NOT take them: one-literal units of every length class at every output alignment; periods of offset
    1..4096 and beyond, with short and long first literals, totals that are no multiple of 16, unaligned outputs, a last
    element in the other copy form, one record with another offset, a stream of more than 4 KiB -- each against the
    oracle, with and without the CRC."""
    torch = torch_mod
    rng = random.Random(0xEA71)
    units = []
    for n in (1, 2, 15, 16, 17, 59, 60, 61, 255, 256, 257, 1023, 4096, 65535, 65536):  # one literal
        data = rng.randbytes(n)
        m = n - 1
        hdr = bytes([m << 2]) if m < 60 else bytes([(59 + max((m.bit_length() + 7) // 8, 1)) << 2]) + m.to_bytes(
            max((m.bit_length() + 7) // 8, 1), "little")
        units.append((hdr + data, data))
    for off, l0, total in ((1, 1, 65536), (1, 1, 65521), (2, 7, 4099), (3, 3, 65536), (10, 10, 65536), (10, 33, 65530),
                           (64, 64, 65536), (255, 300, 65536), (1000, 1000, 30000), (2047, 2047, 65536),
                           (2048, 2100, 65536), (4096, 4096, 65536), (4097, 4097, 65536), (5000, 6000, 65536),
                           (7, 61, 100), (1, 1, 5), (16, 16, 48), (4096, 5000, 9000)):
        lit = rng.randbytes(l0)
        units.append(_period_stream(lit, off, total))
        units.append(_period_stream(lit, off, total, tail_form=1))
    # one record of another offset in the middle: not a period (the indexed decoder takes it)
    b, p = _period_stream(rng.randbytes(40), 20, 65536)
    k = 1 + 40 + 3 * 7
    assert b[k] & 3 == 2
    b2 = b[:k + 1] + (21).to_bytes(2, "little") + b[k + 3:]
    st2, p2 = orc.decode_all_tags(b2, 65536)
    assert st2 == 0
    units.append((b2, p2))
    # a stream of more than 4 KiB that is a period all the same (short copies): the indexed decoder's job
    body, out = bytearray([3 << 2]) + bytearray(b"abcd"), bytearray(b"abcd")
    while len(out) < 65536 - 8:
        body.extend(bytes([((4 - 1) << 2) | 2, 4, 0]))
        out.extend(out[-4:])
    units.append((bytes(body), bytes(out)))
    for b, p in units:
        assert orc.decode_all_tags(b, len(p)) == (0, p)
    nu = len(units)
    in_off, out_off, pos, opos = [], [], 0, 0
    for k, (b, p) in enumerate(units):
        in_off.append(pos)
        pos += len(b) + rng.randint(0, 7)
        opos = (opos + 15) & ~15
        if k % 2:
            opos += rng.randint(1, 15)  # (an unaligned output: the period path declines, the literal path does not care)
        out_off.append(opos)
        opos += len(p) + rng.randint(0, 3)
    stream = np.zeros(pos + 64, np.uint8)
    for j, o in enumerate(in_off):
        stream[o:o + len(units[j][0])] = np.frombuffer(units[j][0], np.uint8)
    ctx = hip.Context(0)
    d_stream = _dev(torch, stream)
    d_in_off = _dev(torch, np.array(in_off, np.int64))
    d_in_len = _dev(torch, np.array([len(u[0]) for u in units], np.int32))
    d_out_off = _dev(torch, np.array(out_off, np.int64))
    d_out_cap = _dev(torch, np.array([len(u[1]) for u in units], np.int32))
    for with_crc in (False, True):
        d_out_len = torch.zeros(nu, dtype=torch.int32, device="cuda")
        d_status = torch.full((nu,), 77, dtype=torch.int32, device="cuda")
        d_dec = torch.full((opos + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        d_crc = torch.zeros(nu, dtype=torch.int32, device="cuda") if with_crc else None
        before = ctx.kernel_ms(10)[1]
        ctx.decode_blocks(d_stream, d_in_off, d_in_len, nu, d_dec, d_out_off, d_out_cap, d_out_len, d_status,
                          unit=hip.UNIT_BODY, d_crc=d_crc)
        ctx.sync()
        assert ctx.kernel_ms(10)[1] - before >= 15 + 10  # (every one-literal unit, the aligned periods of offset <= 4096)
        st = d_status.cpu().numpy()
        ol = d_out_len.cpu().numpy()
        got = d_dec.cpu().numpy()
        prev_end = 0
        for j, (b, p) in enumerate(units):
            assert int(st[j]) == 0 and int(ol[j]) == len(p), (j, int(st[j]), int(ol[j]), len(p))
            assert got[out_off[j]:out_off[j] + len(p)].tobytes() == p, (with_crc, j)
            assert (got[prev_end:out_off[j]] == 0xA5).all(), j  # (nothing written between the units)
            prev_end = out_off[j] + len(p)
            if with_crc:
                assert int(d_crc[j].item()) & 0xffffffff == orc.masked_crc(p), j
        assert (got[prev_end:] == 0xA5).all()
    ctx.close()

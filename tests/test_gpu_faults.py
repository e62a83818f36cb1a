"""The recovery code behind checks that never fail on this hardware, run under the parity tests.

The encoder relies on LDS atomics serving the lanes of one address in ascending lane order (ds_mskor_rtn_b32 for
the hash table's one-trip phase, ds_cmpst_rtn_b32 for the chain walk, encode_kernel.h); the order is not documented,
so every round checks it -- and what runs when a check fails (the slots put back, the plain table form, the
register loop; encoder.nim:281-309 is what all of them restate) has never run on gfx950.  The same holds for the
decoder's bounded wait for a turn (decode2_kernel.h): a turn that is given up on hands the unit to the one-pass
kernel.  These tests build a VARIANT of the library from the same sources with
    -DENC_INJECT_ORDER_FAULT=k   every k-th look at an order check finds it failed
    -DD2_INJECT_GIVE_UP=k        every k-th turn that has to be waited for is given up on
(never the shipped library: nim-snappy_amd/libsnappy_hip.so is built without them), point the package at it with
SNAPPY_HIP_LIBRARY, and run the bit-exactness tests against it in a child process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r'''
import importlib, json, os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tools"), os.path.join(%(root)r, "oracle")]
import numpy as np, torch
hip = importlib.import_module("nim-snappy_amd")
import corpus, pyoracle as orc
assert hip.LIB_PATH == os.environ["SNAPPY_HIP_LIBRARY"]
nb = 96
src = corpus.make_blocks(0, nb)                      # the corpus mix: text, html, repeated strings, periods ...
dev = torch.device("cuda", 0)
ctx = hip.Context(0)
d_in = torch.from_numpy(src.reshape(-1)).to(dev)
d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
d_sizes = torch.empty(nb, dtype=torch.int32, device=dev)
d_off = torch.empty(nb + 1, dtype=torch.int64, device=dev)
ctx.encode_blocks(d_in, nb * 65536, d_slots, d_sizes); ctx.sync()
sizes = d_sizes.cpu().numpy()
slots = d_slots.cpu().numpy().reshape(nb, hip.SLOT_STRIDE)
enc_equal = all(slots[i, :sizes[i]].tobytes() == orc.encode(src[i].tobytes()) for i in range(nb))
d_packed = torch.empty(int(sizes.sum()) + 64, dtype=torch.uint8, device=dev)
ctx.pack(d_slots, d_sizes, nb, d_packed, d_off); ctx.sync()
d_out = torch.zeros(nb * 65536, dtype=torch.uint8, device=dev)
d_oo = torch.arange(nb, dtype=torch.int64, device=dev) * 65536
d_oc = torch.full((nb,), 65536, dtype=torch.int32, device=dev)
d_ol = torch.zeros(nb, dtype=torch.int32, device=dev)
d_st = torch.full((nb,), 77, dtype=torch.int32, device=dev)
before = ctx.kernel_ms(9)[0]
ctx.decode_blocks(d_packed, d_off[:nb].contiguous(), d_sizes, nb, d_out, d_oo, d_oc, d_ol, d_st); ctx.sync()
given_up = ctx.kernel_ms(9)[0] - before
print(json.dumps({"enc_equal": bool(enc_equal), "dec_equal": bool(torch.equal(d_out, d_in)),
                  "status_ok": bool((d_st == 0).all().item()), "given_up": int(given_up)}))
'''


def _variant(k):
    """tools/probes/lib_fault<k>.so, built here (hipcc cross-compiles anywhere) unless it travelled with the tree"""
    path = os.path.join(ROOT, "tools", "probes", "lib_fault%d.so" % k)
    src = os.path.join(ROOT, "nim-snappy_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src))
    if not os.path.exists(path) or os.path.getmtime(path) < newest:
        subprocess.run([os.path.join(ROOT, "tools", "mkvariant.sh"), "fault%d" % k, "-DENC_INJECT_ORDER_FAULT=%d" % k,
                        "-DD2_INJECT_GIVE_UP=%d" % k], check=True, capture_output=True, timeout=900)
    return path


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 7])
def test_parity_with_the_recovery_paths_forced(k):
    lib = _variant(k)
    # (SNAPPY_HIP_ENC_GWAVES=4,1: the encoder's second waves -- tables in global memory, a path of their own with order
    # checks of their own -- run on every batch, not only on those that fill the GPU)
    env = dict(os.environ, SNAPPY_HIP_LIBRARY=lib, SNAPPY_HIP_ENC_GWAVES="4,1")
    # the encoder's byte equality with the oracle (every data file, the structured fuzz, the corpus sample) and the
    # decoder's round trips, with every k-th check failing / every k-th waited-for turn given up on
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py") + "::test_data_files_bit_exact",
                        os.path.join(ROOT, "tests", "test_gpu_batch.py") + "::test_structured_fuzz_round_trip",
                        os.path.join(ROOT, "tests", "test_gpu_batch.py") + "::test_corpus_sample_bit_exact"],
                       capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    # the injected give-ups are seen (in this library only), and harmless: the one-pass kernel redoes those units
    p = subprocess.run([sys.executable, "-c", PROBE % {"root": ROOT}], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    j = json.loads(p.stdout.strip().splitlines()[-1])
    assert j["enc_equal"] and j["dec_equal"] and j["status_ok"], j
    assert j["given_up"] > 0, j
    # ... and never in the shipped library
    env2 = {kk: v for kk, v in os.environ.items() if kk != "SNAPPY_HIP_LIBRARY"}
    env2["SNAPPY_HIP_LIBRARY"] = os.path.join(ROOT, "nim-snappy_amd", "libsnappy_hip.so")
    q = subprocess.run([sys.executable, "-c", PROBE % {"root": ROOT}], capture_output=True, text=True, timeout=900, env=env2)
    assert q.returncode == 0, q.stdout[-2000:] + q.stderr[-2000:]
    j0 = json.loads(q.stdout.strip().splitlines()[-1])
    assert j0["enc_equal"] and j0["dec_equal"] and j0["given_up"] == 0, j0


@pytest.mark.gpu
@pytest.mark.parametrize("gwaves", ["4,1", "0"])
def test_encoder_with_and_without_its_second_waves(gwaves):
    """The shipped library, the encoder's second waves (encode_kernel.h, encode_one_block<true>: the table in global
    memory, same-slot lanes found through an LDS scratch) on EVERY batch however small -- or on none: the encoder's byte
    equality with the oracle either way (blocks are handed out dynamically: both kinds of wave encode every kind of block)."""
    env = dict(os.environ, SNAPPY_HIP_ENC_GWAVES=gwaves)
    env.pop("SNAPPY_HIP_LIBRARY", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py") + "::test_data_files_bit_exact",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py") + "::test_every_small_length",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py") + "::test_match_runs_into_block_end",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py") + "::test_copies_longer_than_a_round_at_every_lane",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py") + "::test_synthetic_patterns_bit_exact",
                        os.path.join(ROOT, "tests", "test_gpu_batch.py") + "::test_structured_fuzz_round_trip",
                        os.path.join(ROOT, "tests", "test_gpu_batch.py") + "::test_corpus_sample_bit_exact"],
                       capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_checksumming_ring_instantiation_variant():
    """decode_indexed_kernel<16384, true> -- the CRC out of the ring's flush -- is not what the shipped library uses for
    decode_blocks' d_crc any more (decode2_kernel.h, D2_FUSED_CRC): a variant library built with it runs the tests that
    compare the decoder's CRCs and bytes with the oracle's (ragged units, ring edges, a framed batch)."""
    path = os.path.join(ROOT, "tools", "probes", "lib_fusedcrc.so")
    src = os.path.join(ROOT, "nim-snappy_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src))
    if not os.path.exists(path) or os.path.getmtime(path) < newest:
        subprocess.run([os.path.join(ROOT, "tools", "mkvariant.sh"), "fusedcrc", "-DD2_FUSED_CRC=1"], check=True,
                       capture_output=True, timeout=900)
    env = dict(os.environ, SNAPPY_HIP_LIBRARY=path)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_batch.py") + "::test_decode_crc_ragged_units",
                        os.path.join(ROOT, "tests", "test_gpu_batch.py") + "::test_ring_window_decoder",
                        os.path.join(ROOT, "tests", "test_gpu_batch.py") + "::test_framed_batch_bit_exact",
                        os.path.join(ROOT, "tests", "test_gpu_batch.py") + "::test_units_the_index_pass_writes_itself"],
                       capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


SPLIT_PROBE = r'''
import importlib, json, os, random, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tools"), os.path.join(%(root)r, "oracle"), os.path.join(%(root)r, "tests")]
hip = importlib.import_module("nim-snappy_amd")
import corpus, pyoracle as orc
from conftest import golden_file
assert hip.LIB_PATH == os.environ["SNAPPY_HIP_LIBRARY"]
rng = random.Random(3)
text = golden_file("alice29.txt") + golden_file("html")
srcs = [corpus.make_blocks(0, 96).tobytes(), text[:300000], rng.randbytes(300000),
        b"".join(rng.randbytes(rng.randint(1000, 9000)) * rng.randint(2, 4) for _ in range(40)),
        (text[5000:5010] * 40000)[:400000]]
ok = True
comps = [orc.encode(src) for src in srcs]
for src, comp in zip(srcs, comps):
    ok = ok and hip.decode(comp) == src
print("CLEAN STREAMS DONE", file=sys.stderr, flush=True)
for comp in comps:  # damaged: the verdict (or the bytes) of the oracle
    bad = bytearray(comp)
    bad[len(bad) // 2] ^= 0x40
    ok = ok and hip.decode(bytes(bad)) == orc.decode(bytes(bad))
print(json.dumps({"ok": bool(ok)}))
'''


@pytest.mark.gpu
@pytest.mark.parametrize("knobs,path", [("24,16", "none"), ("24,1", "looks"), ("24,0", "serial"), ("2,16", "none")])
def test_raw_buffer_splitter_with_its_fallbacks_forced(knobs, path):
    """uncompress of one raw multi-block buffer (split_kernels.h) enqueues its rounds without a look from the host; a
    chain that is not complete behind them is walked again with a look a round, and what that cannot finish goes to the
    one-workgroup walk.  A debug build of the same sources (-DSNAPPY_HIP_DEBUG: it reads SNAPPY_HIP_SPLIT_KNOBS =
    "budget,hops") forces each: a tail round that follows a chain for one hop only leaves the corpus mix incomplete
    behind the blind rounds ("looks"); one that walks nothing never completes anything ("serial"); a budget of two
    elements sends nearly every second walk to the queue.  The bytes and the verdicts are the oracle's every time."""
    lib = os.path.join(ROOT, "tools", "probes", "lib_dbg.so")
    src = os.path.join(ROOT, "nim-snappy_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src))
    if not os.path.exists(lib) or os.path.getmtime(lib) < newest:
        subprocess.run([os.path.join(ROOT, "tools", "mkvariant.sh"), "dbg", "-DSNAPPY_HIP_DEBUG"], check=True,
                       capture_output=True, timeout=900)
    env = dict(os.environ, SNAPPY_HIP_LIBRARY=lib, SNAPPY_HIP_SPLIT_KNOBS=knobs, SNAPPY_HIP_STATS="1")
    p = subprocess.run([sys.executable, "-c", SPLIT_PROBE % {"root": ROOT}], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["ok"], p.stderr[-2000:]
    clean = p.stderr.split("CLEAN STREAMS DONE")[0]  # (a damaged stream may take any of the paths)
    again = "again, with looks" in clean
    gave_up = any(ln.startswith("SPLIT look: root ->") and "root -> fffffffe" not in ln for ln in clean.splitlines())
    assert again == (path != "none"), p.stderr[-1500:]
    assert gave_up == (path == "serial"), p.stderr[-1500:]


FRAMED_TAIL_PROBE = r'''
import importlib, json, os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tools"), os.path.join(%(root)r, "oracle"), os.path.join(%(root)r, "tests")]
import numpy as np, torch
hip = importlib.import_module("nim-snappy_amd")
import corpus, pyoracle as orc
assert hip.LIB_PATH == os.environ["SNAPPY_HIP_LIBRARY"]
MiB = 1 << 20
rng = np.random.default_rng(11)
body = corpus.make_blocks(0, 160).tobytes()
rnd = [rng.integers(0, 256, 65536, dtype=np.uint8).tobytes() for _ in range(17)]  # stored chunks: 65 544 bytes each
body_stream = orc.encode_framed(body)
rnd_chunks = [orc.encode_framed(b)[10:] for b in rnd]
assert all(len(c) == 65544 and c[0] == 1 for c in rnd_chunks)
ctx = hip.Context(0)
res = []
for r in %(tails)r:  # bytes of the stream in its last slice of 1 MiB; the last chunk is a stored one of 65 544 bytes
    # k more stored chunks and one small padding chunk (snappy.nim:262-263) behind the stream identifier give the length
    k, pad = min(((k, (r - len(body_stream) - (k + 1) * 65544) %% MiB) for k in range(16)), key=lambda t: t[1] if t[1] >= 4 else MiB)
    assert 4 <= pad < 70000
    s = body_stream[:10] + b"\xfe" + (pad - 4).to_bytes(3, "little") + bytes(pad - 4) + body_stream[10:] + b"".join(rnd_chunks[:k + 1])
    src = body + b"".join(rnd[:k + 1])
    assert len(s) >= 4 * MiB and len(s) %% MiB == r
    d_in = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
    d_out = torch.zeros(len(src), dtype=torch.uint8, device="cuda")
    print("TAIL %%d" %% r, file=sys.stderr, flush=True)
    st, rd, wr = ctx.uncompress_framed(d_in, len(s), d_out, len(src))
    res.append({"tail": r, "st": st, "rd_ok": rd == len(s), "wr_ok": wr == len(src),
                "bytes_ok": bytes(d_out.cpu().numpy().tobytes()) == src})
print(json.dumps(res))
'''


@pytest.mark.gpu
def test_framed_stream_whose_last_chunk_covers_its_last_slice():
    """The parallel chunk walk (framed_kernels.h) cuts the stream into slices of 1 MiB; when the final chunk starts in
    front of the last slice and ends at the stream's end, the chain never enters that slice (n mod 1 MiB bytes, all of
    them inside the chunk's body).  Round 5's stitch called such a stream irregular and sent it to the serial walk --
    right bytes, five times the time, for 2-4 %% of all stream lengths.  A debug build says which walk ran."""
    lib = os.path.join(ROOT, "tools", "probes", "lib_dbg.so")
    src = os.path.join(ROOT, "nim-snappy_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src))
    if not os.path.exists(lib) or os.path.getmtime(lib) < newest:
        subprocess.run([os.path.join(ROOT, "tools", "mkvariant.sh"), "dbg", "-DSNAPPY_HIP_DEBUG"], check=True,
                       capture_output=True, timeout=900)
    tails = [4, 1000, 65543, 65544, 65545, 300000]
    env = dict(os.environ, SNAPPY_HIP_LIBRARY=lib, SNAPPY_HIP_STATS="1")
    p = subprocess.run([sys.executable, "-c", FRAMED_TAIL_PROBE % {"root": ROOT, "tails": tails}], capture_output=True,
                       text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert [r["tail"] for r in res] == tails
    for r in res:
        assert r["st"] == 0 and r["rd_ok"] and r["wr_ok"] and r["bytes_ok"], r
    for part in p.stderr.split("TAIL ")[1:]:  # every one of them by the parallel walk
        assert "FRAME WALK stitch irregular 0" in part and "fast_ok 1" in part, part[:600]

#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on the MI355X hot path.

    python bench.py --gpus N --steps K --warmup W

metric  : GB/s of UNCOMPRESSED bytes (GB = 1e9), Snappy block codec, 4 GiB many-block corpus
workload: configs[1] of BASELINE.json -- 1 x MI355X block DECOMPRESS of 65 536 x 64 KiB synthetic
          blocks (seeded corpus of SURVEY.md 8d, tools/corpus.py), independent raw-Snappy
          buffers with an offset table, input resident in HBM when the timed region starts.
          One "step" = one decode pass over the whole batch (one kernel launch + hand-over pass).
          The same JSON line also carries the compress and framed numbers of the same corpus
          (configs[2], configs[3]) as extra keys; `value` is the decompress rate.
multi-GPU: every rank owns its own 65 536-block range of the corpus (weak scaling, no data-path
          collective -- blocks are independent); value = all ranks' bytes / max time.

roofline : block decode is HBM-bound byte work.  Dominant kernel = decode_indexed_kernel<32768> (pass 2
          of the v2 decoder, ring-window instantiation; pass 1, index_units_kernel, and the whole-block
          instantiation's launch over the units the ring one passes on are reported beside it; `achieved` divides
          by the two decode launches' durations together, `traffic` is the ring kernel's).  achieved =
          (sum C + sum U) per launch / average kernel duration, measured with HIP events on the
          launch stream inside the timed region (snappy_hip_ctx_kernel_ms).
          peak = 8000 GB/s (MI355X_MICROARCH.md).
cpu_baseline: the CPU oracle (oracle/snappy_oracle.c, a bit-exact restatement of the reference --
          the Nim reference itself cannot be built here) timed single-threaded on this box on a
          bounded sample of the same blocks.  kind = "port".
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0
BLOCK = 65536


def measured_traffic(nb, only):
    """HBM bytes per launch of the dominant kernel from the PMC passes of tools/profile_bench.sh
    (FETCH_SIZE / WRITE_SIZE cannot be read inside this process: they need their own rocprofv3
    passes).  The committed measurement is for the default workload only; anything else: None."""
    if nb != 65536 or only is not None:
        return None
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
    if not cands:
        return None
    with open(os.path.join(ROOT, "profiles", cands[-1])) as f:
        t = json.load(f)
    k = t.get("kernels", {}).get("decode_indexed_kernel<32768>") or t.get("kernels", {}).get("decode_indexed_kernel")
    return None if k is None else k["total_bytes"]


def cpu_baseline(corpus, d_in, d_packed, offsets, sizes, n_sample, budget_s):
    """Oracle (CPU port of the reference) on the first n_sample blocks, one thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes
    import pyoracle as orc
    src = d_in[:n_sample * BLOCK].cpu().numpy()
    slot = 76800
    out = np.empty(n_sample * slot, dtype=np.uint8)
    csz = np.empty(n_sample, dtype=np.uint32)
    t0 = time.perf_counter()
    orc.lib.sor_compress_blocks(src.ctypes.data, src.size, BLOCK, out.ctypes.data, slot,
                                csz.ctypes.data)
    t_enc = time.perf_counter() - t0
    # parity spot check: the GPU's packed stream for these blocks is byte-identical
    gpu = d_packed[:int(offsets[n_sample])].cpu().numpy()
    for i in range(0, n_sample, max(1, n_sample // 256)):
        a = out[i * slot:i * slot + int(csz[i])]
        b = gpu[int(offsets[i]):int(offsets[i]) + int(sizes[i])]
        if int(csz[i]) != int(sizes[i]) or not np.array_equal(a, b):
            raise SystemExit("bench: GPU encoding of block %d differs from the oracle" % i)
    offs = (np.arange(n_sample, dtype=np.uint64) * slot)
    dec = np.empty(n_sample * BLOCK, dtype=np.uint8)
    passes, t_dec = 0, 0.0
    while t_dec < budget_s and passes < 64:
        t0 = time.perf_counter()
        st = orc.lib.sor_uncompress_blocks(out.ctypes.data, offs.ctypes.data, csz.ctypes.data,
                                           n_sample, dec.ctypes.data, BLOCK)
        t_dec += time.perf_counter() - t0
        passes += 1
        assert st == 0
    assert np.array_equal(dec, src)
    # the same sample, block ranges spread over all host threads (SURVEY 8d "B2"; ctypes releases the GIL)
    import concurrent.futures as cf
    nthr = min(os.cpu_count() or 1, 64, n_sample)
    per = (n_sample + nthr - 1) // nthr

    def part(k):
        lo, hi = k * per, min(n_sample, (k + 1) * per)
        if lo >= hi:
            return 0
        return orc.lib.sor_uncompress_blocks(out.ctypes.data, offs[lo:].ctypes.data, csz[lo:].ctypes.data,
                                             hi - lo, dec.ctypes.data + lo * BLOCK, BLOCK)

    def cpart(k):  # compress, block ranges spread over the same threads
        lo, hi = k * per, min(n_sample, (k + 1) * per)
        if lo < hi:
            orc.lib.sor_compress_blocks(src.ctypes.data + lo * BLOCK, (hi - lo) * BLOCK, BLOCK,
                                        out2.ctypes.data + lo * slot, slot, csz2.ctypes.data + 4 * lo)
        return 0

    out2 = np.empty(n_sample * slot, dtype=np.uint8)
    csz2 = np.empty(n_sample, dtype=np.uint32)
    with cf.ThreadPoolExecutor(nthr) as ex:
        list(ex.map(part, range(nthr)))  # warm
        t0 = time.perf_counter()
        mt_passes = 4
        for _ in range(mt_passes):
            assert all(r == 0 for r in ex.map(part, range(nthr)))
        t_mt = time.perf_counter() - t0
        t0 = time.perf_counter()
        list(ex.map(cpart, range(nthr)))
        t_mt_enc = time.perf_counter() - t0
    assert np.array_equal(csz2, csz)
    return {
        "value": round(n_sample * BLOCK * passes / t_dec / 1e9, 4),
        "unit": "GB/s uncompressed (decompress)",
        "cores": 1,
        "kind": "port",
        "sample": "first %d blocks (%d MiB) of the same corpus, %d decode passes; "
                  "oracle/snappy_oracle.c -O3, one thread" % (n_sample, n_sample // 16, passes),
        "compress_value": round(n_sample * BLOCK / t_enc / 1e9, 4),
        "threads": nthr,
        "threads_value": round(n_sample * BLOCK * mt_passes / t_mt / 1e9, 3),
        "compress_threads_value": round(n_sample * BLOCK / t_mt_enc / 1e9, 3),
        "host_cpus": os.cpu_count(),
    }


def config1_alice29(hip):
    """BASELINE configs[0]: tests/data/alice29.txt single-buffer encode / decode, timed the way the
    reference's harness does (tests/benchmark.nim:20-23,93-104: mean over 100 calls, ms per call) --
    the oracle on one host core (the Nim inMemory path cannot be built here), and the HIP library's
    host-buffer calls (PCIe and launch latency included: 3 blocks cannot fill a GPU)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    with open(os.path.join(ROOT, "tests", "golden", "data", "alice29.txt"), "rb") as fh:
        src = fh.read()

    def mean_ms(f, reps=100):
        f()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        return (time.perf_counter() - t0) / reps * 1e3

    enc = orc.encode(src)
    fr = orc.encode_framed(src)
    assert hip.encode(src) == enc and hip.encode_framed(src) == fr and hip.decode(enc) == src
    return {
        "file": "alice29.txt", "bytes": len(src), "calls": 100, "unit": "ms per call (encode / decode)",
        "oracle_inMemory_raw": [round(mean_ms(lambda: orc.encode(src)), 4), round(mean_ms(lambda: orc.decode(enc)), 4)],
        "oracle_inMemory_framed": [round(mean_ms(lambda: orc.encode_framed(src)), 4),
                                   round(mean_ms(lambda: orc.decode_framed(fr)), 4)],
        "hip_host_api_raw": [round(mean_ms(lambda: hip.encode(src)), 4), round(mean_ms(lambda: hip.decode(enc)), 4)],
        "hip_host_api_framed": [round(mean_ms(lambda: hip.encode_framed(src)), 4),
                                round(mean_ms(lambda: hip.decode_framed(fr)), 4)],
        "reference_README_x86_64": {"raw": [0.334, 0.186], "framed": [0.382, 0.251]},
    }


def host_api_rates(hip, src_np, ctx=None, dev=None):
    """The host-buffer C ABI end to end (PCIe copies included; never `value`): GB/s of uncompressed
    bytes for one call over src_np, and two host threads at once."""
    import ctypes
    import threading
    lib = hip.lib
    n = src_np.size
    P = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
    C = lambda a: ctypes.cast(P(a), ctypes.c_char_p)  # noqa: E731

    def best(f, reps=2):
        f()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return min(ts)

    cap = hip.max_compressed_len_framed(n)
    fr = np.empty(cap, dtype=np.uint8)
    back = np.empty(n, dtype=np.uint8)
    w, r = ctypes.c_size_t(), ctypes.c_size_t()

    def cf(inp=src_np, out=fr, ww=w):
        assert lib.snappy_hip_compress_framed(C(inp), inp.size, P(out), out.size, ctypes.byref(ww)) == 0

    def uf():
        assert lib.snappy_hip_uncompress_framed(C(fr), flen, P(back), n, 1, 1, ctypes.byref(r), ctypes.byref(w)) == 0

    res = {"bytes": int(n)}
    res["compress_framed_GBps"] = round(n / best(cf) / 1e9, 2)
    flen = w.value
    res["uncompress_framed_GBps"] = round(n / best(uf) / 1e9, 2)
    assert w.value == n and np.array_equal(back, src_np)
    raw = np.empty(hip.max_compressed_len(n), dtype=np.uint8)

    def cr():
        assert lib.snappy_hip_compress(C(src_np), n, P(raw), raw.size, ctypes.byref(w)) == 0

    res["compress_GBps"] = round(n / best(cr) / 1e9, 2)
    rlen = w.value
    back[:] = 0

    def ur():
        assert lib.snappy_hip_uncompress(C(raw), rlen, P(back), n, ctypes.byref(w)) == 0

    res["uncompress_GBps"] = round(n / best(ur, reps=1) / 1e9, 2)
    assert w.value == n and np.array_equal(back, src_np)
    if ctx is not None:  # the same raw multi-block buffer resident in HBM (snappy_hip_uncompress_d), and 64 MiB of it
        for tag, nblk in (("raw_buffer_uncompress_d_GBps", n // BLOCK), ("raw_buffer_64MiB_uncompress_d_GBps", 1024)):
            part = src_np[:nblk * BLOCK]
            rr = np.empty(hip.max_compressed_len(part.size), dtype=np.uint8)
            assert lib.snappy_hip_compress(C(part), part.size, P(rr), rr.size, ctypes.byref(w)) == 0
            d_raw = torch.from_numpy(rr[:w.value]).to(dev)
            d_back = torch.empty(part.size, dtype=torch.uint8, device=dev)
            rl = int(w.value)

            def ud():
                assert ctx.uncompress(d_raw, rl, d_back, part.size) == (0, part.size)

            res[tag] = round(part.size / best(ud) / 1e9, 2)
            assert np.array_equal(d_back.cpu().numpy(), part)
    half = n // 2
    outs = [np.empty(cap, dtype=np.uint8) for _ in range(2)]
    ws = [ctypes.c_size_t(), ctypes.c_size_t()]

    def work(i):
        cf(src_np[i * half:(i + 1) * half], outs[i], ws[i])

    def both():
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [x.start() for x in th]
        [x.join() for x in th]

    res["compress_framed_half_one_thread_ms"] = round(best(lambda: work(0)) * 1e3, 2)
    res["compress_framed_halves_two_threads_ms"] = round(best(both) * 1e3, 2)
    return res


def per_class_rates(hip, corpus, ctx, dev, nb):
    """Every corpus class alone ("also report each class alone"), nb blocks each:
    device-resident compress and block decompress, kernel times from HIP events."""
    out = {}
    for cls in corpus.CLASSES:
        d_in = corpus.make_blocks_torch(torch, 0, nb, dev, only=cls).reshape(-1)
        d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
        d_sizes = torch.empty(nb, dtype=torch.int32, device=dev)
        d_offsets = torch.empty(nb + 1, dtype=torch.int64, device=dev)
        ctx.encode_blocks(d_in, nb * BLOCK, d_slots, d_sizes)
        ctx.sync()
        ctx.timing(True)
        ctx.encode_blocks(d_in, nb * BLOCK, d_slots, d_sizes)
        ctx.sync()
        enc_ms, _ = ctx.kernel_ms(1)
        ctx.timing(False)
        tot = int(d_sizes.to(torch.int64).sum().item())
        d_packed = torch.empty(tot + 64, dtype=torch.uint8, device=dev)
        ctx.pack(d_slots, d_sizes, nb, d_packed, d_offsets)
        ctx.sync()
        del d_slots
        d_out = torch.empty(nb * BLOCK, dtype=torch.uint8, device=dev)
        d_out_off = torch.arange(nb, dtype=torch.int64, device=dev) * BLOCK
        d_out_cap = torch.full((nb,), BLOCK, dtype=torch.int32, device=dev)
        d_out_len = torch.zeros(nb, dtype=torch.int32, device=dev)
        d_status = torch.zeros(nb, dtype=torch.int32, device=dev)
        d_in_off = d_offsets[:nb].contiguous()

        def dec():
            ctx.decode_blocks(d_packed, d_in_off, d_sizes, nb, d_out, d_out_off, d_out_cap, d_out_len, d_status)

        dec()
        ctx.sync()
        ctx.timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dec()
        ctx.sync()
        t = (time.perf_counter() - t0) / 3
        dec_ms, _ = ctx.kernel_ms(0)
        idx_ms, _ = ctx.kernel_ms(4)
        dec2_ms, _ = ctx.kernel_ms(8)  # (the whole-block instantiation over the units the ring one passed on)
        ctx.timing(False)
        dec_ms += dec2_ms
        assert bool(torch.equal(d_out, d_in)), cls
        u = nb * BLOCK
        out[cls] = {
            "decompress_GBps": round(u / t / 1e9, 1),
            "decode_kernel_ms": round(dec_ms, 3), "passed_on_units_kernel_ms": round(dec2_ms, 3),
            "index_pass_ms": round(idx_ms, 3),
            "decode_frac_of_hbm_peak": round((u + tot) / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if dec_ms else None,
            "compress_GBps": round(u / (enc_ms * 1e-3) / 1e9, 1) if enc_ms else None,
            "compress_kernel_ms": round(enc_ms, 3),
            "compressed_over_uncompressed": round(tot / u, 3),
        }
        del d_in, d_packed, d_out
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=65536, help="blocks per GPU (4 GiB)")
    ap.add_argument("--only", default=None, help="one corpus class (per-class numbers)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-blocks", type=int, default=8192)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # test hook: BENCH_SHARE_DEVICE=1 BENCH_DIST_BACKEND=gloo lets several ranks share GPU 0, so that
    # the N > 1 code path can be exercised on a one-GPU box (RCCL refuses two ranks on one device)
    if os.environ.get("BENCH_SHARE_DEVICE"):
        local = 0
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    hip = importlib.import_module("nim-snappy_amd")  # raises if the HIP library is missing
    import corpus
    import shard
    ctx = hip.Context(local)
    nb = args.blocks

    # ---- corpus for this rank: blocks [rank*nb, (rank+1)*nb) -------------------------------------
    d_in = torch.empty(nb * BLOCK, dtype=torch.uint8, device=dev)
    for b0 in range(0, nb, 4096):
        c = min(4096, nb - b0)
        d_in[b0 * BLOCK:(b0 + c) * BLOCK] = corpus.make_blocks_torch(
            torch, shard.first_block(rank, nb) + b0, c, dev, only=args.only).reshape(-1)
    chk = corpus.make_blocks(shard.first_block(rank, nb), 8, only=args.only).reshape(-1)
    assert np.array_equal(d_in[:8 * BLOCK].cpu().numpy(), chk), "device corpus != numpy corpus"

    # ---- compress on the device (also gives the compress number) -----------------------------------
    d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
    d_sizes = torch.empty(nb, dtype=torch.int32, device=dev)
    d_offsets = torch.empty(nb + 1, dtype=torch.int64, device=dev)
    ctx.encode_blocks(d_in, nb * BLOCK, d_slots, d_sizes, unit=hip.UNIT_RAW)  # warm
    ctx.sync()
    enc_steps = max(1, min(3, args.steps))
    ctx.timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(enc_steps):
        ctx.encode_blocks(d_in, nb * BLOCK, d_slots, d_sizes, unit=hip.UNIT_RAW)
    ctx.sync()
    torch.cuda.synchronize()
    t_enc = (time.perf_counter() - t0) / enc_steps
    enc_ms, _ = ctx.kernel_ms(1)
    ctx.timing(False)
    sum_c = int(d_sizes.to(torch.int64).sum().item())
    d_packed = torch.empty(sum_c + 64, dtype=torch.uint8, device=dev)
    ctx.pack(d_slots, d_sizes, nb, d_packed, d_offsets)
    ctx.sync()
    del d_slots
    d_in_off = d_offsets[:nb].contiguous()

    # ---- framed compress + uncompress of the same bytes (configs[3]): ONE genuine stream --------------
    # compressFramed: stream identifier + one chunk per block (CRC32C of every block, encodeFrame in
    # the encode kernel, contiguous pack); uncompressFramed of that stream: chunk walk, decode, CRC
    # comparison and verdict on the device.  Stream and output stay in HBM.
    fcap = hip.max_compressed_len_framed(nb * BLOCK)
    d_fstream = torch.empty(fcap, dtype=torch.uint8, device=dev)
    flen = ctx.compress_framed(d_in, nb * BLOCK, d_fstream, fcap)  # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flen = ctx.compress_framed(d_in, nb * BLOCK, d_fstream, fcap)
    t_fenc = time.perf_counter() - t0
    d_fout = torch.empty(nb * BLOCK, dtype=torch.uint8, device=dev)
    st_f = ctx.uncompress_framed(d_fstream, flen, d_fout, nb * BLOCK)  # warm
    assert st_f == (0, flen, nb * BLOCK), st_f
    ctx.timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st_f = ctx.uncompress_framed(d_fstream, flen, d_fout, nb * BLOCK)
    t_fdec = time.perf_counter() - t0
    walk_ms, _ = ctx.kernel_ms(6)
    ctx.timing(False)
    assert st_f == (0, flen, nb * BLOCK) and bool(torch.equal(d_fout, d_in)), "framed round trip differs"
    del d_fstream, d_fout

    # ---- the timed hot path: block decompress --------------------------------------------------------
    d_out = torch.empty(nb * BLOCK, dtype=torch.uint8, device=dev)
    d_out_off = torch.arange(nb, dtype=torch.int64, device=dev) * BLOCK
    d_out_cap = torch.full((nb,), BLOCK, dtype=torch.int32, device=dev)
    d_out_len = torch.zeros(nb, dtype=torch.int32, device=dev)
    d_status = torch.zeros(nb, dtype=torch.int32, device=dev)
    d_crc = torch.zeros(nb, dtype=torch.int32, device=dev)

    def step(crc=None):
        ctx.decode_blocks(d_packed, d_in_off, d_sizes, nb, d_out, d_out_off, d_out_cap, d_out_len,
                          d_status, unit=hip.UNIT_RAW, d_crc=crc)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    ctx.sync()
    ctx.timing(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    dec_ms, dec_launches = ctx.kernel_ms(0)   # decode_indexed_kernel (dominant)
    idx_ms, _ = ctx.kernel_ms(4)               # index_units_kernel
    dec2_ms, _ = ctx.kernel_ms(8)              # the whole-block instantiation over the units the ring one passed on
    ctx.timing(False)
    elapsed = shard.max_over_ranks(dist if world > 1 else None, elapsed, dev)
    # the side numbers are whole-job rates too: all ranks' bytes over the slowest rank's time
    t_enc = shard.max_over_ranks(dist if world > 1 else None, t_enc, dev)
    t_fenc = shard.max_over_ranks(dist if world > 1 else None, t_fenc, dev)

    # results must be right, or the number is void
    assert int((d_status != 0).sum().item()) == 0, "decode reported errors"
    assert bool(torch.equal(d_out, d_in)), "decoded bytes differ from the corpus"

    t_fdec = shard.max_over_ranks(dist if world > 1 else None, t_fdec, dev)

    # ---- measured copy bandwidth of this box: the second roofline denominator of SURVEY 8(d) -------
    torch.cuda.synchronize()
    d_out.copy_(d_in)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        d_out.copy_(d_in)
    torch.cuda.synchronize()
    copy_gbps = 3 * 2 * nb * BLOCK / (time.perf_counter() - t0) / 1e9

    if rank == 0:
        u_bytes = nb * BLOCK
        value = world * u_bytes * args.steps / elapsed / 1e9
        # every unit's bytes over BOTH decode launches (the ring-window one and the one over the units it passes
        # on): the ring kernel alone does not move all of them, so its duration alone would flatter it
        achieved = (sum_c + u_bytes) / ((dec_ms + dec2_ms) * 1e-3) / 1e9 if dec_ms > 0 else 0.0
        line = {
            "metric": "GB/s uncompressed throughput (compress + decompress), 4 GiB many-block corpus",
            "value": round(value, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "block decompress of %d x 64 KiB synthetic blocks per GPU "
                            "(BASELINE configs[1]; class mix %s, seed 0x5EED5AA9), raw-Snappy "
                            "units + offset table resident in HBM" % (nb, args.only or "default"),
                "blocks_per_gpu": nb,
                "uncompressed_bytes_per_gpu": u_bytes,
                "compressed_bytes_per_gpu": sum_c,
                "sharding": "block range per rank, no collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "measured_copy_GBps": round(copy_gbps, 1),  # device-to-device copy, read + write
                "frac_of_measured_copy": round(achieved / copy_gbps, 5),
                "traffic": measured_traffic(nb, args.only),
                "kernel": "decode_indexed_kernel<32768>",  # (ring window; <65536> takes the units it passes on)
                "kernel_ms": round(dec_ms, 4),  # (HIP events; rocprof's average for this kernel agrees)
                "kernel_ms_both_decode_launches": round(dec_ms + dec2_ms, 4),  # what `achieved` divides by
                "index_pass_kernel_ms": round(idx_ms, 4),
                "passed_on_units_kernel_ms": round(dec2_ms, 4),
                # turns the indexed decoder gave up on after its bounded wait (must be 0; each costs ~30 ms and
                # hands its unit to the one-pass kernel)
                "decode_turns_given_up": int(ctx.kernel_ms(9)[0]),
                "launches": dec_launches,
                "algorithmic_bytes_per_launch": sum_c + u_bytes,
            },
            "compress_GBps": round(world * u_bytes / t_enc / 1e9, 3),
            "compress_kernel_ms": round(enc_ms, 3),
            # compress + decompress of the same bytes, one after the other (BASELINE's metric names both)
            "roundtrip_GBps": round(world * u_bytes / (t_enc + elapsed / args.steps) / 1e9, 3),
            # one genuine framed stream (10-byte identifier + one chunk per block), stream and output in HBM
            "framed_compress_GBps": round(world * u_bytes / t_fenc / 1e9, 3),
            "framed_decompress_GBps": round(world * u_bytes / t_fdec / 1e9, 3),
            "framed_chunk_walk_ms": round(walk_ms, 3),
        }
        if world == 1 and not args.no_cpu:
            offs = d_offsets.cpu().numpy()
            sizes = d_sizes.cpu().numpy()
            ns = min(args.cpu_blocks, nb)
            line["cpu_baseline"] = cpu_baseline(corpus, d_in, d_packed, offs, sizes, ns, 10.0)
            line["config1_alice29"] = config1_alice29(hip)
            del d_packed, d_out
            line["per_class"] = per_class_rates(hip, corpus, ctx, dev, min(nb, 8192))
            line["host_api"] = host_api_rates(hip, d_in[:min(nb, 16384) * BLOCK].cpu().numpy(), ctx, dev)
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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

roofline : block decode is HBM-bound byte work.  Dominant kernel = decode_indexed_kernel (pass 2
          of the v2 decoder; pass 1, index_units_kernel, is reported beside it).  achieved =
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
    k = t.get("kernels", {}).get("decode_indexed_kernel")
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

    with cf.ThreadPoolExecutor(nthr) as ex:
        list(ex.map(part, range(nthr)))  # warm
        t0 = time.perf_counter()
        mt_passes = 4
        for _ in range(mt_passes):
            assert all(r == 0 for r in ex.map(part, range(nthr)))
        t_mt = time.perf_counter() - t0
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
        "host_cpus": os.cpu_count(),
    }


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

    # ---- framed compress of the same bytes (configs[3]), once ---------------------------------------
    d_fslots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
    d_fsizes = torch.empty(nb, dtype=torch.int32, device=dev)
    ctx.encode_blocks(d_in, nb * BLOCK, d_fslots, d_fsizes, unit=hip.UNIT_FRAME)
    ctx.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.encode_blocks(d_in, nb * BLOCK, d_fslots, d_fsizes, unit=hip.UNIT_FRAME)
    ctx.sync()
    t_fenc = time.perf_counter() - t0
    del d_fslots, d_fsizes

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
    ctx.timing(False)
    elapsed = shard.max_over_ranks(dist if world > 1 else None, elapsed, dev)
    # the side numbers are whole-job rates too: all ranks' bytes over the slowest rank's time
    t_enc = shard.max_over_ranks(dist if world > 1 else None, t_enc, dev)
    t_fenc = shard.max_over_ranks(dist if world > 1 else None, t_fenc, dev)

    # results must be right, or the number is void
    assert int((d_status != 0).sum().item()) == 0, "decode reported errors"
    assert bool(torch.equal(d_out, d_in)), "decoded bytes differ from the corpus"

    # ---- framed decompress pass (decode + CRC32C verify inputs), configs[3] ---------------------------
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(crc=d_crc)
    ctx.sync()
    t_fdec = time.perf_counter() - t0
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
        achieved = (sum_c + u_bytes) / (dec_ms * 1e-3) / 1e9 if dec_ms > 0 else 0.0
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
                "kernel": "decode_indexed_kernel",
                "kernel_ms": round(dec_ms, 4),
                "index_pass_kernel_ms": round(idx_ms, 4),
                "launches": dec_launches,
                "algorithmic_bytes_per_launch": sum_c + u_bytes,
            },
            "compress_GBps": round(world * u_bytes / t_enc / 1e9, 3),
            "compress_kernel_ms": round(enc_ms, 3),
            "framed_compress_GBps": round(world * u_bytes / t_fenc / 1e9, 3),
            "framed_decompress_GBps": round(world * u_bytes / t_fdec / 1e9, 3),
        }
        if world == 1 and not args.no_cpu:
            offs = d_offsets.cpu().numpy()
            sizes = d_sizes.cpu().numpy()
            ns = min(args.cpu_blocks, nb)
            line["cpu_baseline"] = cpu_baseline(corpus, d_in, d_packed, offs, sizes, ns, 10.0)
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Block-range sharded compress of one large input across N GPUs (BASELINE configs[4]).

Every 64 KiB block is encoded independently, so rank r of N owns blocks [r*B, (r+1)*B) of the
input, encodes and packs its range on its own GPU, and the ONLY exchange is the 8-byte shard total
per rank (all_gather over RCCL): an exclusive scan of the totals gives every shard's place in the
final stream.  Each rank then copies its shard to the host; the host concatenates (rank 0 adds the
single varint of the raw format).  No data-path collective: a gather of the shards over xGMI would
only move the bytes one extra time before the PCIe copy they need anyway.

Launch (one process per GPU):
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      tools/sharded_compress.py --blocks-per-gpu 65536 [--out-dir DIR]
Prints one JSON line (rank 0): whole-job GB/s of uncompressed input, shard sizes, and -- with
--verify -- that the concatenation equals the single-GPU / oracle encoding of the same bytes.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BLOCK = 65536


def varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def compress_shard(hip, ctx, d_in, nb, dev):
    """Encode + pack nb blocks resident in d_in; returns (packed device tensor, total bytes)."""
    d_slots = torch.empty(nb * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
    d_sizes = torch.empty(nb, dtype=torch.int32, device=dev)
    d_offsets = torch.empty(nb + 1, dtype=torch.int64, device=dev)
    ctx.encode_blocks(d_in, nb * BLOCK, d_slots, d_sizes, unit=hip.UNIT_BODY)
    ctx.sync()
    total = int(d_sizes.to(torch.int64).sum().item())
    d_packed = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    ctx.pack(d_slots, d_sizes, nb, d_packed, d_offsets)
    ctx.sync()
    return d_packed, total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks-per-gpu", type=int, default=65536)
    ap.add_argument("--verify", action="store_true", help="rank 0 re-encodes everything and compares")
    ap.add_argument("--out-dir", default=None, help="write shard files + the concatenated stream here")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("BENCH_SHARE_DEVICE"):  # test hook: several ranks on GPU 0 (see bench.py)
        local = 0
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    hip = importlib.import_module("nim-snappy_amd")
    import corpus
    import shard
    ctx = hip.Context(local)
    nb = args.blocks_per_gpu
    lo, hi = shard.block_range(rank, world, nb)
    d_in = torch.empty(nb * BLOCK, dtype=torch.uint8, device=dev)
    for b0 in range(0, nb, 4096):
        c = min(4096, nb - b0)
        d_in[b0 * BLOCK:(b0 + c) * BLOCK] = corpus.make_blocks_torch(torch, lo + b0, c, dev).reshape(-1)
    compress_shard(hip, ctx, d_in[:min(nb, 256) * BLOCK], min(nb, 256), dev)  # warm
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    d_packed, total = compress_shard(hip, ctx, d_in, nb, dev)
    # the one exchange: shard totals -> exclusive scan -> where my shard goes
    totals = [total]
    if world > 1:
        t = torch.tensor([total], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        totals = [int(g.item()) for g in gathered]
    n_total = world * nb * BLOCK
    header = varint(n_total) if n_total <= 0xffffffff else b""  # one raw buffer holds < 4 GiB
    base = len(header) + sum(totals[:rank])
    host = torch.empty(total, dtype=torch.uint8, pin_memory=True)
    host.copy_(d_packed[:total], non_blocking=True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(dist if world > 1 else None, elapsed, dev if backend == "nccl" else None)
    if args.out_dir:
        os.makedirs(args.out_dir, exist_ok=True)
        with open(os.path.join(args.out_dir, "shard_%03d.bin" % rank), "wb") as f:
            f.write(host.numpy().tobytes())
    ok = None
    if args.verify:
        # every rank checks its own shard against a fresh single-call encoding of the same blocks
        again, total2 = compress_shard(hip, ctx, d_in, nb, dev)
        ok = total2 == total and bool(torch.equal(again[:total].cpu(), host))
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        try:
            import pyoracle as orc
            k = min(nb, 64)
            want = b"".join(orc.encode_block(corpus.make_blocks(lo, k)[i].tobytes()) for i in range(k))
            ok = ok and host.numpy()[:len(want)].tobytes() == want
        except Exception:  # the oracle is optional here (tools are not tests)
            pass
        if world > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
    if rank == 0:
        print(json.dumps({
            "what": "block-range sharded compress, host concatenate (BASELINE configs[4])",
            "n_gpus": world, "blocks_per_gpu": nb, "uncompressed_bytes": n_total,
            "compressed_bytes": len(header) + sum(totals), "shard_bytes": totals,
            "my_base_offset": base, "raw_header_bytes": len(header),
            "seconds": round(elapsed, 4), "GBps_uncompressed": round(n_total / elapsed / 1e9, 3),
            "verified": ok,
        }), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

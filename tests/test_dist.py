"""N > 1 path on CPU: two gloo ranks shard the corpus by block range exactly as bench.py does.

The codec itself needs a GPU, so each rank runs the ORACLE on its shard here; what is under
test is the sharding contract: ranges are disjoint and complete, every block is generated
from (seed, index) alone, per-rank results concatenate in rank order to the single-process
result (the "final host-side concatenate" of the north star), and the timing reduction is a MAX.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCKS_PER_RANK = 24


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
        sys.path.insert(0, p)
    import corpus
    import pyoracle as orc
    import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.block_range(rank, world, BLOCKS_PER_RANK)
    blocks = corpus.make_blocks(lo, hi - lo)
    units = [orc.encode(blocks[i].tobytes()) for i in range(hi - lo)]
    sizes = np.array([len(u) for u in units], dtype=np.int64)
    # the only exchange of a sharded compress: per-shard totals -> exclusive scan -> offsets
    totals = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(totals, torch.tensor([int(sizes.sum())]))
    base = int(sum(int(t.item()) for t in totals[:rank]))
    crc = sum(orc.masked_crc(blocks[i].tobytes()) for i in range(hi - lo))
    total_crc = shard.sum_over_ranks(dist, crc)
    slowest = shard.max_over_ranks(dist, float(rank + 1))
    np.save(os.path.join(out_dir, "r%d.npy" % rank),
            np.array([lo, hi, base, int(sizes.sum()), total_crc, int(slowest)], dtype=np.int64))
    with open(os.path.join(out_dir, "r%d.bin" % rank), "wb") as fh:
        fh.write(b"".join(units))
    dist.destroy_process_group()


def test_two_rank_block_sharding(tmp_path, orc):
    import corpus
    world = 2
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    meta = [np.load(tmp_path / ("r%d.npy" % r)) for r in range(world)]
    # disjoint, complete, in rank order
    assert [int(m[0]) for m in meta] == [0, BLOCKS_PER_RANK]
    assert [int(m[1]) for m in meta] == [BLOCKS_PER_RANK, 2 * BLOCKS_PER_RANK]
    # single-process reference over the union
    blocks = corpus.make_blocks(0, world * BLOCKS_PER_RANK)
    units = [orc.encode(blocks[i].tobytes()) for i in range(len(blocks))]
    whole = b"".join(units)
    parts = [open(tmp_path / ("r%d.bin" % r), "rb").read() for r in range(world)]
    assert int(meta[0][2]) == 0 and int(meta[1][2]) == len(parts[0])  # scanned offsets
    assert b"".join(parts) == whole  # host-side concatenate == unsharded stream
    crc = sum(orc.masked_crc(blocks[i].tobytes()) for i in range(len(blocks)))
    assert all(int(m[4]) == crc for m in meta)  # checksum of checksums
    assert all(int(m[5]) == world for m in meta)  # MAX over ranks

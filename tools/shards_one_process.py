#!/usr/bin/env python3
"""BASELINE configs[4] from ONE process: block-range sharded compress over N GPUs with the host-side
concatenate (snappy_hip_compress_shards_staged): the blocks go round the GPUs in stages of --stage-blocks
(stage j of GPU k = blocks [(j n + k) S, (j n + k + 1) S)), a GPU encodes a stage while its earlier output travels
to its scanned offset in ONE page-locked host buffer (--stage-blocks 0: one contiguous shard per GPU, encode all,
then download).  Strong scaling: --total-gib is fixed and split N ways (the weak-scaling line is bench.py
--gpus N under torchrun; the multi-process variant of this tool is tools/sharded_compress.py).

    python tools/shards_one_process.py --gpus 8 --total-gib 32
    python tools/shards_one_process.py --gpus 2 --same-gpu --total-gib 1 --check   (a one-GPU box)

Prints one JSON line: GB/s of uncompressed bytes end to end (encode + pack + download), and the phases.
"""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch

BLOCK = 65536


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=torch.cuda.device_count())
    ap.add_argument("--total-gib", type=float, default=4.0)
    ap.add_argument("--raw", action="store_true", help="compress() format (total < 4 GiB) instead of compressFramed")
    ap.add_argument("--same-gpu", action="store_true", help="all contexts on GPU 0 (exercises the path on a one-GPU box)")
    ap.add_argument("--check", action="store_true", help="compare the whole stream with the CPU oracle (small totals)")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--stage-blocks", type=int, default=16384, help="blocks per stage and GPU (0: contiguous shards)")
    a = ap.parse_args()
    hip = importlib.import_module("nim-snappy_amd")
    import corpus
    import shard
    n = a.gpus
    nb_total = int(a.total_gib * (1 << 30)) // BLOCK
    devs = [0 if a.same_gpu else k for k in range(n)]
    ctxs = [hip.Context(d) for d in devs]
    S = min(a.stage_blocks, max(1, nb_total // (2 * n))) if a.stage_blocks else 0  # (at least two stages a GPU)
    d_ins, lens = [], []
    for k in range(n):
        if S:
            ranges = [(lo, min(nb_total, lo + S)) for lo in range(k * S, nb_total, n * S)]
        else:
            ranges = [shard.split_range(k, n, nb_total)]
        dev = torch.device("cuda", devs[k])
        parts = [corpus.make_blocks_torch(torch, b0, min(4096, hi - b0), dev).reshape(-1)
                 for lo, hi in ranges for b0 in range(lo, hi, 4096)]
        d_ins.append(torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device=dev))
        lens.append(sum(hi - lo for lo, hi in ranges) * BLOCK)
    total = sum(lens)
    cap = hip.max_compressed_len_framed(total) if not a.raw else hip.max_compressed_len(total)
    out = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
    ts = []
    for _ in range(a.reps + 1):
        for d in set(devs):
            torch.cuda.synchronize(d)
        t0 = time.perf_counter()
        written, offs = hip.compress_shards(ctxs, d_ins, lens, out.data_ptr(), cap, framed=not a.raw, stage_blocks=S)
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    line = {"tool": "shards_one_process", "n_gpus": n, "same_gpu": a.same_gpu, "format": "raw" if a.raw else "framed",
            "uncompressed_bytes": total, "stream_bytes": written, "seconds": round(t, 5),
            "GBps_uncompressed_end_to_end": round(total / t / 1e9, 2), "stage_blocks": S, "shard_offsets": offs,
            "scaling": "strong (fixed total split n ways)"}
    if a.check:
        import pyoracle as orc
        src = corpus.make_blocks(0, nb_total).tobytes()  # (the blocks in global order)
        want = orc.encode(src) if a.raw else orc.encode_framed(src)
        got = out[:written].numpy().tobytes()
        line["equals_oracle"] = got == want
        assert got == want, "sharded stream differs from the oracle's"
    print(json.dumps(line), flush=True)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()

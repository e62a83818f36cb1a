"""Encoder fuzz at scale: EVERY block of a batch of structured-fuzz blocks (tests/test_gpu_batch.py::_structured_block: random
bytes, repeats at all distances, runs, few-symbol stretches) and corpus blocks against the oracle (all host threads), byte
for byte -- with the encoder's second waves (table in global memory, encode_kernel.h) on every batch.
usage: SNAPPY_HIP_ENC_GWAVES=4,1 python tools/fuzz_encode.py <blocks> <seed> [<seed> ...]"""
import importlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
hip = importlib.import_module("nim-snappy_amd")
import pyoracle as orc
import corpus
from test_gpu_batch import _structured_block
nb = int(sys.argv[1])
slot = hip.SLOT_STRIDE
bad_total = 0
for seed in map(int, sys.argv[2:]):
    rng = random.Random(seed)
    n_fuzz = nb // 2
    blocks = [_structured_block(rng) for _ in range(n_fuzz)]
    # ragged lengths too: the last block of a buffer is short
    src = np.frombuffer(b"".join(blocks), dtype=np.uint8)
    cor = corpus.make_blocks(rng.randrange(1 << 20), nb - n_fuzz).reshape(-1)
    tail = rng.randrange(1, 65536)
    flat_np = np.concatenate([src, cor])[:(nb - 1) * 65536 + tail].copy()
    flat = torch.from_numpy(flat_np).cuda()
    ctx = hip.Context(0)
    d_slots = torch.empty(nb * slot, dtype=torch.uint8, device="cuda")
    d_sizes = torch.empty(nb, dtype=torch.int32, device="cuda")
    for unit in (hip.UNIT_RAW, hip.UNIT_FRAME):
        ctx.encode_blocks(flat, flat.numel(), d_slots, d_sizes, unit=unit)
        ctx.sync()
        sizes = d_sizes.cpu().numpy().astype(np.int64)
        got = d_slots.cpu().numpy().reshape(nb, slot)
        buf = np.empty(nb * slot, dtype=np.uint8)
        csz = np.empty(nb, dtype=np.uint32)
        fn = orc.lib.sor_compress_blocks_mt if unit == hip.UNIT_RAW else orc.lib.sor_encode_frames_mt
        fn(flat_np.ctypes.data, flat_np.size, 65536, buf.ctypes.data, slot, csz.ctypes.data, os.cpu_count() or 1)
        want = buf.reshape(nb, slot)
        bad = [i for i in range(nb) if int(csz[i]) != int(sizes[i]) or not np.array_equal(got[i, :sizes[i]], want[i, :csz[i]])]
        bad_total += len(bad)
        print("seed", seed, "unit", unit, "blocks", nb, "mismatches", bad[:8], "C/U %.3f" % (sizes.sum() / flat_np.size), flush=True)
    ctx.close()
sys.exit(1 if bad_total else 0)

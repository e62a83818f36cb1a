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
          `--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks ITSELF (a child
          `python -m torch.distributed.run`, before this process touches a GPU), relays rank 0's JSON
          line and exits with the child's code; under an external launcher it checks WORLD_SIZE == N.
sharded_compress: BASELINE configs[4] inside the same line -- a FIXED total (32 GiB, or what fits) split
          N ways by block range (tools/shard.split_range), every rank encodes + packs its shard, the N
          shard totals are exchanged and scanned (the reference's serial `written += ...`,
          snappy.nim:56-62,146-153), every shard is copied to its offset in ONE page-locked host buffer
          (a shared mapping all ranks register); strong_GBps = total bytes / slowest rank, and the stream's
          digest is compared with a single-GPU encoding of the same bytes.

roofline : block decode is HBM-bound byte work.  Dominant kernel = decode_indexed_kernel<16384> (pass 2
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


RING_KERNEL = "decode_indexed_kernel<16384>"  # (the ring-window instantiation: decode2_kernel.h, kRingWin)


def measured_traffic(nb, only, kernels=(RING_KERNEL, "decode_indexed_kernel<32768>", "decode_indexed_kernel")):
    """(HBM bytes per launch of a kernel, note) from the PMC passes of tools/profile_bench.sh (FETCH_SIZE / WRITE_SIZE
    cannot be read inside this process: they need their own rocprofv3 passes).  The committed measurement is for
    the default workload only, and for ONE state of the kernel sources: the file carries their sha256
    (tools/build_id.py), and a profile of other sources is not reported -- (None, why) instead."""
    if nb != 65536 or only is not None:
        return None, "the committed profile is of the default workload (65536 blocks, class mix)"
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
    if not cands:
        return None, "no profiles/*_traffic.json"
    import build_id
    mine = build_id.csrc_sha256(ROOT)
    stale = None
    for name in reversed(cands):  # the newest profile of THESE sources
        with open(os.path.join(ROOT, "profiles", name)) as f:
            t = json.load(f)
        if t.get("csrc_sha256") != mine:
            stale = stale or name
            continue
        for k in kernels:
            v = t.get("kernels", {}).get(k)
            if v is not None:
                return v["total_bytes"], "profiles/%s (csrc_sha256 %s)" % (name, mine[:16])
    return None, ("no profile of these kernel sources (csrc_sha256 %s); the newest, profiles/%s, is of %s: run "
                  "tools/profile_bench.sh" % (mine[:16], stale, "other sources" if stale else "nothing"))


STEP_KERNELS = ("index_units_kernel", RING_KERNEL, "decode_indexed_kernel<65536>")  # the kernels of one decode step


def measured_step_traffic(nb, only):
    """HBM bytes of one whole decode step (index pass + ring launch + the passed-on units' launch) from the same
    profile as measured_traffic, or None"""
    total = 0.0
    for k in STEP_KERNELS:
        v, _ = measured_traffic(nb, only, (k,))
        if v is None:
            return None
        total += v
    return total


def cpu_baseline(corpus, d_in, d_packed, offsets, sizes, n_sample, budget_s, nb_all):
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
    # SURVEY 8d "B2 ... all host cores": the oracle over every host CPU (native threads inside the oracle: runs
    # of 8 blocks from a shared counter), on a larger sample so that every thread has work; 64 threads beside it
    ncpu = os.cpu_count() or 1
    n_mt = min(nb_all, max(n_sample, 32768))
    src_mt = d_in[:n_mt * BLOCK].cpu().numpy()
    out2 = np.empty(n_mt * slot, dtype=np.uint8)
    csz2 = np.empty(n_mt, dtype=np.uint32)
    offs2 = (np.arange(n_mt, dtype=np.uint64) * slot)
    dec2 = np.empty(n_mt * BLOCK, dtype=np.uint8)

    def mt_rates(nthr):
        best_e, best_d = 0.0, 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            orc.lib.sor_compress_blocks_mt(src_mt.ctypes.data, src_mt.size, BLOCK, out2.ctypes.data, slot,
                                           csz2.ctypes.data, nthr)
            best_e = max(best_e, n_mt * BLOCK / (time.perf_counter() - t0) / 1e9)
        for _ in range(4):
            t0 = time.perf_counter()
            st = orc.lib.sor_uncompress_blocks_mt(out2.ctypes.data, offs2.ctypes.data, csz2.ctypes.data, n_mt,
                                                  dec2.ctypes.data, BLOCK, nthr)
            best_d = max(best_d, n_mt * BLOCK / (time.perf_counter() - t0) / 1e9)
            assert st == 0
        return round(best_d, 3), round(best_e, 3)

    # a sweep of thread counts: boxes with a CPU quota below their CPU count run slower with one thread per CPU
    sweep = {}
    for nthr in sorted({ncpu, max(1, ncpu // 2), min(64, ncpu), min(32, ncpu)}, reverse=True):
        sweep[nthr] = mt_rates(nthr)
        if nthr == ncpu:
            assert np.array_equal(csz2[:n_sample], csz) and np.array_equal(dec2, src_mt)
    all_d = max(v[0] for v in sweep.values())
    all_e = max(v[1] for v in sweep.values())
    t64_d, t64_e = sweep[min(64, ncpu)]
    return {
        "value": round(n_sample * BLOCK * passes / t_dec / 1e9, 4),
        "unit": "GB/s uncompressed (decompress)",
        "cores": 1,
        "kind": "port",
        "sample": "first %d blocks (%d MiB) of the same corpus, %d decode passes; "
                  "oracle/snappy_oracle.c -O3, one thread (threaded legs: first %d blocks, best of 3-4 passes)"
                  % (n_sample, n_sample // 16, passes, n_mt),
        "compress_value": round(n_sample * BLOCK / t_enc / 1e9, 4),
        # every host CPU: the best of a sweep of thread counts up to os.cpu_count(); 64 threads as a second figure
        "threads": ncpu,
        "threads_value": all_d,
        "compress_threads_value": all_e,
        "threads_sweep_decompress_compress_GBps": {str(k): list(v) for k, v in sweep.items()},
        # the sweep's spread (the box's CPU quota makes the legs noisy: quote the range, not only the best)
        "threads_value_min_max": [min(v[0] for v in sweep.values()), all_d],
        "compress_threads_value_min_max": [min(v[1] for v in sweep.values()), all_e],
        "threads64_value": t64_d,
        "compress_threads64_value": t64_e,
        "host_cpus": ncpu,
    }


class AbiCaller:
    """The reference's in-memory API through a C ABI on PREALLOCATED numpy buffers, the way a Nim caller holds its
    seqs (snappy.nim:66-82,118-128): no Python-side allocation or copy inside the timed call.  `lib` is the HIP
    library (snappy_hip_*) or the oracle (sor_*): same five-argument shape."""

    def __init__(self, lib, prefix, src, cap_raw, cap_framed):
        import ctypes
        self.ct = ctypes
        self.lib, self.prefix = lib, prefix
        self.src = np.frombuffer(src, dtype=np.uint8)
        self.raw = np.empty(cap_raw, dtype=np.uint8)
        self.fr = np.empty(cap_framed, dtype=np.uint8)
        self.back = np.empty(max(len(src), 1), dtype=np.uint8)
        self.w, self.r = ctypes.c_size_t(), ctypes.c_size_t()
        self.raw_len = self.fr_len = 0

    def _p(self, a):  # (an input pointer: both libraries declare it c_char_p)
        return self.ct.cast(self.ct.c_void_p(a.ctypes.data), self.ct.c_char_p)

    def _o(self, a):  # (an output pointer)
        return self.ct.c_void_p(a.ctypes.data)

    def encode(self):
        st = getattr(self.lib, self.prefix + "compress")(self._p(self.src), self.src.size, self._o(self.raw), self.raw.size,
                                                         self.ct.byref(self.w))
        assert st == 0, st
        self.raw_len = self.w.value

    def decode(self):
        st = getattr(self.lib, self.prefix + "uncompress")(self._p(self.raw), self.raw_len, self._o(self.back), self.src.size,
                                                           self.ct.byref(self.w))
        assert st == 0 and self.w.value == self.src.size, (st, self.w.value)

    def encode_framed(self):
        st = getattr(self.lib, self.prefix + "compress_framed")(self._p(self.src), self.src.size, self._o(self.fr), self.fr.size,
                                                                self.ct.byref(self.w))
        assert st == 0, st
        self.fr_len = self.w.value

    def decode_framed(self):
        st = getattr(self.lib, self.prefix + "uncompress_framed")(self._p(self.fr), self.fr_len, self._o(self.back),
                                                                  self.src.size, 1, 1, self.ct.byref(self.r), self.ct.byref(self.w))
        assert st == 0 and self.w.value == self.src.size, (st, self.w.value)

    def check_against(self, other):
        """same bytes out of both libraries, and the source back"""
        for leg in ("encode", "encode_framed"):
            getattr(self, leg)()
            getattr(other, leg)()
        assert self.raw_len == other.raw_len and np.array_equal(self.raw[:self.raw_len], other.raw[:other.raw_len])
        assert self.fr_len == other.fr_len and np.array_equal(self.fr[:self.fr_len], other.fr[:other.fr_len])
        for c in (self, other):
            c.back[:] = 0
            c.decode()
            assert np.array_equal(c.back[:c.src.size], c.src)
            c.back[:] = 0
            c.decode_framed()
            assert np.array_equal(c.back[:c.src.size], c.src)


def _callers(hip, orc, src):
    """(HIP, oracle) C-ABI callers over the same source bytes"""
    cap_raw = hip.max_compressed_len(len(src))
    cap_fr = hip.max_compressed_len_framed(len(src))
    h = AbiCaller(hip.lib, "snappy_hip_", src, cap_raw, cap_fr)
    o = AbiCaller(orc.lib, "sor_", src, cap_raw, cap_fr)
    h.check_against(o)
    return h, o


def config1_alice29(hip):
    """BASELINE configs[0]: tests/data/alice29.txt single-buffer encode / decode, timed the way the
    reference's harness does (tests/benchmark.nim:20-23,93-104: mean over 100 calls, ms per call) --
    the oracle on one host core (the Nim inMemory path cannot be built here), and the HIP library's
    host-buffer calls (PCIe and launch latency included: 3 blocks cannot fill a GPU).  Both through their C ABI on
    preallocated buffers (AbiCaller): what is timed is the library call, not Python's buffer handling."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    with open(os.path.join(ROOT, "tests", "golden", "data", "alice29.txt"), "rb") as fh:
        src = fh.read()

    def mean_ms(f, reps=100):
        f()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        return round((time.perf_counter() - t0) / reps * 1e3, 4)

    h, o = _callers(hip, orc, src)
    return {
        "file": "alice29.txt", "bytes": len(src), "calls": 100, "unit": "ms per call (encode / decode)",
        "through": "the C ABI on preallocated buffers (both libraries)",
        "oracle_inMemory_raw": [mean_ms(o.encode), mean_ms(o.decode)],
        "oracle_inMemory_framed": [mean_ms(o.encode_framed), mean_ms(o.decode_framed)],
        "hip_host_api_raw": [mean_ms(h.encode), mean_ms(h.decode)],
        "hip_host_api_framed": [mean_ms(h.encode_framed), mean_ms(h.decode_framed)],
        "reference_README_x86_64": {"raw": [0.334, 0.186], "framed": [0.382, 0.251]},
    }


README_TABLE = {  # /root/reference/README.md:99-124, inMemory column: ms per call, encode / decode (x86_64, one thread)
    "html": ((0.086, 0.056), (0.117, 0.093)), "urls.10K": ((1.052, 0.480), (1.260, 0.775)),
    "fireworks.jpeg": ((0.008, 0.005), (0.051, 0.047)), "paper-100k.pdf": ((0.010, 0.006), (0.046, 0.050)),
    "html_x_4": ((0.374, 0.218), (0.491, 0.386)), "alice29.txt": ((0.334, 0.186), (0.382, 0.251)),
    "asyoulik.txt": ((0.300, 0.165), (0.343, 0.220)), "lcet10.txt": ((0.907, 0.483), (1.053, 0.675)),
    "plrabn12.txt": ((1.241, 0.646), (1.387, 0.856)), "geo.protodata": ((0.076, 0.050), (0.110, 0.095)),
    "kppkn.gtb": ((0.279, 0.183), (0.346, 0.261)), "Mark.Twain-Tom.Sawyer.txt": ((0.024, 0.018), (0.030, 0.021)),
}
README_STATE = {"bytes": 38942424, "raw": (23.814, 8.608), "framed": (36.075, 25.389)}  # README.md:123-124 (50 calls)


def config_readme_files(hip, corpus, ctx, dev, calls=30):
    """The reference's own benchmark table (README.md:97-125, tests/benchmark.nim:93-126,178-180) on this box: every
    data file of the table, raw and framed, ms per call encode / decode, mean over `calls` calls the way
    benchmark.nim's timeit does -- the oracle on one host core (the Nim inMemory column cannot be built here)
    beside the HIP library's host-buffer calls (PCIe and launch latency included).  The table's last row is a
    38 942 424-byte beacon state that is not in the reference's tree: a synthetic buffer of that size (the bench's
    corpus mix), through the host calls and through the device-resident calls (p50 of 20)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc

    def mean_ms(f, reps):
        f()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        return round((time.perf_counter() - t0) / reps * 1e3, 4)

    def p50_ms(f, reps=20):
        f()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return round(sorted(ts)[len(ts) // 2] * 1e3, 4)

    rows = []
    for name, (ref_raw, ref_fr) in README_TABLE.items():
        with open(os.path.join(ROOT, "tests", "golden", "data", name), "rb") as fh:
            src = fh.read()
        h, o = _callers(hip, orc, src)  # (also checks: same bytes out of both, the source back)
        rows.append({
            "file": name, "bytes": len(src),
            "oracle_raw": [mean_ms(o.encode, calls), mean_ms(o.decode, calls)],
            "hip_host_raw": [mean_ms(h.encode, calls), mean_ms(h.decode, calls)],
            "oracle_framed": [mean_ms(o.encode_framed, calls), mean_ms(o.decode_framed, calls)],
            "hip_host_framed": [mean_ms(h.encode_framed, calls), mean_ms(h.decode_framed, calls)],
            "reference_README_inMemory": {"raw": list(ref_raw), "framed": list(ref_fr)},
        })
    # the 38.9 MB single buffer
    n = README_STATE["bytes"]
    nb = -(-n // BLOCK)
    d_src = corpus.make_blocks_torch(torch, 0, nb, dev).reshape(-1)[:n].contiguous()
    src = d_src.cpu().numpy().tobytes()
    h, o = _callers(hip, orc, src)
    enc, fr = h.raw[:h.raw_len].tobytes(), h.fr[:h.fr_len].tobytes()
    d_fr = torch.empty(hip.max_compressed_len_framed(n), dtype=torch.uint8, device=dev)
    d_back = torch.empty(n, dtype=torch.uint8, device=dev)
    flen = ctx.compress_framed(d_src, n, d_fr, d_fr.numel())
    assert flen == len(fr) and ctx.uncompress_framed(d_fr, flen, d_back, n) == (0, flen, n) and bool(torch.equal(d_back, d_src))
    d_raw = torch.frombuffer(bytearray(enc), dtype=torch.uint8).to(dev)
    assert ctx.uncompress(d_raw, len(enc), d_back, n) == (0, n) and bool(torch.equal(d_back, d_src))
    state = {
        "bytes": n, "data": "synthetic (corpus mix; the reference's state file is not in its tree)",
        "compressed_bytes": {"raw": len(enc), "framed": len(fr)},
        "oracle_raw": [mean_ms(o.encode, 3), mean_ms(o.decode, 3)],
        "oracle_framed": [mean_ms(o.encode_framed, 3), mean_ms(o.decode_framed, 3)],
        "hip_host_raw_p50": [p50_ms(h.encode), p50_ms(h.decode)],
        "hip_host_framed_p50": [p50_ms(h.encode_framed), p50_ms(h.decode_framed)],
        "hip_host_raw_GBps": None, "hip_host_framed_GBps": None,  # (filled in below)
        # input and output resident in HBM (snappy_hip_compress_framed_d / _uncompress_framed_d / _uncompress_d)
        "hip_device_framed_p50": [p50_ms(lambda: ctx.compress_framed(d_src, n, d_fr, d_fr.numel())),
                                  p50_ms(lambda: ctx.uncompress_framed(d_fr, flen, d_back, n))],
        "hip_device_raw_decode_p50": p50_ms(lambda: ctx.uncompress(d_raw, len(enc), d_back, n)),
        "reference_README_inMemory": {"raw": list(README_STATE["raw"]), "framed": list(README_STATE["framed"]),
                                      "note": "another buffer (a real beacon state), x86_64, one thread, 50 calls"},
    }
    for k in ("raw", "framed"):
        state["hip_host_%s_GBps" % k] = [round(n / (t * 1e-3) / 1e9, 2) for t in state["hip_host_%s_p50" % k]]
    return {"what": "README.md:97-125 / tests/benchmark.nim on this box: ms per call, [encode, decode]; both libraries "
                    "through their C ABI on preallocated buffers", "calls": calls, "files": rows, "state_38_9MB": state}


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

            t_ud = best(ud)
            res[tag] = round(part.size / t_ud / 1e9, 2)
            res[tag.replace("_GBps", "_ms")] = round(t_ud * 1e3, 4)
            res[tag.replace("_GBps", "_stream_bytes")] = rl
            res[tag.replace("_GBps", "_bytes")] = int(part.size)
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

        sp0, spn0 = ctx.kernel_ms(10)
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
        sp1, spn1 = ctx.kernel_ms(10)  # bytes / units the INDEX PASS decoded itself (few, long elements): a running count
        sparse_bytes = (sp1 - sp0) / 4 if nb else 0  # (4 decodes since sp0: one warm-up, three timed)
        dec_ms += dec2_ms
        assert bool(torch.equal(d_out, d_in)), cls
        u = nb * BLOCK
        # the class as ONE framed stream (configs[3]): compressFramed once, uncompressFramed timed over three calls
        del d_packed
        fcap = hip.max_compressed_len_framed(u)
        d_fs = torch.empty(fcap, dtype=torch.uint8, device=dev)
        flen = ctx.compress_framed(d_in, u, d_fs, fcap)
        d_out.zero_()
        assert ctx.uncompress_framed(d_fs, flen, d_out, u) == (0, flen, u), cls
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.uncompress_framed(d_fs, flen, d_out, u)
        t_f = (time.perf_counter() - t0) / 3
        assert bool(torch.equal(d_out, d_in)), cls
        del d_fs
        out[cls] = {
            "blocks": nb,
            "decompress_GBps": round(u / t / 1e9, 1),
            "decode_kernel_ms": round(dec_ms, 3), "passed_on_units_kernel_ms": round(dec2_ms, 3),
            # units of few, long elements are decoded inside the index pass (sparse_kernel.h): their bytes are not the
            # decode launches', and the fraction below is (the other units' bytes) / (the decode launches' time)
            "units_decoded_by_index_pass": int((spn1 - spn0) // 4), "their_bytes": int(sparse_bytes),
            "decode_step_frac_of_hbm_peak": round((u + tot) / ((dec_ms + idx_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if dec_ms else None,
            "index_pass_ms": round(idx_ms, 3),
            "decode_frac_of_hbm_peak": (round((u + tot - sparse_bytes) / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                                        if dec_ms and u + tot - sparse_bytes > 0.05 * (u + tot) else None),
            "compress_GBps": round(u / (enc_ms * 1e-3) / 1e9, 1) if enc_ms else None,
            "compress_kernel_ms": round(enc_ms, 3),
            "compressed_over_uncompressed": round(tot / u, 3),
            "framed_decompress_GBps": round(u / t_f / 1e9, 1), "framed_stream_bytes": int(flen),
            "framed_decompress_frac_of_hbm_peak": round((flen + u) / t_f / 1e9 / HBM_PEAK_GBPS, 4),
        }
        del d_in, d_out
    return out


def _tree_sha256(buf, n, piece=64 << 20, threads=16):
    """sha256 over the per-piece sha256 digests of buf[0:n] (pieces of 64 MiB of the STREAM, so the digest
    does not depend on how the stream was sharded); pieces are hashed on several threads."""
    import concurrent.futures as cf
    import hashlib
    mv = memoryview(buf)[:n]

    def one(i):
        return hashlib.sha256(mv[i:min(n, i + piece)]).digest()

    with cf.ThreadPoolExecutor(threads) as ex:
        digs = list(ex.map(one, range(0, max(n, 1), piece)))
    return hashlib.sha256(b"".join(digs)).hexdigest()


def _host_room_bytes():
    """What this process may take of host memory: MemAvailable, capped by the cgroup's limit."""
    room = None
    try:
        import psutil
        room = int(psutil.virtual_memory().available)
    except Exception:
        pass
    for f in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(f).read().strip()
            if v.isdigit():
                lim = int(v)
                used = 0
                for g in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
                    try:
                        used = int(open(g).read().strip())
                        break
                    except Exception:
                        pass
                room = min(room, lim - used) if room is not None else lim - used
        except Exception:
            pass
    try:
        st = os.statvfs("/dev/shm")
        shm = st.f_bavail * st.f_frsize
        room = min(room, shm) if room is not None else shm
    except Exception:
        pass
    return room if room is not None else 8 << 30


def sharded_compress_leg(hip, ctx, corpus, shard, rank, world, dev, backend, total_blocks, requested_blocks):
    """BASELINE configs[4]: block-range sharded compress of a FIXED corpus (strong scaling) with the host-side
    concatenate, in STAGES so that a GPU encodes while its earlier output travels.  With S = the stage size,
    rank r owns the global block ranges [(j N + r) S, (j N + r + 1) S), j = 0, 1, ... (resident in its HBM before the
    timed region): the stream is the stages in order, each the ranks in order -- the blocks in global order.  A rank
    encodes a stage as framed chunks (encodeFrame, encoder.nim:385-426) and packs it; the ONE exchange per stage is
    an all_gather of the N stage sizes, whose running sum (the reference's serial `written += ...`,
    snappy.nim:56-62,146-153) places the stage in the ONE host buffer -- a /dev/shm mapping every rank has
    page-locked -- and the rank's copy goes out on a second stream, over its own GPU's link, while the next stage is
    encoded.  No data-path collective.  (snappy_hip_compress_shards_staged is the same from one process.)"""
    import mmap
    dd = dist if world > 1 else None
    cdev = dev if backend == "nccl" else None
    S = max(1, min(16384, total_blocks // (2 * world)))  # stage: 1 GiB of input, at least two stages a rank
    n_stages = -(-total_blocks // (S * world))
    mine_b = []  # my block range of every stage (possibly empty in the last one)
    for j in range(n_stages):
        lo = min(total_blocks, (j * world + rank) * S)
        mine_b.append((lo, min(total_blocks, lo + S)))
    nbk = sum(hi - lo for lo, hi in mine_b)
    d_sh = torch.empty(max(nbk, 1) * BLOCK, dtype=torch.uint8, device=dev)
    at_b = 0
    for lo, hi in mine_b:
        for b0 in range(lo, hi, 4096):
            c = min(4096, hi - b0)
            d_sh[at_b * BLOCK:(at_b + c) * BLOCK] = corpus.make_blocks_torch(torch, b0, c, dev).reshape(-1)
            at_b += c
    d_slots = torch.empty(S * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev)
    d_sizes = torch.empty(S, dtype=torch.int32, device=dev)
    d_offsets = torch.empty(S + 1, dtype=torch.int64, device=dev)

    def exchange(size):
        """the one exchange of a stage: N sizes"""
        if world == 1:
            return [size]
        t = torch.tensor([size], dtype=torch.int64, device=cdev if cdev is not None else "cpu")
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        return [int(g.item()) for g in got]

    def run(d_packed, host_t, copy_stream):
        """every stage: encode + pack behind the stage before it, exchange the sizes, send the stage to its place in
        the host buffer (when there is one) on the copy stream; returns the table of sizes [stage][rank]"""
        table, at, in_b, base = [], 0, 0, 10
        for lo, hi in mine_b:
            c = hi - lo
            end = at
            if c:
                ctx.encode_blocks(d_sh[in_b * BLOCK:(in_b + c) * BLOCK], c * BLOCK, d_slots, d_sizes, unit=hip.UNIT_FRAME)
                if d_packed is None:  # sizing pass
                    ctx.sync()
                    end = at + int(d_sizes[:c].to(torch.int64).sum().item())
                else:
                    ctx.pack(d_slots, d_sizes, c, d_packed, d_offsets, base=at)
                    ctx.sync()
                    end = int(d_offsets[c].item())
            sizes = exchange(end - at)
            if host_t is not None and end > at:
                off = base + sum(sizes[:rank])
                with torch.cuda.stream(copy_stream):  # (the stage is complete: ctx.sync() above)
                    host_t[off:off + end - at].copy_(d_packed[at:end], non_blocking=True)
            base += sum(sizes)
            table.append(sizes)
            at, in_b = end, in_b + c
        if copy_stream is not None:
            copy_stream.synchronize()
        return table

    table0 = run(None, None, None)  # warm-up + sizes (the encoding is deterministic)
    mine = sum(row[rank] for row in table0)
    stream_len = 10 + sum(sum(row) for row in table0)
    totals0 = [sum(row[r] for row in table0) for r in range(world)]
    d_packed = torch.empty(mine + 64, dtype=torch.uint8, device=dev)
    # the ONE host buffer: a shared mapping, page-locked by every rank (the caller's output buffer of
    # snappy_hip_compress_shards_staged, here shared between processes)
    path = "/dev/shm/snappy_bench_%s_%d.bin" % (os.environ.get("MASTER_PORT", "0"), os.getppid() if world > 1 else os.getpid())
    fh = mm = None
    registered = []
    try:
        if rank == 0:
            with open(path, "wb") as f0:
                f0.truncate(stream_len)
        if world > 1:
            dist.barrier()
        fh = open(path, "r+b")
        mm = mmap.mmap(fh.fileno(), stream_len)
        if world > 1:
            dist.barrier()
        if rank == 0:  # every rank has it mapped: the name can go (nothing is left behind if a rank dies later)
            try:
                os.unlink(path)
            except OSError:
                pass
        host = np.frombuffer(mm, dtype=np.uint8)
        host_t = torch.from_numpy(host)
        # my pieces of it: touched (shm pages are allocated on first touch) and page-locked
        my_pieces, base = [], 10
        for row in table0:
            if row[rank]:
                my_pieces.append((base + sum(row[:rank]), row[rank]))
            base += sum(row)
        if world == 1:
            my_pieces = [(0, stream_len)]
        pinned = bool(my_pieces)
        for off, ln in my_pieces:
            host_t[off:off + ln:4096] = 0
            try:
                ok = int(torch.cuda.cudart().cudaHostRegister(host_t[off:].data_ptr(), ln, 0)) == 0
            except Exception:
                ok = False
            if ok:
                registered.append(host_t[off:].data_ptr())
            pinned = pinned and ok
        if rank == 0:
            host[:10] = np.frombuffer(bytes([0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59]), dtype=np.uint8)
        copy_stream = torch.cuda.Stream(device=dev)
        best = None
        for _ in range(2):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            table = run(d_packed, host_t, copy_stream)
            torch.cuda.synchronize()
            t_rank = time.perf_counter() - t0
            assert table == table0, "stage sizes changed between two encodings"
            if world > 1:
                dist.barrier()
            t = shard.max_over_ranks(dd, t_rank, cdev)
            best = t if best is None else min(best, t)
        res = _sharded_result(hip, ctx, corpus, rank, world, dev, total_blocks, requested_blocks, S, n_stages, table0,
                              totals0, stream_len, mine, d_packed, d_slots, d_sizes, d_offsets, host, mm, best, pinned,
                              lambda: run(d_packed, None, None))
    finally:
        for ptr in registered:
            try:
                torch.cuda.cudart().cudaHostUnregister(ptr)
            except Exception:
                pass
        host = host_t = None
        if mm is not None:
            try:
                mm.close()
            except BufferError:
                pass
        if fh is not None:
            fh.close()
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
    if world > 1:
        dist.barrier()
    del d_sh, d_slots, d_packed
    return res


def _sharded_result(hip, ctx, corpus, rank, world, dev, total_blocks, requested_blocks, S, n_stages, table0, totals0,
                    stream_len, mine, d_packed, d_slots, d_sizes, d_offsets, host, mm, best, pinned, encode_again):
    """rank 0: the stream's digest, the byte comparison with a single-GPU encoding, the result object"""
    PIECE = 65536  # blocks per encode launch of the comparison (4 GiB: the slots of one launch take 5 GB)
    pc = d_sizes.numel()
    ctx.timing(True)
    encode_again()
    enc_only = ctx.kernel_ms(1)[0]  # (average over the stages' launches)
    ctx.timing(False)
    if world > 1:
        dist.barrier()
    res = None
    if rank == 0:
        digest = _tree_sha256(mm, stream_len)
        # the same bytes from ONE GPU: this rank encodes every block range itself, piece by piece, and the packed
        # chunks are compared byte for byte with the host buffer (equal bytes: equal digests)

        def same(d_t, n, at):
            for c0 in range(0, n, 256 << 20):
                c1 = min(n, c0 + (256 << 20))
                if not np.array_equal(d_t[c0:c1].cpu().numpy(), host[at + c0:at + c1]):
                    return False
            return True

        if world == 1:
            equal = same(d_packed, mine, 10)  # (d_packed holds a later, separate encoding than the one copied out)
        else:
            equal, at = True, 10
            d_tmp = torch.empty(min(PIECE, total_blocks) * BLOCK, dtype=torch.uint8, device=dev)
            for b0 in range(0, total_blocks, PIECE):
                c = min(PIECE, total_blocks - b0)
                for g0 in range(0, c, 4096):
                    g = min(4096, c - g0)
                    d_tmp[g0 * BLOCK:(g0 + g) * BLOCK] = corpus.make_blocks_torch(torch, b0 + g0, g, dev).reshape(-1)
                big = c > pc
                sl = torch.empty(c * hip.SLOT_STRIDE, dtype=torch.uint8, device=dev) if big else d_slots
                sz = torch.empty(c, dtype=torch.int32, device=dev) if big else d_sizes
                of = torch.empty(c + 1, dtype=torch.int64, device=dev) if big else d_offsets
                ctx.encode_blocks(d_tmp[:c * BLOCK], c * BLOCK, sl, sz, unit=hip.UNIT_FRAME)
                ctx.sync()
                tot = int(sz[:c].to(torch.int64).sum().item())
                d_pk = torch.empty(tot + 64, dtype=torch.uint8, device=dev)
                ctx.pack(sl, sz, c, d_pk, of, base=0)
                ctx.sync()
                equal = equal and at + tot <= stream_len and same(d_pk, tot, at)
                at += tot
                del d_pk, sl, sz, of
            equal = equal and at == stream_len
            del d_tmp
        if not np.array_equal(host[:10], np.frombuffer(bytes([0xff, 6, 0, 0, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59]), dtype=np.uint8)):
            equal = False
        u = total_blocks * BLOCK
        res = {
            "what": "BASELINE configs[4]: block-range sharded compressFramed in stages (stage j of rank r = blocks "
                    "[(j N + r) S, (j N + r + 1) S)): N stage sizes -> host scan -> the stage copied to its offset in ONE "
                    "page-locked host buffer on a second stream while the next stage is encoded; fixed total (strong scaling)",
            "total_blocks": total_blocks,
            "uncompressed_bytes": u,
            "reduced_from_blocks": requested_blocks if requested_blocks != total_blocks else None,
            "stream_bytes": stream_len,
            "n_shards": world,
            "stage_blocks": S,
            "n_stages": n_stages,
            "shard_bytes": totals0,
            "stage0_sizes": table0[0],
            "seconds": round(best, 5),
            "strong_GBps": round(u / best / 1e9, 3),
            "encode_kernel_ms_per_stage_rank0": round(enc_only, 3),
            "host_buffer_page_locked": pinned,
            "stream_sha256_tree64MiB": digest,  # sha256 of the sha256 digests of the stream's 64 MiB pieces
            # a single-GPU encoding of all blocks on rank 0, compared byte for byte with the host buffer
            "equals_single_gpu_sha256": bool(equal),
        }
        if not res["equals_single_gpu_sha256"]:
            raise SystemExit("bench: the sharded stream differs from the single-GPU stream")
    return res


def self_launch(args, argv):
    """--gpus N without a launcher: start the N ranks as a CHILD (this process does not touch a GPU: the devices
    are counted from the kernel driver's topology files, not through the runtime), relay rank 0's JSON line, exit
    with the child's code."""
    import socket
    import subprocess

    def gpus_in_sysfs():
        """GPU nodes of /sys/class/kfd/kfd/topology (simd_count > 0; CPUs have 0); None if it cannot be read"""
        top = "/sys/class/kfd/kfd/topology/nodes"
        try:
            n = 0
            for node in os.listdir(top):
                with open(os.path.join(top, node, "properties")) as fh:
                    props = dict(ln.split()[:2] for ln in fh if len(ln.split()) >= 2)
                n += 1 if int(props.get("simd_count", "0")) > 0 else 0
            return n
        except (OSError, ValueError):
            return None

    # (a clear refusal instead of N - M ranks dying in set_device; unknown count: the child's exit code speaks)
    have = gpus_in_sysfs()
    if have is not None and have < args.gpus and not os.environ.get("BENCH_SHARE_DEVICE"):
        sys.stderr.write("bench.py: --gpus %d, but this node shows %d GPU(s)\n" % (args.gpus, have))
        sys.exit(2)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    sys.exit(proc.returncode if proc.returncode else (0 if lines else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=65536, help="blocks per GPU (4 GiB)")
    ap.add_argument("--only", default=None, help="one corpus class (per-class numbers)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-blocks", type=int, default=8192)
    ap.add_argument("--shard-gib", type=float, default=None,
                    help="fixed total of the sharded-compress leg (default 32 GiB = 8 x --blocks; 0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
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
    if hip.LIB_OVERRIDDEN and not os.environ.get("BENCH_ALLOW_LIBRARY_OVERRIDE"):
        # (SNAPPY_HIP_LIBRARY is a test / A-B hook: a fault-injection build must not be measured by accident)
        print("bench: SNAPPY_HIP_LIBRARY is set (%s); set BENCH_ALLOW_LIBRARY_OVERRIDE=1 to measure a variant library"
              % hip.LIB_PATH, file=sys.stderr)
        sys.exit(2)
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
    fenc_ts = []
    for _ in range(max(1, min(3, args.steps))):
        t0 = time.perf_counter()
        flen = ctx.compress_framed(d_in, nb * BLOCK, d_fstream, fcap)  # (returns the length: synchronous)
        fenc_ts.append(time.perf_counter() - t0)
    t_fenc = sum(fenc_ts) / len(fenc_ts)
    d_fout = torch.empty(nb * BLOCK, dtype=torch.uint8, device=dev)
    st_f = ctx.uncompress_framed(d_fstream, flen, d_fout, nb * BLOCK)  # warm
    assert st_f == (0, flen, nb * BLOCK), st_f
    ctx.timing(True)
    torch.cuda.synchronize()
    fdec_ts = []  # every call timed on its own (a call returns the verdict: it is synchronous): mean and min over --steps
    for _ in range(max(1, args.steps)):
        t0 = time.perf_counter()
        st_f = ctx.uncompress_framed(d_fstream, flen, d_fout, nb * BLOCK)
        fdec_ts.append(time.perf_counter() - t0)
        assert st_f == (0, flen, nb * BLOCK), st_f
    t_fdec = sum(fdec_ts) / len(fdec_ts)
    walk_ms, _ = ctx.kernel_ms(6)
    ctx.timing(False)
    assert bool(torch.equal(d_fout, d_in)), "framed round trip differs"
    flen_stream = int(flen)
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
    sp0, spn0 = ctx.kernel_ms(10)  # (running counts: bytes / units the index pass decoded itself)
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
    sp1, spn1 = ctx.kernel_ms(10)
    ctx.timing(False)
    # units of few, long elements are decoded inside the index pass (sparse_kernel.h): per step, their bytes and number
    sparse_bytes = (sp1 - sp0) / max(args.steps, 1)
    sparse_units = int((spn1 - spn0) // max(args.steps, 1))
    elapsed = shard.max_over_ranks(dist if world > 1 else None, elapsed, dev)
    # the side numbers are whole-job rates too: all ranks' bytes over the slowest rank's time
    t_enc = shard.max_over_ranks(dist if world > 1 else None, t_enc, dev)
    t_fenc = shard.max_over_ranks(dist if world > 1 else None, t_fenc, dev)

    # results must be right, or the number is void
    assert int((d_status != 0).sum().item()) == 0, "decode reported errors"
    assert bool(torch.equal(d_out, d_in)), "decoded bytes differ from the corpus"

    t_fdec = shard.max_over_ranks(dist if world > 1 else None, t_fdec, dev)
    t_fdec_min = shard.max_over_ranks(dist if world > 1 else None, min(fdec_ts), dev)

    # ---- measured copy bandwidth of this box: the second roofline denominator of SURVEY 8(d) -------
    torch.cuda.synchronize()
    d_out.copy_(d_in)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        d_out.copy_(d_in)
    torch.cuda.synchronize()
    copy_gbps = 3 * 2 * nb * BLOCK / (time.perf_counter() - t0) / 1e9

    # ---- BASELINE configs[4]: fixed-total block-range sharded compress with the host concatenate -------------
    sharded = None
    want_blocks = 8 * nb if args.shard_gib is None else int(args.shard_gib * (1 << 30)) // BLOCK
    if want_blocks > 0 and args.only is None:
        total_blocks = max(world, want_blocks)
        free_b, _ = torch.cuda.mem_get_info(dev)
        room = _host_room_bytes()
        # The total is FIXED (configs[4] is strong scaling): a box that cannot hold it does not get a smaller one
        # silently -- the leg is refused, with the reason in the line (and on stderr); `--shard-gib` names another total.
        per_rank = (total_blocks + world - 1) // world * BLOCK
        gpu_need = per_rank * 1.6 + min(65536, per_rank // BLOCK) * hip.SLOT_STRIDE + (4 << 30)
        if os.environ.get("BENCH_SHARE_DEVICE"):
            gpu_need *= world
        host_need = 0.6 * total_blocks * BLOCK * 2.5 + (2 << 30)
        fits = 1 if (gpu_need <= free_b and host_need <= room) else 0
        if world > 1:  # (every rank must come to the same decision)
            t = torch.tensor([fits], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            fits = int(t.item())
        if fits:
            sharded = sharded_compress_leg(hip, ctx, corpus, shard, rank, world, dev, backend, total_blocks, want_blocks)
        else:
            why = ("%.1f GiB in %d shard(s) needs %.0f GB of device memory per rank (free: %.0f) and %.0f GB of host memory "
                   "for the one host buffer, /dev/shm included (room: %.0f): not run; --shard-gib names a smaller fixed total"
                   % (total_blocks * BLOCK / 2**30, world, gpu_need / 1e9, free_b / 1e9, host_need / 1e9, room / 1e9))
            if rank == 0:
                print("bench: sharded_compress refused: " + why, file=sys.stderr)
            sharded = {"refused": why, "total_blocks": total_blocks, "n_shards": world}

    give_ups = int(ctx.kernel_ms(9)[0])
    if give_ups:
        raise SystemExit("bench: the indexed decoder gave up on %d turns (each hands its unit to the one-pass "
                         "kernel at ~30 ms): the number is void" % give_ups)

    if rank == 0:
        u_bytes = nb * BLOCK
        value = world * u_bytes * args.steps / elapsed / 1e9
        # the bytes of the units the two decode launches decode (the ring-window one and the one over the units it passes
        # on) over their durations: the ring kernel alone does not move all of them, so its duration alone would flatter
        # it -- and the units the index pass decodes itself (few, long elements) are not theirs at all
        achieved = (sum_c + u_bytes - sparse_bytes) / ((dec_ms + dec2_ms) * 1e-3) / 1e9 if dec_ms > 0 else 0.0
        step_achieved = (sum_c + u_bytes) / ((dec_ms + dec2_ms + idx_ms) * 1e-3) / 1e9 if dec_ms > 0 else 0.0
        step_traffic = measured_step_traffic(nb, args.only)
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
                "traffic": measured_traffic(nb, args.only)[0],
                "traffic_source": measured_traffic(nb, args.only)[1],  # (null traffic: why)
                # the whole step (index pass + both decode launches): every unit's bytes over the three kernels' time, the
                # PMC bytes of the three, and how much more than the algorithmic bytes that is
                "frac_step": round(step_achieved / HBM_PEAK_GBPS, 5),
                "traffic_step": step_traffic,
                "traffic_over_algorithmic": round(step_traffic / (sum_c + u_bytes), 4) if step_traffic else None,
                "kernel": RING_KERNEL,  # (ring window; <65536> takes the units it passes on)
                "kernel_ms": round(dec_ms, 4),  # (HIP events; rocprof's average for this kernel agrees)
                "kernel_ms_both_decode_launches": round(dec_ms + dec2_ms, 4),  # what `achieved` divides by
                "units_decoded_by_index_pass": sparse_units,  # (one literal / one period / few long elements: sparse_kernel.h)
                "their_bytes_per_launch": int(sparse_bytes),   # ... not in `achieved`'s numerator
                # every unit's bytes over all three kernels of a step (index pass + both decode launches)
                "step_achieved": round(step_achieved, 2), "step_frac": round(step_achieved / HBM_PEAK_GBPS, 5),
                "index_pass_kernel_ms": round(idx_ms, 4),
                "passed_on_units_kernel_ms": round(dec2_ms, 4),
                # turns the indexed decoder gave up on after its bounded wait (must be 0; each costs ~30 ms and
                # hands its unit to the one-pass kernel)
                "decode_turns_given_up": give_ups,
                "launches": dec_launches,
                "algorithmic_bytes_per_launch": int(sum_c + u_bytes - sparse_bytes),
                "algorithmic_bytes_per_step": sum_c + u_bytes,
            },
            "compress_GBps": round(world * u_bytes / t_enc / 1e9, 3),
            "compress_kernel_ms": round(enc_ms, 3),
            # the encode kernel against the same roofline (compress = U + C algorithmic bytes)
            "roofline_compress": {
                "bound": "hbm",
                "achieved": round((u_bytes + sum_c) / (enc_ms * 1e-3) / 1e9, 2) if enc_ms else 0.0,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round((u_bytes + sum_c) / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if enc_ms else 0.0,
                "traffic": measured_traffic(nb, args.only, ("encode_blocks_kernel",))[0],
                "kernel": "encode_blocks_kernel",
                "kernel_ms": round(enc_ms, 4),
                "algorithmic_bytes_per_launch": sum_c + u_bytes,
            },
            "library": {"path": os.path.relpath(hip.LIB_PATH, ROOT), "overridden": bool(hip.LIB_OVERRIDDEN)},
            "sharded_compress": sharded,
            # compress + decompress of the same bytes, one after the other (BASELINE's metric names both)
            "roundtrip_GBps": round(world * u_bytes / (t_enc + elapsed / args.steps) / 1e9, 3),
            # one genuine framed stream (10-byte identifier + one chunk per block), stream and output in HBM
            "framed_compress_GBps": round(world * u_bytes / t_fenc / 1e9, 3),
            "framed_decompress_GBps": round(world * u_bytes / t_fdec / 1e9, 3),  # (mean over the timed calls)
            "framed_decompress_best_GBps": round(world * u_bytes / t_fdec_min / 1e9, 3),
            "framed_decompress_calls": len(fdec_ts), "framed_compress_calls": len(fenc_ts),
            "framed_decompress_over_value": round((world * u_bytes / t_fdec / 1e9) / value, 4),
            "framed_chunk_walk_ms": round(walk_ms, 3),
        }
        if world == 1 and not args.no_cpu:
            offs = d_offsets.cpu().numpy()
            sizes = d_sizes.cpu().numpy()
            ns = min(args.cpu_blocks, nb)
            line["cpu_baseline"] = cpu_baseline(corpus, d_in, d_packed, offs, sizes, ns, 10.0, nb)
            cb = line["cpu_baseline"]
            line["gpu_over_all_host_cpus"] = {
                "compress": round(line["compress_GBps"] / cb["compress_threads_value"], 2),
                "decompress": round(line["value"] / cb["threads_value"], 2), "host_cpus": cb["host_cpus"]}
            line["config1_alice29"] = config1_alice29(hip)
            line["config_readme_files"] = config_readme_files(hip, corpus, ctx, dev)
            del d_packed, d_out
            line["per_class"] = per_class_rates(hip, corpus, ctx, dev, min(nb, 16384))  # (16 waves of workgroups of the ring kernel)
            line["host_api"] = host_api_rates(hip, d_in[:min(nb, 16384) * BLOCK].cpu().numpy(), ctx, dev)
            # (one raw multi-block buffer resident in HBM, snappy.nim:84-110 on the device: also at the line's top level)
            for k in ("raw_buffer_uncompress_d_GBps", "raw_buffer_64MiB_uncompress_d_GBps"):
                if k in line["host_api"]:
                    line[k] = line["host_api"][k]
        # ---- every number README.md quotes, INSIDE `roofline`: the driver's record keeps that dict whole and cuts the rest
        # of the line to key names.  Each: the rate in GB/s of uncompressed bytes, and the algorithmic bytes (SURVEY 8d:
        # stream + uncompressed, both directions) over the same time as a fraction of the HBM peak.
        rf = line["roofline"]
        t_step = elapsed / args.steps

        def entry(value_gbps, alg_bytes, seconds, **more):
            d = {"value": round(value_gbps, 2), "unit": "GB/s uncompressed",
                 "frac": round(alg_bytes / seconds / 1e9 / HBM_PEAK_GBPS, 5), "ms": round(seconds * 1e3, 4)}
            d.update(more)
            return d

        rf["decompress_step"] = entry(value, world * (sum_c + u_bytes), t_step)
        rf["compress"] = entry(line["compress_GBps"], world * (sum_c + u_bytes), t_enc, kernel="encode_blocks_kernel",
                               kernel_ms=round(enc_ms, 4))
        rf["round_trip"] = entry(line["roundtrip_GBps"], 2 * world * (sum_c + u_bytes), t_enc + t_step)
        # BASELINE's north star: block decompress of ONE 4 GiB framed stream (configs[3]), CRC32C of every chunk verified
        rf["framed"] = entry(line["framed_decompress_GBps"], world * (flen_stream + u_bytes), t_fdec, achieved=round(
            world * (flen_stream + u_bytes) / t_fdec / 1e9, 2), stream_bytes=flen_stream, calls=len(fdec_ts),
            best_value=line["framed_decompress_best_GBps"], over_value=line["framed_decompress_over_value"])
        rf["framed_compress"] = entry(line["framed_compress_GBps"], world * (flen_stream + u_bytes), t_fenc)
        if "per_class" in line:
            # configs[1] read literally -- "synthetic blocks (tests/randgen.nim)" is uniform random bytes, class R: one
            # literal a block -- beside the survey's harder mix that `value` is quoted on
            for cls, tag in (("R", "class_R"), ("T_TEXT", "class_T_TEXT"), ("T_HTML", "class_T_HTML")):
                pc = line["per_class"].get(cls)
                if pc:
                    u_c = pc["blocks"] * BLOCK
                    c_c = pc["compressed_over_uncompressed"] * u_c
                    rf[tag] = {"value": pc["decompress_GBps"], "unit": "GB/s uncompressed", "blocks": pc["blocks"],
                               "frac": round((u_c + c_c) * pc["decompress_GBps"] / u_c / HBM_PEAK_GBPS, 5),
                               "framed_value": pc["framed_decompress_GBps"],
                               "framed_frac": pc["framed_decompress_frac_of_hbm_peak"],
                               "compress_value": pc["compress_GBps"]}
        if "host_api" in line:
            ha = line["host_api"]
            for tag, key in (("raw_buffer_1GiB", "raw_buffer_uncompress_d"), ("raw_buffer_64MiB", "raw_buffer_64MiB_uncompress_d")):
                if key + "_GBps" in ha:
                    rf[tag] = entry(ha[key + "_GBps"], ha[key + "_stream_bytes"] + ha[key + "_bytes"], ha[key + "_ms"] * 1e-3,
                                    bytes=ha[key + "_bytes"])
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Lane-level model of encode_kernel.h's round structure (dense multi-match rounds).  Not a test by
itself: tests/test_encode_model.py runs it against the oracle.  It exists so that the exactness
argument of the kernel (which table writes a round may leave behind, when a round has to stop)
can be checked on the CPU, byte for byte, before any GPU time is spent.

The model follows the kernel step for step with 64 "lanes" and 64-bit masks:

  fresh round (idx0 == 0): lane i <-> position base + i (consecutive positions).  Every lane
    hashes its position, reads the table (all reads before all writes) and takes as candidate the
    nearest earlier lane of the round with the same slot, else the table value -- right if every
    earlier lane were inserted.  The chain then walks the round in the order of the sequential
    loop (encoder.nim:255-383): insert ip-1, copy-loop probe at ip, scan probes ip+1.. with the
    reference's step pattern, and at a match jumps to the copy's end INSIDE the round.  S collects
    the lanes the sequential loop really inserted; a probe whose nearest earlier same-slot lane is
    not in S saw a wrong candidate, so the round stops in front of it.  At the end the table is
    left as S alone would have left it.
  continuing round (idx0 > 0): lanes are the next 64 probes of a long scan (sparse positions),
    one match ends the round.
"""
M64 = (1 << 64) - 1
PAT = 0x55555555FFFFFFFF  # offsets of a scan's first 48 probes from its start: 0..31, 32, 34, .. 62


def _ctz(x):
    return (x & -x).bit_length() - 1


def _ld32(b, p):
    return int.from_bytes(b[p:p + 4], "little")


def probe_sequence(count=320):
    off, step, skip, o = [], [], 32, 0
    for _ in range(count):
        st = skip >> 5
        step.append(st)
        off.append(o)
        o += st
        skip += st
    return off, step


SEQ_OFF, SEQ_STEP = probe_sequence()


def _emit_literal(out, data, a, b):
    n = b - a - 1
    if n < 60:
        out.append(n << 2)
    elif n < 256:
        out += bytes([60 << 2, n])
    else:
        out += bytes([61 << 2, n & 255, n >> 8])
    out += data[a:b]


def _emit_copy(out, offset, length):
    while length >= 68:
        out += bytes([(63 << 2) | 2, offset & 255, offset >> 8])
        length -= 64
    if length > 64:
        out += bytes([(59 << 2) | 2, offset & 255, offset >> 8])
        length -= 60
    if length >= 12 or offset >= 2048:
        out += bytes([((length - 1) << 2) | 2, offset & 255, offset >> 8])
    else:
        out += bytes([((offset >> 8) << 5) | ((length - 4) << 2) | 1, offset & 255])


def _match_len(data, a, b, n):
    k = 0
    while b + k < n and data[a + k] == data[b + k]:
        k += 1
    return k


def encode_block(data, stats=None, wide=None):
    data = bytes(data)
    n = len(data)
    out = bytearray()
    if n < 17:
        if n:
            _emit_literal(out, data, 0, n)
        return bytes(out)
    ts = 256
    while ts < 16384 and ts < n:
        ts <<= 1
    mask = ts - 1
    table = [0] * ts
    ip_limit = n - 15

    def hsh(u):
        return (((u * 0x1e35a7bd) & 0xffffffff) >> 18) & mask

    has0, next_emit, s0, idx0 = False, 0, 1, 0
    tail_from = None
    rounds = dense_rounds = 0
    while tail_from is None:
        rounds += 1
        if idx0 == 0 and has0 and wide is not None:
            # encode2_kernel.h: a WIDE round (tools/encode2_model.py) takes this place when it can
            r = wide(table, hsh, s0 - 2, ip_limit, out)
            if r is not None:
                ended, tf, nstate = r
                if ended:
                    tail_from = tf
                else:
                    has0, s0, idx0, next_emit = nstate
                continue
        if idx0 == 0:
            dense_rounds += 1
            base = s0 - 2 if has0 else 0
            p = [base + i for i in range(64)]
            active = [p[i] <= ip_limit and (has0 or i > 0) for i in range(64)]
            d = [_ld32(data, p[i]) if active[i] else 0 for i in range(64)]
            h = [hsh(d[i]) for i in range(64)]
            old = [table[h[i]] if active[i] else 0 for i in range(64)]
            dep = [64] * 64
            grp = [1 << i for i in range(64)]
            last = {}
            for i in range(64):
                if not active[i]:
                    continue
                if h[i] in last:
                    dep[i] = last[h[i]]
                last[h[i]] = i
            groups = {}
            for i in range(64):
                if active[i]:
                    groups[h[i]] = groups.get(h[i], 0) | (1 << i)
            for i in range(64):
                if active[i]:
                    grp[i] = groups[h[i]]
            cand = [p[dep[i]] if dep[i] < 64 else old[i] for i in range(64)]
            conf = sum(1 << i for i in range(64) if dep[i] < 64)
            m4 = 0
            eq = [0] * 64
            for i in range(64):
                if active[i] and _ld32(data, cand[i]) == d[i]:
                    m4 |= 1 << i
                    k = 0
                    while k < 16 and p[i] + k < n and data[cand[i] + k] == data[p[i] + k]:
                        k += 1
                    eq[i] = k
            L1 = sum(1 << i for i in range(64) if p[i] + 1 <= ip_limit and active[i])
            L2 = sum(1 << i for i in range(64) if p[i] + 2 <= ip_limit and active[i])

            S = MS = COVER = 0
            lens = [0] * 64
            first = True
            e = 1
            lo = 1 if has0 else 0
            ended = False  # block ends inside this round
            nstate = None
            while True:
                ops = has0 or not first
                if ops:
                    if e > 62:
                        nstate = (True, base + e + 1, 0, base + e)
                        break
                    opA, opB, ls = 1 << (e - 1), 1 << e, e + 1
                else:
                    opA, opB, ls = 0, 0, 1
                probes = (PAT << ls) & M64
                VS = (((0xFFFFFFFF << ls) & L1) | ((0x5555555500000000 << ls) & L2)) & M64
                hit = (probes & ~VS) != 0
                pm_mask = (opB | VS) & m4
                bad = 0
                if conf:
                    T = S | opA | opB | VS
                    for j in range(64):
                        if ((opB | VS) >> j) & 1 and dep[j] < 64 and not (T >> dep[j]) & 1:
                            bad |= 1 << j
                m = _ctz(pm_mask) if pm_mask else None
                fb = _ctz(bad) if bad else None
                if fb is not None and (m is None or fb <= m):
                    if first:
                        below = (1 << fb) - 1
                        S |= (opA | opB | VS) & below
                        nstate = (False, s0, bin(VS & below).count("1"), next_emit)
                    else:
                        nstate = (True, base + e + 1, 0, base + e)
                    break
                if m is None:
                    if first:
                        S |= opA | opB | VS
                        if hit:
                            ended = True
                            tail_from = next_emit
                        else:
                            nstate = (False, s0, bin(VS).count("1"), next_emit)
                    else:
                        nstate = (True, base + e + 1, 0, base + e)
                    break
                S |= opA | opB | (VS & ((2 << m) - 1))
                pm, c, L = p[m], cand[m], eq[m]
                if L == 16 and pm + 16 < n:
                    L = _match_len(data, c, pm, n)
                MS |= 1 << m
                lens[m] = L
                COVER |= (((1 << L) - 1) << m) & M64
                e = m + L
                first = False
                if base + e > ip_limit:
                    ended = True
                    tail_from = base + e
                    break
            # ---- the table as S alone leaves it --------------------------------------------------
            for i in range(64):
                if not active[i]:
                    continue
                gs = grp[i] & S
                if gs == 0:
                    table[h[i]] = old[i]
                elif gs.bit_length() - 1 == i:
                    table[h[i]] = p[i]
            # ---- this round's elements, position-parallel -------------------------------------
            if MS:
                hi = min(e, 64)
                LIT = sum(1 << i for i in range(lo, hi) if not (COVER >> i) & 1)
                sizes, chunks = [0] * 64, [b""] * 64
                # A last copy of more than 64 bytes is several elements (emitCopy, encoder.nim:97-125).  Like the kernel's
                # common-round loop: lane mlast + j emits element j as a copy of at most 64 bytes that "starts" there (the
                # lanes behind mlast lie inside the copy and have nothing else to emit); more elements than lanes left:
                # the one call of _emit_copy below (the kernel's slow drain).
                elen, eoff, EMS = list(lens), [p[i] - cand[i] for i in range(64)], MS
                mlast = MS.bit_length() - 1
                if lens[mlast] > 64:
                    ll = lens[mlast]
                    k64 = (ll - 68) // 64 + 1 if ll >= 68 else 0
                    rem = ll - 64 * k64
                    has60 = 1 if rem > 64 else 0
                    nel = k64 + has60 + 1
                    if mlast + nel <= 64:
                        if stats is not None:
                            stats["long_split"] = stats.get("long_split", 0) + 1
                        for j in range(nel):
                            elen[mlast + j] = 64 if j < k64 else (60 if j == k64 and has60 else rem - 60 * has60)
                            eoff[mlast + j] = p[mlast] - cand[mlast]
                            assert not (LIT >> (mlast + j)) & 1
                        EMS |= ((1 << nel) - 1) << mlast
                for i in range(64):
                    if (LIT >> i) & 1:
                        rs = i == 0 or not (LIT >> (i - 1)) & 1
                        b = bytearray()
                        if rs:
                            rl = _ctz(~(LIT >> i) & M64)
                            if rl <= 60:
                                b.append((rl - 1) << 2)
                            else:
                                b += bytes([60 << 2, rl - 1])
                        b.append(data[p[i]])
                        chunks[i] = bytes(b)
                    elif (EMS >> i) & 1:
                        b = bytearray()
                        _emit_copy(b, eoff[i], elen[i])
                        chunks[i] = bytes(b)
                for i in range(64):
                    out += chunks[i]
            if not ended:
                has0, s0, idx0, next_emit = nstate
            continue
        # ---- continuing round: the next 64 probes of a long scan --------------------------------
        p, valid = [0] * 64, [False] * 64
        for i in range(64):
            si = idx0 + i
            if si < len(SEQ_OFF):
                p[i] = s0 + SEQ_OFF[si]
                valid[i] = p[i] + SEQ_STEP[si] <= ip_limit
        if not any(valid):
            tail_from = next_emit
            break
        win = None
        snapshot = []
        for i in range(64):
            if not valid[i]:
                break
            dd = _ld32(data, p[i])
            hh = hsh(dd)
            c = table[hh]
            table[hh] = p[i]
            if _ld32(data, c) == dd:
                win = (i, c)
                break
        if win is None:
            if all(valid):
                idx0 += 64
                continue
            tail_from = next_emit
            break
        i, c = win
        pm = p[i]
        L = _match_len(data, c, pm, n)
        if pm > next_emit:
            _emit_literal(out, data, next_emit, pm)
        _emit_copy(out, pm - c, L)
        ip = pm + L
        if ip > ip_limit:
            tail_from = ip
            break
        has0, s0, idx0, next_emit = True, ip + 1, 0, ip
    if tail_from < n:
        _emit_literal(out, data, tail_from, n)
    if stats is not None:
        stats["rounds"] = rounds
        stats["dense_rounds"] = dense_rounds
    return bytes(out)


# ---- round 5: the second wave's scratch (encode_kernel.h, encode_one_block<true>) ------------------------------------
# Its table lies in global memory, so the lanes of a round that share a table slot find each other in LDS: a scratch of
# 1 024 entries indexed by the hash's low ten bits, through which the lanes go IN LANE ORDER (ds_mskor_rtn_b32 serves the
# lanes of one address in ascending order), each leaving (valid, the hash's other bits, its lane) and taking what the lane
# before it on that entry left.  These two functions are that procedure; tests/test_encode_model.py checks them against the
# definitions they implement: "the nearest earlier lane of the round on my table slot" and "of the inserted lanes on one
# slot, the last one writes".
SCRATCH_BITS = 10


def scratch_predecessors(hashes):
    """dep[i] = the nearest earlier lane with hashes[i], or 64 -- through the scratch exchange, with the lanes whose entry
    was last used by another slot's lane settled the way the kernel settles them (a look over all lanes of that hash)"""
    scratch = {}
    dep = [64] * len(hashes)
    ambiguous = []
    for lane, h in enumerate(hashes):  # ascending service order
        key, tag = h & ((1 << SCRATCH_BITS) - 1), h >> SCRATCH_BITS
        prev = scratch.get(key)
        scratch[key] = (tag, lane)
        if prev is not None:
            if prev[0] == tag:
                dep[lane] = prev[1]
            else:
                ambiguous.append(lane)
    done = set()
    for j in ambiguous:  # (one pass per hash value among them: ballot(h == hj))
        if hashes[j] in done:
            continue
        done.add(hashes[j])
        group = [i for i, h in enumerate(hashes) if h == hashes[j]]
        for a, b in zip(group, group[1:]):
            dep[b] = a
    return dep


def scratch_writers(hashes, inserted):
    """which lanes write their position to the table: of the INSERTED lanes of a slot the last one -- the second exchange,
    among the inserted lanes only, and a look at who is on the entry last"""
    scratch = {}
    for lane, h in enumerate(hashes):
        if inserted[lane]:
            scratch[h & ((1 << SCRATCH_BITS) - 1)] = (h >> SCRATCH_BITS, lane)
    writer = [False] * len(hashes)
    ambiguous = []
    for lane, h in enumerate(hashes):
        if not inserted[lane]:
            continue
        tag, last = scratch[h & ((1 << SCRATCH_BITS) - 1)]
        if last == lane:
            writer[lane] = True
        elif tag != (h >> SCRATCH_BITS) or last < lane:
            ambiguous.append(lane)
    done = set()
    for j in ambiguous:
        if hashes[j] in done:
            continue
        done.add(hashes[j])
        group = [i for i, h in enumerate(hashes) if h == hashes[j] and inserted[i]]
        for i in group:
            writer[i] = i == group[-1]
    return writer

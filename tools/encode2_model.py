"""Lane-level model of encode2_kernel.h: WIDE fresh rounds (R = 64 * W consecutive positions worked on by W
waves of one workgroup) in front of the 64-lane rounds of encode_kernel.h (tools/encode_model.py), which take
every case the wide round declines (the block's first round, rounds near ipLimit, a first segment that finds
nothing or runs into a wrong candidate, long scans).  tests/test_encode_model.py runs it against the oracle.

The wide round, lane i <-> position base + i (base = ip - 1 right after a copy that ended at ip):
  * every lane hashes its position, reads the table (all reads before all writes), writes its position;
    candidate = the nearest earlier lane of the round on the same slot, else the table value;
  * every lane measures its match against that candidate (up to CAP bytes; longer: "long", extended only
    when the chain reaches it);
  * every lane works out "if a copy ENDED at me: which lane matches first among my copy-loop probe and the
    scan's probes (offsets 0..31, 32, 34, .. 62), and where does that copy end" (nxt);
  * one wave walks the chain from lane 1; E = the ends it went on from, MS = the matches it took;
  * S = the lanes the sequential loop really touches; a probe in S whose nearest earlier same-slot lane is
    not in S saw a wrong candidate: everything from the copy end in front of the first such probe is undone;
  * the table is left as S alone leaves it; the round's elements are emitted position-parallel.
"""
import encode_model as em

PAT = em.PAT
CAP = 80  # bytes a lane measures on its own (16 in registers + 4 more pieces of 16)


def _ctz(x):
    return (x & -x).bit_length() - 1


def wide_round(data, n, table, hsh, base, R, ip_limit, out, stats):
    """One wide fresh round after a copy that ended at base + 1.  Returns None if the round declines (the table
    is then untouched), else (ended, tail_from, next state)."""
    MR = (1 << R) - 1
    p = [base + i for i in range(R)]
    d = [em._ld32(data, q) for q in p]
    h = [hsh(x) for x in d]
    old = [table[x] for x in h]
    dep = [R] * R
    last = {}
    for i in range(R):
        if h[i] in last:
            dep[i] = last[h[i]]
        last[h[i]] = i
    succ = [0] * R
    for i in range(R):
        if dep[i] < R:
            succ[dep[i]] = i
    cand = [p[dep[i]] if dep[i] < R else old[i] for i in range(R)]
    m4 = 0
    eq = [0] * R
    for i in range(R):
        if em._ld32(data, cand[i]) == d[i]:
            m4 |= 1 << i
            k = 0
            while k < CAP and p[i] + k < n and data[cand[i] + k] == data[p[i] + k]:
                k += 1
            eq[i] = k
    # per lane: where the sequential loop goes if a copy ended here
    mine = (PAT << 1) | 1
    mv = [None] * R
    for i in range(R):
        c = (mine << i) & m4 & MR
        if c:
            mv[i] = _ctz(c)
    # the chain, walked by one wave
    E = MS = 0
    lens = list(eq)
    e = 1
    while True:
        if e > R - 2:
            break
        m = mv[e]
        if m is None:
            break
        E |= 1 << e
        MS |= 1 << m
        if eq[m] == CAP:
            lens[m] = em._match_len(data, cand[m], p[m], n)
        e = m + lens[m]
    if MS == 0:
        stats["wide_declined"] = stats.get("wide_declined", 0) + 1
        return None
    # what the sequential loop inserted (S), what the copies cover
    ce, run = [1] * R, 1
    covered = [False] * R
    for i in range(R):
        ce[i] = run
        endv = i + lens[i] if (MS >> i) & 1 else 0
        covered[i] = i < max(run, endv)
        run = max(run, endv)
    mlast = MS.bit_length() - 1
    in_s = [False] * R
    for i in range(R):
        if i > mlast:
            continue
        if i >= ce[i]:
            o = i - ce[i] - 1
            in_s[i] = i == ce[i] or o < 32 or not (o & 1)
        else:
            in_s[i] = i + 1 == ce[i]
    bad = [i for i in range(R) if in_s[i] and dep[i] < R and not in_s[dep[i]] and not (E >> (i + 1)) & 1]
    if bad:
        fb = bad[0]
        eb = E & ((2 << fb) - 1)
        e = eb.bit_length() - 1
        MS &= (1 << e) - 1
        covered = [covered[i] and i < e for i in range(R)]
        in_s = [in_s[i] and i + 1 < e for i in range(R)]
        stats["wide_cut"] = stats.get("wide_cut", 0) + 1
    if MS == 0:  # the cut took everything: decline (the model has not touched the table yet)
        stats["wide_declined"] = stats.get("wide_declined", 0) + 1
        return None
    # the table as S alone leaves it: all members of a slot read the same `old`
    for i in range(R):
        if dep[i] < R:
            continue  # a group's first member decides for the group
        top, j = (i if in_s[i] else None), i
        while succ[j]:
            j = succ[j]
            if in_s[j]:
                top = j
        table[h[i]] = p[top] if top is not None else old[i]
    # the elements, position-parallel
    LIT = sum(1 << i for i in range(1, min(e, R)) if not covered[i])
    for i in range(R):
        if (LIT >> i) & 1:
            if not (LIT >> (i - 1)) & 1:
                rl = _ctz(~(LIT >> i))
                assert rl <= 63
                if rl <= 60:
                    out.append((rl - 1) << 2)
                else:
                    out += bytes([60 << 2, rl - 1])
            out.append(data[p[i]])
        elif (MS >> i) & 1:
            em._emit_copy(out, p[i] - cand[i], lens[i])
    stats["wide_rounds"] = stats.get("wide_rounds", 0) + 1
    stats["wide_positions"] = stats.get("wide_positions", 0) + e - 1
    stats["wide_hops"] = stats.get("wide_hops", 0) + bin(MS).count("1")
    if base + e > ip_limit:
        return True, base + e, None
    return False, None, (True, base + e + 1, 0, base + e)


def encode_block(data, R=256, stats=None):
    """encode_model.encode_block with the wide round in front of its dense round."""
    stats = {} if stats is None else stats
    data = bytes(data)
    n = len(data)

    def hook(table, hsh, base, ip_limit, out):
        if base + R + 32 <= ip_limit:
            return wide_round(data, n, table, hsh, base, R, ip_limit, out, stats)
        return None

    return em.encode_block(data, stats, wide=hook)

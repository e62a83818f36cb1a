"""CPU model of the raw-buffer splitter (nim-snappy_amd/csrc/split_kernels.h), segment by segment, the way the
kernels do it: candidate entries per 256-byte segment, walks that hand their exits on, follow-through of long
literals, successor pointers, marking by pointer doubling.  tests/test_split_model.py checks that the marked
entries are exactly the sequential parse's (decoder.nim:39-109) on the streams the GPU tests use.  Not the product:
the product is the HIP code; this is its executable description."""

SEG, CAND, CLEAN, FOLLOW_MIN = 256, 6, 8, 1024
PENDING, END, BAD = -1, -2, -3


def element(s, p):
    """(output bytes, stream bytes, tag) of the element at p, or None if invalid (decode_element_bf, index_kernel.h)"""
    n = len(s)
    tag = s[p]
    rem = n - p - 1
    t, hi6 = tag & 3, tag >> 2
    if t == 0:
        if hi6 >= 60:
            ll = hi6 - 59
            if rem < 61:
                return None
            L = (int.from_bytes(s[p + 1:p + 5].ljust(4, b"\0"), "little") & ((1 << (8 * ll)) - 1)) + 1
            if L >= 1 << 32:
                return None
            h = 1 + ll
        else:
            L, h = hi6 + 1, 1
        if rem - (h - 1) < L:
            return None
        return L, h + L, tag
    size = (2, 3, 5)[t - 1]
    if rem < size - 1:
        return None
    return (4 + (hi6 & 7) if t == 1 else 1 + hi6), size, tag


def native(tag):
    return not ((tag & 3) == 3 or ((tag & 3) == 0 and (tag >> 2) >= 62))


def split(s, max_looks=6):
    """-> (entries {segment: entry position} of the marked chain, rounds) or None (the kernels would fall back)"""
    n = len(s)
    nseg = (n + SEG - 1) // SEG
    ent = [[] for _ in range(nseg)]       # (pos) -- every listed candidate is trusted
    ext = [[] for _ in range(nseg)]
    ent[0].append(0)
    ext[0].append(PENDING)
    overflow = False

    def add(pos):
        nonlocal overflow
        t = pos // SEG
        if pos in ent[t]:
            return t, ent[t].index(pos)
        if len(ent[t]) >= CAND:
            overflow = True
            return None
        ent[t].append(pos)
        ext[t].append(PENDING)
        return t, len(ent[t]) - 1

    def walk(t, pos, clean):
        hi = min((t + 1) * SEG, n)
        last = 0
        while pos < hi:
            e = element(s, pos)
            if e is None:
                return BAD, clean, last
            clean = clean + 1 if native(e[2]) else 0
            pos += e[1]
            last = e[1]
        return pos, clean, last

    def hand_on(pos, clean, last):
        if pos >= n or clean < CLEAN:
            return
        node = add(pos)
        if last < FOLLOW_MIN:
            return
        while node is not None:  # follow-through
            t, c = node
            if ext[t][c] != PENDING:
                break
            e = element(s, pos)
            if e is None or (e[2] & 3) != 0 or (e[2] >> 2) >= 62 or e[1] < FOLLOW_MIN:
                break
            pos += e[1]
            ext[t][c] = END if pos == n else pos
            if pos >= n:
                break
            node = add(pos)

    rounds = 0
    for look in range(max_looks):
        for r in range(4):
            todo = [(t, c) for t in range(nseg) for c in range(len(ent[t])) if ext[t][c] == PENDING]
            results = []
            if look == 0 and r == 0:
                for t in range(1, nseg):  # the guesses (no slot, nobody's successor)
                    results.append((None, walk(t, t * SEG, 0)))
            for t, c in todo:
                results.append(((t, c), walk(t, ent[t][c], CLEAN)))
            for node, (pos, clean, last) in results:
                if node is not None:
                    ext[node[0]][node[1]] = BAD if pos == BAD else (END if pos == n else pos)
                if pos != BAD:
                    hand_on(pos, clean, last)
            rounds += 1
        # successor pointers and the marking
        ids = {(t, c): i for i, (t, c) in enumerate((t, c) for t in range(nseg) for c in range(len(ent[t])))}
        keys = list(ids)
        jump = []
        for t, c in keys:
            x = ext[t][c]
            if x in (PENDING, END, BAD):
                jump.append(x)
            else:
                tt = x // SEG
                jump.append(ids[(tt, ent[tt].index(x))] if x in ent[tt] else PENDING)
        reach = [False] * len(keys)
        reach[0] = True
        steps = 1
        while (1 << (2 * steps)) < nseg + 1:
            steps += 1
        for _ in range(steps):  # four-fold jumps: a marked node marks 1, 2 and 3 hops of the current pointers
            nj = list(jump)
            for i, j in enumerate(jump):
                mark = j >= 0 and reach[i]
                for _hop in range(3):
                    if j < 0:
                        break
                    if mark:
                        reach[j] = True
                    j = jump[j]
                nj[i] = j
            jump = nj
        if jump[0] == END:
            return {t: ent[t][c] for (t, c), i in ids.items() if reach[i]}, rounds
        if jump[0] == BAD or (overflow and look >= 1):
            return None
    return None


def sequential_entries(s):
    """{segment: position of the first element that starts in it} along the sequential parse, or None if invalid"""
    out, p, n = {}, 0, len(s)
    while p < n:
        e = element(s, p)
        if e is None:
            return None
        out.setdefault(p // SEG, p)
        p += e[1]
    return out

"""CPU model of the raw-buffer splitter (nim-snappy_amd/csrc/split_kernels.h), segment by segment, the way the
kernels do it: candidate entries per 256-byte segment, walks that hand their exits on (a wave's local rounds, walks that
stop where they fall into step with an earlier one), follow-through of long literals, successor pointers, marking by
pointer doubling.  tests/test_split_model.py checks that the marked
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


WG, LOCAL_MAX, LAUNCHES_FIRST, LAUNCHES_LATER = 64, 66, 4, 4


def split(s, max_looks=6, with_out=False, late=False):
    """-> (entries {segment: entry position} of the marked chain, launches) or None (the kernels would fall back).
    with_out: a third item, {segment: output bytes of the elements the marked walk of it covers}.
    late: a candidate handed to a segment of ANOTHER wave is seen by that wave in the next launch only (the waves of a
    launch run side by side; the model runs them one after the other, which is the other extreme).

    One launch of the walk kernel = every wave (64 segments, their 16 KiB staged in LDS) that has something new
    (first launch: every wave; later: its dirty flag) runs LOCAL rounds until none of its segments has a candidate
    that is not walked: a candidate handed to a segment of the same wave is walked in the next local round, without
    a launch.  A segment's first walk of THIS launch leaves checkpoints (the first element start in each of its eight
    32-byte blocks); a later walk that enters a block at its checkpoint has fallen into step with the first one: it
    stops, and takes that walk's exit (which that walk has handed on already) and output bytes from there (found by
    walking the first walk again up to the meeting point: a few elements).  If the first walk did not hand its exit on
    (it ended without its credit of native elements), the walk goes on to the end on its own."""
    n = len(s)
    nseg = (n + SEG - 1) // SEG
    nwg = (nseg + WG - 1) // WG
    ent = [[] for _ in range(nseg)]       # (pos) -- every listed candidate is trusted
    ext = [[] for _ in range(nseg)]
    ob = [[] for _ in range(nseg)]
    ent[0].append(0)
    ext[0].append(PENDING)
    ob[0].append(0)
    born = [[] for _ in range(nseg)]      # (launch, wave that added it)
    born[0].append((-1, -1))
    now = [0, -1]                         # the launch, the wave that is running
    dirty = [False] * nwg
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
        ob[t].append(0)
        born[t].append((now[0], now[1]))
        dirty[t // WG] = True
        return t, len(ent[t]) - 1

    def hand_on(pos, clean, last):
        """-> handed on (the exit is a candidate now, or was one)"""
        if pos >= n or clean < CLEAN:
            return False
        node = add(pos)
        if node is None:
            return False
        if last < FOLLOW_MIN:
            return True
        while node is not None:  # follow-through
            t, c = node
            if ext[t][c] != PENDING:
                break
            e = element(s, pos)
            if e is None or (e[2] & 3) != 0 or (e[2] >> 2) >= 62 or e[1] < FOLLOW_MIN:
                break
            pos += e[1]
            ob[t][c] = e[0]
            ext[t][c] = END if pos == n else pos
            if pos >= n:
                break
            node = add(pos)
        return True

    def launch(first_launch):
        for w in range(nwg):
            if not first_launch and not dirty[w]:
                continue
            dirty[w] = False
            now[1] = w
            segs = range(w * WG, min((w + 1) * WG, nseg))
            # a segment's FIRST walk of this launch (the first launch: the guess) leaves checkpoints -- per 32-byte block,
            # the first element start in it; a later walk that enters a block at its checkpoint is in step with it
            first = {}                     # t -> [checkpoints {block: pos}, entry, exit code, output bytes, handed on]
            walked = set()
            for it in range(LOCAL_MAX):
                todo = [(t, c) for t in segs for c in range(len(ent[t])) if ext[t][c] == PENDING and (t, c) not in walked
                        and not (late and born[t][c][0] == now[0] and born[t][c][1] != w)]
                if first_launch and it == 0:
                    todo = [(t, None) for t in segs if t != 0] + todo  # the guesses (no slot, nobody's successor)
                if not todo:
                    if late and any(ext[t][c] == PENDING for t in segs for c in range(len(ent[t]))):
                        dirty[w] = True   # (the flag its sender sets behind the wave's look at its lists)
                    break
                for t, c in todo:
                    hi = min((t + 1) * SEG, n)
                    entry = t * SEG if c is None else ent[t][c]
                    pos, out, clean, last = entry, 0, (0 if c is None else CLEAN), 0
                    code, handed = None, False
                    mode = 2 if t in first else 1
                    cps = first[t][0] if t in first else {}
                    blk_prev = None
                    while pos < hi:
                        blk = (pos - t * SEG) // 32
                        if blk != blk_prev:
                            blk_prev = blk
                            if mode == 1:
                                cps[blk] = pos
                            elif mode == 2 and cps.get(blk) == pos:
                                _, e0, x0, o0, h0 = first[t]
                                if x0 in (END, BAD) or h0:
                                    p2, o2 = e0, 0   # what the first walk put out up to here
                                    while p2 < pos:
                                        e = element(s, p2)
                                        o2 += e[0]
                                        p2 += e[1]
                                    assert p2 == pos
                                    code, out, handed = x0, out + o0 - o2, True
                                    break
                                mode = 0   # that walk kept its exit to itself: go on alone
                        e = element(s, pos)
                        if e is None:
                            code = BAD
                            break
                        clean = clean + 1 if native(e[2]) else 0
                        out += e[0]
                        pos += e[1]
                        last = e[1]
                    if code is None:
                        code = END if pos == n else pos
                        handed = hand_on(pos, clean, last)
                    if c is not None:
                        ext[t][c] = code
                        ob[t][c] = out
                        walked.add((t, c))
                    if t not in first:
                        first[t] = [cps, entry, code, out, handed]

    launches = 0
    for look in range(max_looks):
        for r in range(LAUNCHES_FIRST if look == 0 else LAUNCHES_LATER):
            now[0] = launches
            launch(look == 0 and r == 0)
            launches += 1
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
            entries = {t: ent[t][c] for (t, c), i in ids.items() if reach[i]}
            if with_out:
                return entries, launches, {t: ob[t][c] for (t, c), i in ids.items() if reach[i]}
            return entries, launches
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


def sequential_out(s):
    """{segment: output bytes of the walk that enters it on the sequential parse} -- what the marked walk of a segment must
    report: the elements from its entry to the first element start behind the segment, or, for a walk that followed a
    long literal through, that one element"""
    out, p, n = {}, 0, len(s)
    cur = None
    while p < n:
        e = element(s, p)
        if e is None:
            return None
        t = p // SEG
        if t != cur:
            out[t] = 0
            cur = t
        out[t] += e[0]
        p += e[1]
    return out

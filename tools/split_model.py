"""CPU model of the raw-buffer splitter (nim-snappy_amd/csrc/split_kernels.h), segment by segment, the way the
kernels do it: candidate entries per 256-byte segment, walks that hand their exits on (the bulk launch's first walks and
local rounds, walks that stop where they fall into step with the segment's first one, the queue of the tail rounds),
follow-through of long literals, successor pointers, marking by pointer doubling.  tests/test_split_model.py checks that the marked
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


WG, BUDGET, HOPS, ROUNDS_BLIND, ROUNDS_MAX, JUMP = 64, 64, 16, 4, 12, 8


def split(s, with_out=False, rounds=ROUNDS_BLIND):
    """-> (entries {segment: entry position} of the marked chain, tail rounds that had work) or None (the chain is not
    complete behind `rounds` tail rounds, or never: the kernels would try again with ROUNDS_MAX, then fall back).
    with_out: a third item, {segment: output bytes of the elements the marked walk of it covers}.

    The bulk launch: every segment's FIRST walk, from its first byte (a guess; segment 0: the root), which leaves
    checkpoints (the first element start in each of its eight 32-byte blocks) and a summary; an exit into the next
    segment of the same wave (64 segments) is walked there at once, with a budget of elements: a walk that enters a
    block at the first walk's checkpoint has fallen into step with it and takes over its exit and its output bytes
    from there; what is not done within the budget, and every other exit, goes on the queue.  A tail round: every
    queued node by a lane of its own, the same way without a budget, and on along what it hands on, HOPS nodes at most;
    the rest onto the next round's queue."""
    n = len(s)
    nseg = (n + SEG - 1) // SEG
    nwg = (nseg + WG - 1) // WG
    # per segment: the listed candidates (every one is trusted); slot 0 is the segment's own (None: empty) -- the root's
    # in segment 0, else what the first walk of the segment before hands over inside a wave of the bulk launch
    ent = [[None] for _ in range(nseg)]
    ext = [[PENDING] for _ in range(nseg)]
    ob = [[0] for _ in range(nseg)]
    ent[0][0] = 0
    first = {}                            # t -> (checkpoints {block: pos}, entry, exit code, output bytes, handed on)
    state = {"overflow": False}

    def add(pos):
        """-> (node, is_new)"""
        t = pos // SEG
        if pos in ent[t]:
            return (t, ent[t].index(pos)), False
        if len(ent[t]) >= CAND:  # (slot 0 is not given out here)
            state["overflow"] = True
            return None, False
        ent[t].append(pos)
        ext[t].append(PENDING)
        ob[t].append(0)
        return (t, len(ent[t]) - 1), True

    def hand_on(pos, clean, last, my_wg):
        """-> (handed on, the node that still has to be walked and is not one of wave my_wg's own, or None)"""
        if pos >= n or clean < CLEAN:
            return False, None
        node, is_new = add(pos)
        if node is None:
            return False, None
        open_ = node if is_new else None
        if last >= FOLLOW_MIN:
            while node is not None:  # follow-through
                t, c = node
                if ext[t][c] != PENDING:
                    open_ = None
                    break
                e = element(s, pos)
                if e is None or (e[2] & 3) != 0 or (e[2] >> 2) >= 62 or e[1] < FOLLOW_MIN:
                    break
                pos += e[1]
                ob[t][c] = e[0]
                ext[t][c] = END if pos == n else pos
                open_ = None
                if pos >= n:
                    break
                node, is_new = add(pos)
                if is_new:
                    open_ = node
        if open_ is not None and open_[0] // WG == my_wg:
            open_ = None
        return True, open_

    def walk(t, pos, clean, mode, budget, cps=None):
        """-> (pos, out, clean, last, bad, hit); mode 1 fills cps"""
        hi = min((t + 1) * SEG, n)
        out, last, blk_prev = 0, 0, None
        if mode == 2:
            cps = first[t][0]
        while pos < hi and budget != 0:
            budget -= 1
            blk = (pos - t * SEG) // 32
            if blk != blk_prev:
                blk_prev = blk
                if mode == 1:
                    cps[blk] = pos
                elif mode == 2 and cps.get(blk) == pos:
                    return pos, out, clean, last, False, True
            e = element(s, pos)
            if e is None:
                return pos, out, clean, last, True, False
            clean = clean + 1 if native(e[2]) else 0
            out += e[0]
            pos += e[1]
            last = e[1]
        return pos, out, clean, last, False, False

    def candidate(t, c, budget, my_wg):
        """-> the node for the next round's queue (its exit's, or its own), or None"""
        hi = min((t + 1) * SEG, n)
        pos, out, clean, last, bad, hit = walk(t, ent[t][c], CLEAN, 2, budget)
        code = None
        if hit:
            _, e0, x0, o0, h0 = first[t]
            if x0 in (END, BAD) or h0:
                p2, o2 = e0, 0   # what the first walk put out up to here
                while p2 < pos:
                    e = element(s, p2)
                    o2 += e[0]
                    p2 += e[1]
                assert p2 == pos
                code, out = x0, out + o0 - o2
            else:  # that walk kept its exit to itself: go on alone
                if budget != -1:
                    return (t, c)
                p3, o3, clean, last, bad, _ = walk_from(t, pos, clean)
                pos, out = p3, out + o3
        elif not bad and pos < hi:
            return (t, c)   # (the budget)
        own = code is None and not bad
        if bad:
            code = BAD
        if own:
            code = END if pos == n else pos
        ext[t][c] = code
        ob[t][c] = out
        return hand_on(pos, clean, last, my_wg)[1] if own else None

    def walk_from(t, pos, clean):
        hi = min((t + 1) * SEG, n)
        out, last = 0, 0
        while pos < hi:
            e = element(s, pos)
            if e is None:
                return pos, out, clean, last, True, False
            clean = clean + 1 if native(e[2]) else 0
            out += e[0]
            pos += e[1]
            last = e[1]
        return pos, out, clean, last, False, False

    # the bulk launch: the first walks; a first walk's exit into the NEXT segment of the same wave goes there by shuffle
    # and is slot 0 of that segment's list (written by the segment's own lane), walked at once, with the budget; every
    # other exit is listed by compare-and-swap (slots 1 .. 5) and goes on the queue
    queue = []
    for w in range(nwg):
        segs = range(w * WG, min((w + 1) * WG, nseg))
        direct = {}
        for t in segs:
            cps = {}
            pos, out, clean, last, bad, _ = walk(t, t * SEG, CLEAN if t == 0 else 0, 1, -1, cps)
            code = BAD if bad else (END if pos == n else pos)
            if t == 0:
                ext[0][0], ob[0][0] = code, out
            handed = False
            if not bad and pos < n and clean >= CLEAN:
                if t + 1 in segs and pos < min((t + 1) * SEG, n) + SEG and last < FOLLOW_MIN:
                    direct[t + 1] = pos
                    handed = True
                else:
                    handed, open_ = hand_on(pos, clean, last, None)
                    if open_ is not None:
                        queue.append(open_)
            first[t] = (cps, t * SEG, code, out, handed)
        for t, pos in direct.items():
            assert ent[t][0] is None
            ent[t][0] = pos
            nxt = candidate(t, 0, BUDGET, None)
            if nxt is not None:
                queue.append(nxt)
    # the tail rounds
    worked = 0
    for r in range(rounds):
        if not queue:
            break
        worked += 1
        nxt_queue = []
        for node in queue:
            for hop in range(HOPS):
                t, c = node
                if ext[t][c] != PENDING:
                    node = None
                    break
                node = candidate(t, c, -1, None)
                if node is None:
                    break
            if node is not None:
                nxt_queue.append(node)
        queue = nxt_queue
    # successor pointers and the marking
    ids = {(t, c): i for i, (t, c) in enumerate((t, c) for t in range(nseg) for c in range(len(ent[t])) if ent[t][c] is not None)}
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
    steps, far = 1, JUMP
    while far < nseg + 1:
        steps, far = steps + 1, far * JUMP
    for _ in range(steps):  # JUMP-fold jumps: a marked node marks 1 .. JUMP - 1 hops of the current pointers
        nj = list(jump)
        for i, j in enumerate(jump):
            mark = j >= 0 and reach[i]
            for _hop in range(JUMP - 1):
                if j < 0:
                    break
                if mark:
                    reach[j] = True
                j = jump[j]
            nj[i] = j
        jump = nj
    if jump[0] != END:
        return None
    entries = {t: ent[t][c] for (t, c), i in ids.items() if reach[i]}
    if with_out:
        return entries, worked, {t: ob[t][c] for (t, c), i in ids.items() if reach[i]}
    return entries, worked


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

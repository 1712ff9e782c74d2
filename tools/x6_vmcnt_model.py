#!/usr/bin/env python3
"""Counted s_waitcnt vmcnt(N) values of csrc/layer_x6.hip, derived by replaying the loop's vector-memory instruction stream.
Every wave issues, per 32-row tile, a fixed sequence of vector-memory instructions (LDS-DMA rows D_i of tile t+2, stores S_q of tile
t-1's results, dgrad: mask loads m_q of tile t); vmcnt counts them in order.  The staged row i is read back at step 2 i (gap 0) of the
NEXT tile: the wait before that read must let exactly the younger instructions stay in flight.  This script lists the (step, gap)
schedule, replays three tiles and prints the counts the kernel hard-codes (X6_VM_FWD / X6_VM_DGRAD / the mask wait).
The schedule must match the kernel's: (step, gap) of every instruction below is what the source does."""


def schedule(dgrad):
    ops = []                      # (step, gap, order-in-gap, name)
    for i in range(8):
        if i <= 6:
            ops.append((2 * i + 2, 3, 0, f"D{i}"))
        else:
            ops.append((15, 5, 0, f"D{i}"))
    for q in range(4):
        if q <= 2:
            ops.append((10 + 2 * q, 4, 0, f"S{q}"))
        else:
            ops.append((15, 5, 1, f"S{q}"))
    if dgrad:
        for q in range(4):
            ops.append((2 * q, 2, 0, f"m{q}"))
    return sorted(ops)


def counts(dgrad):
    tile = schedule(dgrad)
    stream = [(t, s, g, o, n) for t in range(3) for (s, g, o, n) in tile]
    out = []
    for i in range(8):
        # raw_read(i) happens in tile 2 at (step 2 i, gap 0) BEFORE anything else of that gap; it needs D_i of tile 1
        idx = next(k for k, x in enumerate(stream) if x[0] == 1 and x[4] == f"D{i}")
        younger = [x for x in stream[idx + 1:] if (x[0], x[1], x[2]) < (2, 2 * i, 0)]
        out.append(len(younger))
    mask_wait = None
    if dgrad:   # all four masks of tile 1 are needed after step 15 of tile 1: everything of tile 1 issued after m3 may stay in flight
        idx = next(k for k, x in enumerate(stream) if x[0] == 1 and x[4] == "m3")
        mask_wait = len([x for x in stream[idx + 1:] if x[0] == 1])
    return out, mask_wait


if __name__ == "__main__":
    for dgrad in (False, True):
        print("dgrad" if dgrad else "forward", "order per tile:", " ".join(n for (_, _, _, n) in schedule(dgrad)))
        c, mw = counts(dgrad)
        print("   vmcnt before raw_read(i), i = 0..7:", c, "  mask wait:", mw)


def lgkm_counts():
    """LDS instruction stream of one tile (F = fragment read, R = staging read, W = plane write) and the lgkmcnt every wait needs."""
    stream, marks = [], {}
    stream += [("F", 0)] * 3                                  # rd(0), behind the barrier
    for j in range(16):
        i, even = j >> 1, (j & 1) == 0
        marks[("start", j)] = len(stream)                     # wait for the fragments of step j
        if j + 1 < 16:
            stream += [("F", j + 1)] * 3
        if even:
            stream.append(("R", i))
        else:
            marks[("raw", j)] = len(stream)                   # wait for the staging read of piece i
        if even and i > 0:
            stream += [("W", i - 1)] * 2                      # split_e of the previous piece
        if not even:
            stream.append(("W", i))                           # split_c
        if j == 15:
            stream += [("W", 7)] * 2
    out = {}
    for (kind, j), pos in marks.items():
        want = ("F", j) if kind == "start" else ("R", j >> 1)
        last = max(k for k in range(pos) if stream[k] == want)
        out[(kind, j)] = pos - 1 - last                       # instructions issued after the wanted one, before the wait
    return out


if __name__ == "__main__":
    lg = lgkm_counts()
    print("lgkmcnt at step start, j = 0..15:", [lg[("start", j)] for j in range(16)])
    print("lgkmcnt before the first split (odd steps), j = 1,3,..,15:", [lg[("raw", j)] for j in range(1, 16, 2)])

#!/usr/bin/env python3
"""Counted s_waitcnt vmcnt(N) values of csrc/layer_x6.hip (the eight-wave form), derived by replaying the loop's vector-memory stream.

Every wave issues, per 32-row tile, a FIXED sequence of vector-memory instructions; vmcnt counts them in issue order and retires them in
order.  A tile has 8 k-steps (j) of 6 MFMA gaps each; piece i (the wave's staged row i of the NEXT tile, i = 0..3) is read back at
(step 2 i, gap 0) and was DMA'd ("D_i") during the previous tile.  The wait in front of that read-back must let exactly the younger
instructions stay in flight.  Variants (template arguments of k_layer_x6) and what they add to the stream:

  forward           D0 D1 S0 D2 S1 D3        S_q = store of the previous tile's 16 columns (steps 4, 6, gap 4)
  dgrad             m0 m1 + forward          m_q = ReLU-mask loads of THIS tile's rows (steps 0, 2, gap 2), needed at the end of the tile
  forward, OUTV=1   forward + P              P = store of the output layer's partial sums of the previous tile (step 2, gap 4)
  forward, OUTV=2   D0 P D1 D2 D3            hidden activation not written
  dgrad, K3W        p D0 D1 D2 D3            p = load of this tile's row positions (step 0, gap 2), needed at the end of the tile; no stores
  dgrad, BM         b + forward              b = load of this tile's sign byte (step 0, gap 2) instead of m0 m1
  forward, BM       forward + B              B = store of the previous tile's sign byte (step 2, gap 4): the table of OUTV=1

(GEN, the generated-input forward, has its own one-line argument in the source: vmcnt(2) everywhere.)
Prints the tables the kernel hard-codes as X8_VM[variant][i] and the end-of-tile waits.  The (step, gap) of every instruction below is
what the source does -- change both together."""


def schedule(variant):
    ops = []                                        # (step, gap, name)
    dgrad = variant in ("dgrad", "k3w", "dgrad_bm")
    stores = variant in ("forward", "dgrad", "outv1", "fwd_bm", "dgrad_bm")
    ops += [(2, 3, "D0"), (4, 3, "D1"), (6, 3, "D2"), (7, 5, "D3")]
    if stores:
        ops += [(4, 4, "S0"), (6, 4, "S1")]
    if variant == "dgrad":
        ops += [(0, 2, "m0"), (2, 2, "m1")]
    if variant == "k3w":
        ops += [(0, 2, "p")]
    if variant in ("outv1", "outv2"):
        ops += [(2, 4, "P")]
    if variant == "fwd_bm":                         # forward that also writes the sign bytes: one byte store where P goes
        ops += [(2, 4, "B")]
    if variant == "dgrad_bm":                       # dgrad that reads sign bytes: one load instead of the two mask loads
        ops += [(0, 2, "b")]
    assert dgrad or variant in ("forward", "outv1", "outv2", "fwd_bm")
    return sorted(ops)


def counts(variant):
    tile = schedule(variant)
    stream = [(t, s, g, n) for t in range(3) for (s, g, n) in tile]
    waits = []
    for i in range(4):
        # the read-back of piece i happens in tile 2 at (step 2 i, gap 0), before anything else of that gap; it needs D_i of tile 1
        idx = next(k for k, x in enumerate(stream) if x[0] == 1 and x[3] == f"D{i}")
        younger = [x for x in stream[idx + 1:] if (x[0], x[1], x[2]) < (2, 2 * i, 0)]
        waits.append(len(younger))
    tail = None
    last_needed = {"dgrad": "m1", "k3w": "p", "dgrad_bm": "b"}.get(variant)
    if last_needed:                                 # needed after step 7 of its own tile: everything of the tile issued after it may stay in flight
        idx = next(k for k, x in enumerate(stream) if x[0] == 1 and x[3] == last_needed)
        tail = len([x for x in stream[idx + 1:] if x[0] == 1])
    return waits, tail


if __name__ == "__main__":
    for row, variant in enumerate(("forward", "dgrad", "outv1", "outv2", "k3w", "dgrad_bm", "fwd_bm")):
        w, tail = counts(variant)
        print(f"X8_VM[{row}] {variant:8s} stream per tile: {' '.join(n for (_, _, n) in schedule(variant)):28s} waits {w}"
              + (f"   end-of-tile wait vmcnt({tail})" if tail is not None else ""))

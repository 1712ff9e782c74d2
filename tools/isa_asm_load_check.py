#!/usr/bin/env python3
"""Guard for hand-issued loads in a hipcc listing (hipcc -S --cuda-device-only): between an inline-asm load (ds_read_* / global_load_* inside
;;#ASMSTART ... ;;#ASMEND) and the NEXT s_waitcnt that names the counter it is tracked by, no instruction may read or overwrite the
destination registers -- the compiler does not know the data is still in flight, so a register copy (a spill to an AGPR, a phi copy at a
branch merge) would read stale bits.  Prints every violation; exit code 1 if any.

Second check, "counted waits" (round 6): an LDS-DMA (global_load_lds_* / buffer_load_* ... lds) has no destination register, so a wrong hand-counted
``s_waitcnt vmcnt(N)`` in front of the ds_read of the slot it fills is invisible to the check above.  The kernels mark both ends in the listing
(csrc/gemm_common.h: ``; @dma <tag>`` in front of the DMA, ``; @use <tag> <allow>`` behind the counted wait, in front of the first read of
that slot).  Every loop of the listing that holds such a mark is replayed TWICE (the DMA of one trip is consumed in the next) over its
vector-memory stream -- every global_ / buffer_ / flat_ instruction counts, loads, stores and atomics, retired in issue order -- and at a
``@use`` at most <allow> DMAs of that tag may still be in flight; it also reports the largest count that would have been safe there.
usage: python tools/isa_asm_load_check.py file.s"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def operands(line):
    parts = line.strip().split(None, 1)
    if len(parts) < 2:
        return []
    return [t.strip() for t in re.split(r",\s*", parts[1].split(";")[0]) if t.strip()]


bad = 0
lines = open(sys.argv[1]).read().splitlines()
in_asm = False
queue = {"lgkm": [], "vm": []}       # every outstanding LDS / vector-memory instruction in issue order: (dest registers or None, line, text)
for n, l in enumerate(lines, 1):
    t = l.strip()
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if re.match(r"^_Z\w+:", t):
        queue = {"lgkm": [], "vm": []}
        continue
    op = t.split()[0]
    if op == "s_waitcnt":
        for key, name in (("lgkm", "lgkmcnt"), ("vm", "vmcnt")):
            m = re.search(name + r"\((\d+)\)", t)
            if m:                                   # in-order counters: all but the N youngest are complete
                k = int(m.group(1))
                queue[key] = queue[key][len(queue[key]) - k:] if k else []
        continue
    if op in ("s_barrier",):
        continue
    ops = operands(t)
    touched = set()
    for o in ops:
        touched |= regs(o.split()[0])
    is_lds = op.startswith("ds_")
    is_vm = op.startswith(("global_", "buffer_", "flat_"))
    for key in ("lgkm", "vm"):
        for dst, ln, txt in queue[key]:
            if dst and touched & dst:
                print(f"line {n}: '{t}' touches v{sorted(touched & dst)} still in flight from line {ln}: '{txt}'")
                bad += 1
    if is_lds:
        dst = regs(ops[0].split()[0]) if (in_asm and op.startswith("ds_read") and ops) else None
        queue["lgkm"].append((dst, n, t))
    elif is_vm:
        dst = regs(ops[0].split()[0]) if (in_asm and op.startswith("global_load") and "lds" not in op and ops) else None
        queue["vm"].append((dst, n, t))


def counted_wait_check(lines):
    """Returns (violations, notes).  See the module docstring."""
    viol, notes = [], []
    # kernels: name -> list of (line number, text)
    kernels, cur = [], None
    for n, l in enumerate(lines, 1):
        t = l.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            cur = (m.group(1), [])
            kernels.append(cur)
        elif t.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None and t and not t.startswith(";;#"):
            cur[1].append((n, t))
    for name, body in kernels:
        if not any("@use" in t for _, t in body):
            continue
        labels = {t.split(":")[0]: k for k, (_, t) in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", t)}
        loops = []
        for k, (_, t) in enumerate(body):
            m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", t) or re.match(r"^s_branch\s+(\.LBB\d+_\d+)", t)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                loops.append((labels[m.group(1)], k))
        for a, b in loops:
            seg = body[a:b + 1]
            if not any("@use" in t for _, t in seg):
                continue
            queue, pending = [], None                    # in-flight vector-memory instructions, oldest first: (tag or None, line)
            for trip in range(2):
                for n, t in seg:
                    if t.startswith(";"):
                        m = re.match(r"^;\s*@dma\s+(\S+)", t)
                        if m:
                            pending = (m.group(1), n)
                            continue
                        m = re.match(r"^;\s*@use\s+(\S+)\s+(\d+)", t)
                        if m and trip == 1:
                            tag, allow = m.group(1), int(m.group(2))
                            flying = [q for q in queue if q[0] == tag]
                            # the largest vmcnt that still retires all but `allow` of this tag's DMAs: everything younger than the (allow+1)-th youngest
                            idx = [i for i, q in enumerate(queue) if q[0] == tag]
                            if len(flying) > allow:
                                viol.append(f"{name}: line {n}: '@use {tag} {allow}' with {len(flying)} DMA(s) of that tag still in flight (issued at line(s) "
                                            f"{[q[1] for q in flying]}): the s_waitcnt vmcnt in front of it lets too many instructions stay outstanding")
                            notes.append((name, n, tag, allow, len(queue)))
                        continue
                    op = t.split()[0]
                    if op == "s_waitcnt":
                        m = re.search(r"vmcnt\((\d+)\)", t)
                        if m:
                            k = int(m.group(1))
                            queue = queue[len(queue) - k:] if k else []
                        continue
                    if op.startswith(("global_", "buffer_", "flat_")) and not op.startswith("buffer_wbl2") and not op.startswith("buffer_inv"):
                        tag = None
                        if pending is not None:
                            if "lds" not in t:
                                viol.append(f"{name}: line {pending[1]}: '@dma {pending[0]}' is not followed by an LDS-DMA (next vector-memory instruction: '{t}')")
                            tag = pending[0]
                            pending = None
                        queue.append((tag, n))
    return viol, notes


v2, notes = counted_wait_check(lines)
for v in v2:
    print(v)
if "-v" in sys.argv:
    for name, n, tag, allow, depth in notes:
        print(f"  {name[:40]} line {n}: use {tag} (allow {allow}) with {depth} vector-memory instruction(s) outstanding behind the wait")
print("violations:", bad + len(v2), f"(in-flight registers {bad}, counted waits {len(v2)}; {len(notes)} marked use(s) replayed)")
sys.exit(1 if (bad or v2) else 0)

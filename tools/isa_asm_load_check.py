#!/usr/bin/env python3
"""Guard for hand-issued loads in a hipcc listing (hipcc -S --cuda-device-only): between an inline-asm load (ds_read_* / global_load_* inside
;;#ASMSTART ... ;;#ASMEND) and the NEXT s_waitcnt that names the counter it is tracked by, no instruction may read or overwrite the
destination registers -- the compiler does not know the data is still in flight, so a register copy (a spill to an AGPR, a phi copy at a
branch merge) would read stale bits.  Prints every violation; exit code 1 if any.
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
print("violations:", bad)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Instruction mix of every basic block that holds matrix instructions, per kernel of a hipcc device listing (.isa/<name>.s).
usage: python tools/isa_loop_mix.py file.s [kernel-name-substring]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    print(name)
    lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith(";")]
    blocks, cur, label = [], [], "entry"
    for l in lines:
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((label, cur))
            cur, label = [], l
        else:
            cur.append(l)
    blocks.append((label, cur))
    for n, b in blocks:
        c = collections.Counter(x.split()[0] for x in b)
        mf = sum(v for k, v in c.items() if k.startswith("v_mfma"))
        if mf:
            valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
            print(f"  {n:14s} {len(b):5d} instr: mfma {mf:3d} valu {valu:4d} ds {sum(v for k, v in c.items() if k.startswith('ds_')):3d} "
                  f"vmem {sum(v for k, v in c.items() if k.startswith(('global', 'buffer'))):3d} salu {sum(v for k, v in c.items() if k.startswith('s_') and k not in ('s_nop', 's_waitcnt')):3d} "
                  f"s_nop {c.get('s_nop', 0):3d} s_waitcnt {c.get('s_waitcnt', 0):3d} barrier {c.get('s_barrier', 0)}")

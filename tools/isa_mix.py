"""Instruction mix of the MFMA-carrying basic blocks of a gfx950 assembly listing (hipcc -S --cuda-device-only): a quick check that a
hand-scheduled loop kept its shape (no v_accvgpr copies, no scratch, the expected number of fillers per MFMA).
usage: python tools/isa_mix.py file.s [substring of the kernel's mangled name]"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):.*\n", s, flags=re.M):
    name = m.group(1)
    if want not in name:
        continue
    i = m.end()
    j = s.index(".Lfunc_end", i)
    blocks, cur, label = [], [], "entry"
    for l in s[i:j].splitlines():
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((label, cur)); label, cur = l.strip(), []
        else:
            cur.append(l)
    blocks.append((label, cur))
    for label, b in blocks:
        n = sum("v_mfma" in l for l in b)
        if n == 0:
            continue
        cnt = {}
        for l in b:
            t = l.strip().split(" ")[0] if l.strip() else ""
            if t and not t.startswith(";") and not t.startswith("."):
                cnt[t] = cnt.get(t, 0) + 1
        print(name, label, "mfma", n, "instructions", sum(cnt.values()))
        print("   ", dict(sorted(cnt.items(), key=lambda x: -x[1])[:40]))

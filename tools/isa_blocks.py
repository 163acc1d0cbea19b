#!/usr/bin/env python3
"""Basic blocks of one kernel in a hipcc device .s file: label, size, opcode mix; marks backward branches (loops).
usage: isa_blocks.py file.s kernel-name-regex [min-size]"""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
pat = re.compile(sys.argv[2])
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
inside, blocks, cur = False, [], None
order = {}
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        inside = bool(pat.search(m.group(1)))
        if inside:
            cur = ["<entry>", collections.Counter(), []]
            blocks.append(cur)
        continue
    if not inside:
        continue
    if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
        inside = False
        continue
    m = re.match(r"^(\.LBB\w+):", ln)
    if m:
        cur = [m.group(1), collections.Counter(), []]
        order[m.group(1)] = len(blocks)
        blocks.append(cur)
        continue
    m = re.match(r"^\t([a-z_0-9]+)\s*(.*)", ln)
    if m and not m.group(1).startswith("."):
        cur[1][m.group(1)] += 1
        if m.group(1).startswith("s_cbranch") or m.group(1) == "s_branch":
            cur[2].append(m.group(2).strip())
for i, (lab, c, br) in enumerate(blocks):
    tot = sum(c.values())
    if tot < minsz:
        continue
    back = [t for t in br if t in order and order[t] <= i]
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    ds = sum(v for k, v in c.items() if k.startswith("ds_"))
    print(f"{lab:14s} n={tot:5d} valu={valu:5d} ds={ds:4d} {'LOOP->' + ','.join(back) if back else ''}")
    print("      " + ", ".join(f"{k}:{v}" for k, v in c.most_common(14)))

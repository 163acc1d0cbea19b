#!/usr/bin/env python3
"""Instruction histogram of the kernels in a hipcc --save-temps device .s file (static counts).
usage: isa_hist.py file.s [regex-on-demangled-or-mangled-name]"""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
cur, funcs = None, {}
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1)
        funcs[cur] = collections.Counter()
        continue
    if cur is None:
        continue
    if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
        cur = None
        continue
    m = re.match(r"^\t([a-z_0-9]+)", ln)
    if m and not m.group(1).startswith("."):
        funcs[cur][m.group(1)] += 1
for name, c in funcs.items():
    if pat and not pat.search(name):
        continue
    tot = sum(c.values())
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    pk = sum(v for k, v in c.items() if k.startswith("v_pk_"))
    ds = sum(v for k, v in c.items() if k.startswith("ds_"))
    print(f"== {name}\n   total {tot}  valu {valu} (pk {pk})  ds {ds}  salu {sum(v for k, v in c.items() if k.startswith('s_'))}")
    print("   " + ", ".join(f"{k}:{v}" for k, v in c.most_common(28)))

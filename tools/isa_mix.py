#!/usr/bin/env python3
"""Instruction mix per kernel from a hipcc -S device assembly file: tools/isa_mix.py file.s [name-substring ...]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:]
names = re.findall(r"^(_ZN3wbc\S+):", txt, flags=re.M)
for name in names:
    if want and not any(w in name for w in want):
        continue
    body = txt.split("\n" + name + ":", 1)[1].split(".Lfunc_end", 1)[0]
    ops = collections.Counter()
    for l in body.split("\n"):
        l = l.strip()
        if not l or l.startswith((".", ";", "//")) or l.endswith(":"):
            continue
        ops[l.split()[0]] += 1
    grp = collections.Counter()
    for k, v in ops.items():
        if k.startswith("v_") and "f64" in k:
            grp["valu_f64"] += v
        elif k.startswith("v_"):
            grp["valu_other"] += v
        elif k.startswith("s_waitcnt"):
            grp["s_waitcnt"] += v
        elif k.startswith("s_"):
            grp["salu"] += v
        elif k.startswith("ds_"):
            grp["lds"] += v
        elif k.startswith(("global_", "buffer_", "flat_", "scratch_")):
            grp["vmem_" + ("store" if "store" in k else "load")] += v
        else:
            grp[k] += v
    print(name[:60], "total", sum(ops.values()), dict(grp))
    print("   ", ops.most_common(16))

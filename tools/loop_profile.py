#!/usr/bin/env python3
"""Static instruction profile of a kernel's hottest loop by source line (needs an asm dump with line tables:
hipcc -O3 --offload-arch=gfx950 -gline-tables-only -S --cuda-device-only wbc_api.hip -o /tmp/asm/wbc_g.s).
usage: loop_profile.py ASM KERNEL_SUBSTR FILE_SUBSTR [top]"""
import collections
import re
import sys

asm, kern, fsub = sys.argv[1], sys.argv[2], sys.argv[3]
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
lines = open(asm).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l) or re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"', l)
    if m:
        files[int(m.group(1))] = m.group(2)
# kernel body
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and kern in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
labels = {}
ins = []   # (idx_in_body, text, file, line)
cur = (0, 0)
for i, l in enumerate(body):
    s = l.strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'^(\.LBB\S+):', s)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if not s or s.startswith(".") or s.startswith(";") or s.startswith("//"):
        continue
    ins.append((s, cur))
# loops = backward branches; pick the one with the largest span
best = None
for k, (s, _) in enumerate(ins):
    m = re.match(r's_cbranch\S*\s+(\.LBB\S+)|s_branch\s+(\.LBB\S+)', s)
    if m:
        tgt = labels.get(m.group(1) or m.group(2))
        if tgt is not None and tgt < k and (best is None or k - tgt > best[1] - best[0]):
            best = (tgt, k)
print("kernel instrs", len(ins), "largest loop", best, "=", best[1] - best[0], "instrs")
cnt = collections.Counter()
kind = collections.defaultdict(collections.Counter)
for s, (f, ln) in ins[best[0]:best[1] + 1]:
    key = (files.get(f, "?"), ln)
    cnt[key] += 1
    op = s.split()[0]
    k = ("dpp" if "dpp" in s else "f64" if "_f64" in op else "cndmask" if "cndmask" in op else "salu" if op.startswith("s_") else
         "lds" if op.startswith("ds_") else "valu_other" if op.startswith("v_") else "other")
    kind[key][k] += 1
tot = collections.Counter()
for key in kind:
    tot.update(kind[key])
print("mix:", dict(tot))
src = {}
for (fn, ln), c in cnt.most_common(top):
    if fsub in fn and fn not in src:
        try:
            src[fn] = open("/root/repo/wbc_quadruped_dob_amd/csrc/" + fn).read().split("\n")
        except OSError:
            src[fn] = []
    text = src.get(fn, [])[ln - 1].strip()[:100] if fn in src and 0 < ln <= len(src[fn]) else ""
    print("%4d  %-22s %-5d %s | %s" % (c, fn[-22:], ln, dict(kind[(fn, ln)]), text))

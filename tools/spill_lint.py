#!/usr/bin/env python3
"""Scan the gfx950 ISA of the product library for VGPR spill stores that execute with EXEC possibly empty.

Why: hipcc 7.2 once placed `scratch_store ... Folded Spill` instructions into the exit block of a divergent loop BEFORE the
`s_or_b64 exec, exec, <saved>` that re-enables the lanes (observer_kernel<double>, caught by the parity tests as garbage
rhat).  A spill store in that position writes nothing for the lanes that are masked off and the later reload returns junk.
VGPR spills themselves are fine; this flags only spill stores (and reloads) that sit between the label of a JOIN block and
its exec restore with nothing but bookkeeping (scalar ops, lane writes) around them -- a block that does real work before the
restore is the body of a masked region, whose spills serve the lanes that run it.
`v_writelane` SGPR spills ignore EXEC and are not flagged.

usage: tools/spill_lint.py [file.s]     (without a file: compiles the kernel units csrc/k_*.hip for both scalar types to
                                         device assembly and lints the concatenation, /tmp/asm/wbc_lint.s)
exit status 1 when something is flagged.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


KUNITS = [("k_sweep", ()), ("k_rnea", ()), ("k_qp", ()), ("k_qp_general", ()), ("k_misc", ()), ("k_fused", ()), ("k_rollout", ()),
          ("k_rollout", ("-DWBC_ROLLOUT_TRACK=1",)), ("k_tile", ())]


def compile_asm(out="/tmp/asm/wbc_lint.s", extra=()):
    """Device assembly of every kernel unit x scalar type (the flags of csrc/Makefile), compiled side by side and
    concatenated into `out`."""
    os.makedirs(os.path.dirname(out), exist_ok=True)
    csrc = os.path.join(ROOT, "wbc_quadruped_dob_amd", "csrc")
    jobs = []
    for unit, defs in KUNITS:
        for scalar in ("double", "float"):
            part = "%s.%s%s.%s.s" % (out, unit, "_track" if defs else "", scalar)
            cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-DWBC_SCALAR=" + scalar, *defs, *extra, "-S",
                   "--cuda-device-only", "-w", "-o", part, unit + ".hip"]
            jobs.append((part, subprocess.Popen(cmd, cwd=csrc)))
    with open(out, "w") as f:
        for part, proc in jobs:
            if proc.wait() != 0:
                raise RuntimeError("hipcc failed for " + part)
            f.write(open(part).read())
            os.remove(part)
    return out


def resources(path):
    """{kernel symbol: dict(vgpr, agpr, scratch, lds, scratch_insts)} from the .amdhsa_kernel blocks of `path`; scratch_insts = scratch / stack
    instructions in the kernel's body (a kernel can carry a private segment that nothing touches: a frame object the backend created for SGPR
    spills and then served from VGPR lanes)."""
    txt = open(path).read()
    out = {}
    insts = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*\.end_amdhsa_kernel", txt, flags=re.S | re.M):
        insts[m.group(1)] = len(re.findall(r"^\s*(scratch_(load|store)|buffer_(load|store)\S*\s[^\n]*\boffen\b)", m.group(2), flags=re.M))
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, flags=re.S):
        b = m.group(2)
        g = lambda k: int(re.search(k + r"\s+(\d+)", b).group(1))
        nv, acc = g("amdhsa_next_free_vgpr"), g("amdhsa_accum_offset")
        out[m.group(1)] = dict(vgpr=min(nv, acc), agpr=max(0, nv - acc), scratch=g("amdhsa_private_segment_fixed_size"), lds=g("amdhsa_group_segment_fixed_size"),
                               scratch_insts=insts.get(m.group(1), -1))
    return out


def lint(path):
    """Returns [(kernel, label, line_no, text)] for every masked spill store."""
    bad = []
    kernel, label, work = None, None, False
    pending = []          # spill stores seen in the current block before any exec restore
    restore = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
    spill = re.compile(r"^\s*(scratch|buffer)_(store|load)\S*\s.*Folded (Spill|Reload)")
    lab = re.compile(r"^(\.LBB\d+_\d+):")
    fn = re.compile(r"^(_Z\w+):")
    # bookkeeping that may legitimately sit in front of the exec restore of a join block
    book = re.compile(r"^\s*($|;|\.|s_|v_writelane|v_readlane|v_readfirstlane)")
    with open(path) as f:
        for no, line in enumerate(f, 1):
            m = fn.match(line)
            if m:
                kernel, label, pending, work = m.group(1), None, [], False
                continue
            m = lab.match(line)
            if m:
                label, pending, work = m.group(1), [], False
                continue
            if label is None:
                continue
            if spill.match(line):
                pending.append((no, line.strip()))
            elif restore.match(line):
                if not work:
                    bad += [(kernel, label, n, t) for n, t in pending]
                pending = []
            elif re.match(r"^\s*s_(cbranch|branch|endpgm)", line):
                pending = []      # block ends without restoring exec: the stores ran under the block's own mask
            elif not book.match(line):
                # real work before the restore: this block is the BODY of a masked region (its spills / reloads serve the
                # lanes that run it, the restore at its end closes the region) -- not a join block
                work = True
    return bad


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else compile_asm()
    bad = lint(path)
    for k, l, n, t in bad:
        print(f"{path}:{n}: {k[:60]} {l}: spill store / reload before the exec restore: {t}")
    print(f"spill_lint: {len(bad)} masked spill store(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

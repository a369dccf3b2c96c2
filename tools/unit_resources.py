#!/usr/bin/env python3
"""Registers / scratch / LDS of the kernels of ONE kernel unit (csrc/k_<unit>.hip), without building the library.
usage: tools/unit_resources.py <unit> [double|float] [pattern] [-- extra hipcc flags]"""
import importlib.util
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    unit = args[0]
    scalar = args[1] if len(args) > 1 else "double"
    pat = args[2] if len(args) > 2 else ""
    out = "/tmp/asm/%s_%s.s" % (unit, scalar)
    os.makedirs("/tmp/asm", exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-DWBC_SCALAR=" + scalar, *extra, "-S",
                           "--cuda-device-only", "-w", "-o", out, unit + ".hip"], cwd=os.path.join(ROOT, "wbc_quadruped_dob_amd", "csrc"))
    spec = importlib.util.spec_from_file_location("spill_lint", os.path.join(ROOT, "tools", "spill_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for k, v in sorted(mod.resources(out).items()):
        if re.search(pat, k):
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
            print("%-70s vgpr %3d agpr %3d scratch %4d (%d insts) lds %6d" % (name[-70:], v["vgpr"], v["agpr"], v["scratch"], v["scratch_insts"], v["lds"]))


if __name__ == "__main__":
    main()

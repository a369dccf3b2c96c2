#!/bin/bash
# tools/regs.sh [extra hipcc flags]: registers / scratch / LDS of every kernel of the library (device assembly of all
# kernel units x scalar types, compiled side by side into /tmp/asm/wbc.s)
cd /root/repo || exit 1
python3 - "$@" <<'PY'
import importlib.util, re, sys
spec = importlib.util.spec_from_file_location("spill_lint", "tools/spill_lint.py")
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
txt = open(mod.compile_asm("/tmp/asm/wbc.s", tuple(sys.argv[1:]))).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, flags=re.S):
    name = m.group(1); b = m.group(2)
    g = lambda k: re.search(k + r'\s+(\d+)', b).group(1)
    short = re.sub(r'_ZN3wbc\d+', '', name)[:44]
    print(short.ljust(46), 'vgpr', g('amdhsa_next_free_vgpr'), 'accum_off', g('amdhsa_accum_offset'), 'sgpr', g('amdhsa_next_free_sgpr'),
          'scratch', g('amdhsa_private_segment_fixed_size'), 'lds', g('amdhsa_group_segment_fixed_size'))
PY

#!/bin/bash
# tools/regs_unit.sh <unit> <double|float> [extra flags]: registers / scratch / LDS of the kernels of ONE kernel unit (csrc/<unit>.hip)
u=$1; t=$2; shift 2
cd /root/repo/wbc_quadruped_dob_amd/csrc || exit 1
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DWBC_SCALAR=$t $([ "$t" = double ] && echo -DWBC_SCALAR_IS_DOUBLE=1) "$@" -S --cuda-device-only -o /tmp/asm/$u.$t.s $u.hip 2>/dev/null || exit 1
python3 - /tmp/asm/$u.$t.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, flags=re.S):
    name = m.group(1); b = m.group(2)
    g = lambda k: re.search(k + r'\s+(\d+)', b).group(1)
    short = re.sub(r'_ZN3wbc\d+', '', name)[:48]
    print(short.ljust(50), 'vgpr', g('amdhsa_next_free_vgpr'), 'sgpr', g('amdhsa_next_free_sgpr'), 'scratch', g('amdhsa_private_segment_fixed_size'), 'lds', g('amdhsa_group_segment_fixed_size'))
PY

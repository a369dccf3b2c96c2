#!/bin/bash
# tools/regs.sh [extra hipcc flags]: rebuild the library and print registers/scratch/LDS of every kernel
cd /root/repo/wbc_quadruped_dob_amd/csrc || exit 1
make 2>&1 | grep -E "error|warning: var" | head
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 "$@" -S --cuda-device-only -o /tmp/asm/wbc.s wbc_api.hip 2>&1 | grep error
python3 - <<'PY'
import re
txt=open('/tmp/asm/wbc.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel',txt,flags=re.S):
    name=m.group(1); b=m.group(2)
    g=lambda k: re.search(k+r'\s+(\d+)',b).group(1)
    short=re.sub(r'_ZN3wbc\d+','',name)[:28]
    print(short.ljust(30), 'vgpr',g('amdhsa_next_free_vgpr'),'accum_off',g('amdhsa_accum_offset'),'sgpr',g('amdhsa_next_free_sgpr'),'scratch',g('amdhsa_private_segment_fixed_size'),'lds',g('amdhsa_group_segment_fixed_size'))
PY

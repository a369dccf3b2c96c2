#!/bin/bash
# A/B of library builds on one device: tools/ab_lib.sh libA.so libB.so ...   (files under wbc_quadruped_dob_amd/lib)
set -u
mkdir -p gpurun_out; : > gpurun_out/ablib.log
for L in "$@"; do
  export WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib/$L
  echo "=== $L" >> gpurun_out/ablib.log
  python -m pytest tests -x -q -m gpu 2>&1 | tail -2 >> gpurun_out/ablib.log
done
for rep in 1 2; do for L in "$@"; do
  export WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib/$L
  for args in "--batch 4096" "--config 3 --batch 4096" "--batch 262144 --steps 50 --warmup 5"; do
    python bench.py --steps 300 --warmup 30 --no-cpu --no-latency --large-batch 0 $args 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernels']; f=lambda x:'-' if x is None else '%.1f'%x
print('$L rep $rep [$args] ms/step %.4f  steps/s %.4e  fused %s dyn %s qp %s'%(r['ms_per_step'],r['value'],f(k.get('fused_tick_us')),f(k['dyn_sweep_us']),f(k['qp_us'])))" >> gpurun_out/ablib.log
  done
done; done
cat gpurun_out/ablib.log

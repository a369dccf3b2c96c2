#!/bin/bash
# A/B of split vs fused sweep (env WBC_SWEEP) on one device
set -u
export WBC_FUSED_MAX=${WBC_FUSED_MAX:-0}   # kernel-level A/B of the two-kernel tick: keep small batches off the fused launch
mkdir -p gpurun_out
: > gpurun_out/absw.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 >> gpurun_out/absw.log
for rep in 1 2; do
for K in split fused; do
  for B in 4096 262144; do
    echo "== sweep=$K batch $B rep $rep" >> gpurun_out/absw.log
    WBC_SWEEP=$K python bench.py --steps 200 --warmup 20 --no-cpu --no-latency --large-batch 0 --batch $B 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernels']; print('ms/step %.4f  steps/s %.3e  dyn %.1f us (%.0f GB/s, frac %.3f)  rnea %s us  qp %.1f us'%(r['ms_per_step'],r['value'],k['dyn_sweep_us'],r['roofline']['achieved'],r['roofline']['frac'],('%.1f'%k['rnea_step_us']) if k['rnea_step_us'] else '-',k['qp_us']))" >> gpurun_out/absw.log
  done
done
done
cat gpurun_out/absw.log

#!/bin/bash
# A/B of the two QP kernels (env WBC_QP_KERNEL) on one device
set -u
export WBC_FUSED_MAX=${WBC_FUSED_MAX:-0}   # kernel-level A/B of the two-kernel tick: keep small batches off the fused launch
mkdir -p gpurun_out
: > gpurun_out/abqp.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 >> gpurun_out/abqp.log
for rep in 1 2; do
for K in group16 wave; do
  for B in 4096 262144; do
    echo "== qp=$K batch $B rep $rep" >> gpurun_out/abqp.log
    WBC_QP_KERNEL=$K python bench.py --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --batch $B 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('ms/step %.4f  steps/s %.3e  dyn %.1f us (%.0f GB/s, frac %.3f)  qp %.1f us  iters mean %.2f max %d'%(r['ms_per_step'],r['value'],r['kernels']['dyn_sweep_us'],r['roofline']['achieved'],r['roofline']['frac'],r['kernels']['qp_us'],r['qp']['iters_mean'],r['qp']['iters_max']))" >> gpurun_out/abqp.log
  done
done
done
cat gpurun_out/abqp.log

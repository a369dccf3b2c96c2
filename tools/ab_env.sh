#!/bin/bash
# A/B of an environment switch on one device: tools/ab_env.sh VAR val1 val2 [batches...]
set -u
export WBC_FUSED_MAX=${WBC_FUSED_MAX:-0}   # kernel-level A/B of the two-kernel tick: keep small batches off the fused launch
VAR=$1; V1=$2; V2=$3; shift 3
BATCHES=${@:-"4096 262144"}
mkdir -p gpurun_out; : > gpurun_out/abenv.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 >> gpurun_out/abenv.log
for rep in 1 2; do for V in $V1 $V2; do for B in $BATCHES; do
  echo "== $VAR=$V batch $B rep $rep" >> gpurun_out/abenv.log
  env $VAR=$V python bench.py --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --batch $B 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernels']; print('ms/step %.4f  steps/s %.3e  dyn %.1f us  qp %.1f us'%(r['ms_per_step'],r['value'],k['dyn_sweep_us'],k['qp_us']))" >> gpurun_out/abenv.log
done; done; done
cat gpurun_out/abenv.log

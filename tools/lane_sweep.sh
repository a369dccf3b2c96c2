#!/bin/bash
# A/B of the QP stage: dense active-set kernel alone (qp_lane = -1) vs per-lane semismooth Newton + hand-over (qp_lane = 1)
export TMPDIR=/tmp
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; print("%-38s %8.1f M steps/s  %8.4f ms  sweep %7.1f  qp_lane %7.1f  qp(dense) %7.1f  obs/rnea %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], k.get("dyn_sweep_us") or 0, k.get("qp_lane_us") or 0, k.get("qp_us") or 0, k.get("rnea_step_us")))'
for cfg in ${CFGS:-2 3 4}; do
for n in ${NS:-12288 32768 65536 262144}; do
  for l in -1 1; do
    WBC_FUSED_MAX=0 WBC_QP_LANE=$l $B --steps 40 --warmup 5 --batch $n --config $cfg | python -c "$pick" "cfg$cfg n=$n qp_lane=$l"
  done
done; done

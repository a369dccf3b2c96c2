#!/bin/bash
# A/B of the QP dealing: one-wavefront workgroups vs tiles of 64 ... 512 states, several batch sizes
export TMPDIR=/tmp
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; print("%-34s %8.1f M steps/s  sweep %7.1f  qp %7.1f us" % (sys.argv[1], d["value"]/1e6, k.get("dyn_sweep_us") or 0, k.get("qp_us") or 0))'
for n in ${NS:-16384 32768 65536 131072 262144}; do
  for t in ${TILES:--1 64 128 256 512}; do
    WBC_FUSED_MAX=0 WBC_QP_TILE=$t $B --steps 40 --warmup 5 --batch $n ${EXTRA:-} | python -c "$pick" "n=$n tile=$t ${EXTRA:-}"
  done
done

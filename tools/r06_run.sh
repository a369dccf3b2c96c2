#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_run.log; : > $L
timeout 600 python tools/tt64_check.py >> $L 2>&1
B="python bench.py --no-cpu --no-latency --large-batch 0 --no-closed-loop"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.2f" % x; print("%-12s %-44s %9.1f M steps/s  %8.4f ms/step  fused %s sweep %s  qp %s lane %s" % (sys.argv[1], sys.argv[2], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us"))))'
for n in 4096 6144 8192 11264 12288 16384 24576 28672 32768 57344 65536 131072 262144; do
  st=$(( 2000000 / n + 10 ))
  A="--steps $st --warmup 5 --batch $n"
  $B $A 2>/dev/null | python -c "$pick" "default" "$A" >> $L
  WBC_TILE_TICK=1 WBC_FUSED_MAX=0 $B $A 2>/dev/null | python -c "$pick" "tile_tick" "$A" >> $L
done
cat $L

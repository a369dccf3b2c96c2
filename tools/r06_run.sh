#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_run.log; : > $L
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> $L
B="python bench.py --no-cpu --no-latency --large-batch 0 --no-closed-loop"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.2f" % x; print("%-12s %-44s %9.1f M steps/s  %8.4f ms/step  fused %s sweep %s  qp %s lane %s" % (sys.argv[1], sys.argv[2], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us"))))'
for n in 36864 40960 45056 49152 57344 65536 73728 81920 90112 98304 114688 131072 163840 196608 229376 262144; do
  st=$(( 3000000 / n + 10 ))
  A="--steps $st --warmup 5 --batch $n --config 4"
  WBC_TILE_TICK=-1 $B $A 2>/dev/null | python -c "$pick" "two-launch" "$A" >> $L
  WBC_TILE_TICK=1 $B $A 2>/dev/null | python -c "$pick" "tile_tick" "$A" >> $L
done
cat $L

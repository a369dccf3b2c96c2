#!/bin/bash
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-34s %8.1f M/s %8.4f ms/step sweep %s qp %s lane %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us"))))'
for n in 12288 16384 24576 32768 49152 65536; do
  for t in -1 32 64 128; do
    WBC_QP_LANE=-1 WBC_QP_TILE=$t $B --steps 100 --warmup 10 --batch $n | python -c "$pick" "f64 n$n tile=$t"
  done
  WBC_QP_LANE=1 $B --steps 100 --warmup 10 --batch $n | python -c "$pick" "f64 n$n lane"
done
for n in 16384 32768 49152 65536; do
  for t in -1 32 64; do
    WBC_QP_LANE=-1 WBC_QP_TILE=$t $B --steps 100 --warmup 10 --batch $n --config 4 | python -c "$pick" "f32 cfg4 n$n tile=$t"
  done
done

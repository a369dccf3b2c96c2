#!/bin/bash
# fp64 tiles sized so that the launch is ONE round of three workgroups per CU (768 workgroups): 32 / 40 / 44 / 48 / 64 states per tile
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-34s %8.1f M/s %8.4f ms/step sweep %s qp %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("dyn_sweep_us")), f(k.get("qp_us"))))'
for n in ${SIZES:-24576 28672 30720 32768 36864 40960}; do
  st=$(( 3000000 / n + 20 ))
  for t in 32 40 44 48 64; do
    WBC_QP_LANE=-1 WBC_QP_TILE=$t $B --steps $st --warmup 10 --batch $n | python -c "$pick" "cfg2 f64 n$n tile $t"
  done
  WBC_QP_LANE=-1 WBC_QP_TILE=48 $B --steps $st --warmup 10 --batch $n --config 3 | python -c "$pick" "cfg3 f64 n$n tile 48"
  WBC_QP_LANE=-1 WBC_QP_TILE=64 $B --steps $st --warmup 10 --batch $n --config 3 | python -c "$pick" "cfg3 f64 n$n tile 64"
done

#!/bin/bash
# as r03_tiles.sh, the other two combinations: fp32 on the hard (standing) batch, fp64 on the easy (trot, observer on) batch
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-34s %8.1f M/s %8.4f ms/step sweep %s qp %s lane %s rnea %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us"))))'
for n in ${SIZES:-16384 24576 32768}; do
  st=$(( 3000000 / n + 20 ))
  for t in -1 32 64; do
    WBC_FUSED_MAX=0 WBC_QP_TILE=$t $B --steps $st --warmup 10 --batch $n --dtype f32 | python -c "$pick" "cfg2 f32 n$n tile $t"
    WBC_FUSED_MAX=0 WBC_QP_TILE=$t $B --steps $st --warmup 10 --batch $n --config 3 | python -c "$pick" "cfg3 f64 n$n tile $t"
  done
done

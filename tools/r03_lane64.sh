#!/bin/bash
# fp64: per-lane QP pair against tiles at 98 304 ... 131 072 states (standing and trot batch)
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-36s %8.1f M/s %8.4f ms/step sweep %s qp %s lane %s rnea %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us"))))'
for n in 98304 114688 131072 163840; do
  WBC_QP_LANE=-1 $B --steps 40 --warmup 10 --batch $n | python -c "$pick" "cfg2 f64 n$n tiles"
  WBC_QP_LANE=1 $B --steps 40 --warmup 10 --batch $n | python -c "$pick" "cfg2 f64 n$n lane"
  WBC_QP_LANE=-1 $B --steps 40 --warmup 10 --batch $n --config 3 | python -c "$pick" "cfg3 f64 n$n tiles"
  WBC_QP_LANE=1 $B --steps 40 --warmup 10 --batch $n --config 3 | python -c "$pick" "cfg3 f64 n$n lane"
done

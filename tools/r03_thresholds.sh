#!/bin/bash
# re-measured at the end of round 3: observer as its own kernel (WBC_OBS_SPLIT_MIN) for fp64 observer-on ticks, states per workgroup of the persistent rollout
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-40s %8.1f M/s %8.4f ms/step sweep %s qp %s lane %s rnea %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us"))))'
for n in 12288 16384 20480 24576 32768; do
  st=$(( 3000000 / n + 20 ))
  WBC_OBS_SPLIT_MIN=100000000 $B --steps $st --warmup 10 --batch $n --config 3 | python -c "$pick" "cfg3 f64 n$n all-in-one"
  WBC_OBS_SPLIT_MIN=1 $B --steps $st --warmup 10 --batch $n --config 3 | python -c "$pick" "cfg3 f64 n$n split"
done
p5='import sys,json; d=json.loads(sys.stdin.read()); print("%-40s %8.2f M/s %8.4f ms/rollout" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"]))'
for n in 512 1024 2048 4096; do
  for spw in 4 16; do
    WBC_ROLLOUT_SPW=$spw python bench.py --config 5 --steps 60 --warmup 6 --batch $n | python -c "$p5" "cfg5 n$n spw $spw"
  done
done

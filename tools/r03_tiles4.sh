#!/bin/bash
# fp32 tiles (two workgroups per CU: 512 resident): 64 / 72 / 80 / 96 states per tile between 32 768 and 65 536 states; fp64 with 52 / 56 / 60 above 36 864
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-34s %8.1f M/s %8.4f ms/step sweep %s qp %s rnea %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("rnea_step_us"))))'
for n in 32768 36864 40960 45056 49152; do
  st=$(( 3000000 / n + 20 ))
  for t in 64 72 80 96; do
    WBC_QP_TILE=$t $B --steps $st --warmup 10 --batch $n --config 4 | python -c "$pick" "cfg4 f32 n$n tile $t"
  done
done
for n in 40960 45056 49152; do
  st=$(( 3000000 / n + 20 ))
  for t in 52 56 60 64; do
    WBC_QP_LANE=-1 WBC_QP_TILE=$t $B --steps $st --warmup 10 --batch $n | python -c "$pick" "cfg2 f64 n$n tile $t"
  done
  $B --steps $st --warmup 10 --batch $n | python -c "$pick" "cfg2 f64 n$n default"
done

#!/bin/bash
# default dispatch over the mid-range batch sizes (two-kernel ticks, QP tiles): fp64 standing / trot batch, fp32 trot batch
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-30s %8.1f M/s %8.4f ms/step sweep %s qp %s lane %s rnea %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us"))))'
for n in ${SIZES:-14336 20480 24576 28672 30720 32768 36864 40960 45056 49152 57344 65536}; do
  st=$(( 3000000 / n + 20 ))
  $B --steps $st --warmup 10 --batch $n | python -c "$pick" "cfg2 f64 n$n"
  $B --steps $st --warmup 10 --batch $n --config 3 | python -c "$pick" "cfg3 f64 n$n"
  $B --steps $st --warmup 10 --batch $n --config 4 | python -c "$pick" "cfg4 f32 n$n"
done

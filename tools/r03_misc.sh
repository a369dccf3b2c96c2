#!/bin/bash
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-40s %8.1f M/s %8.4f ms/step sweep %s qp %s lane %s rnea %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us"))))'
for n in 34816 36864 38912; do
  $B --steps 100 --warmup 10 --batch $n --config 4 | python -c "$pick" "cfg4 f32 n$n default"
  WBC_OBS_SPLIT_MIN=33000 $B --steps 100 --warmup 10 --batch $n --config 4 | python -c "$pick" "cfg4 f32 n$n split"
  WBC_F32_PACK2=-1 $B --steps 100 --warmup 10 --batch $n --config 4 | python -c "$pick" "cfg4 f32 n$n nopack2"
done

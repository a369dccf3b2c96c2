#!/bin/bash
# (A/B behind the tile rule of wbc_api.cpp for fp32 batches past 65 536 states) fp32 batches between one and two rounds of the dense-body tiles (four workgroups per CU): default tile rule against tiles of 64
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-40s %8.1f M/s %8.4f ms/step sweep %s qp %s lane %s rnea %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us"))))'
for n in 73728 81920 98304 114688 131072 163840; do
  $B --steps 40 --warmup 10 --batch $n --config 4 | python -c "$pick" "cfg4 f32 n$n default"
  WBC_QP_TILE=64 $B --steps 40 --warmup 10 --batch $n --config 4 | python -c "$pick" "cfg4 f32 n$n tile 64"
done

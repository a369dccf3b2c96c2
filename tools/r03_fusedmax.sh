#!/bin/bash
# observer-on fp64 ticks between 4 096 and 8 192 states: two rounds of the fused tick (WBC_FUSED_MAX=8192) against the two-kernel tick (default)
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-36s %8.1f M/s %8.4f ms/step fused %s sweep %s qp %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), f(k.get("dyn_sweep_us")), f(k.get("qp_us"))))'
for n in 5120 6144 8192; do
  $B --steps 300 --warmup 30 --batch $n --config 3 | python -c "$pick" "cfg3 f64 n$n default"
  WBC_FUSED_MAX=8192 $B --steps 300 --warmup 30 --batch $n --config 3 | python -c "$pick" "cfg3 f64 n$n fused"
done

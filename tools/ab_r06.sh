#!/bin/bash
# round 6 A/B on one device: tools/ab_r06.sh "<bench args>" libdirA libdirB ...  (directories under wbc_quadruped_dob_amd/), alternating, 3 reps
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
ARGS=$1; shift
B="python bench.py --no-cpu --no-latency --large-batch 0 --no-closed-loop"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.2f" % x; print("%-12s %-40s %9.1f M steps/s  %8.4f ms/step  fused %s  sweep %s  qp %s  lane %s  iters %.2f" % (sys.argv[1], sys.argv[2], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), (d.get("qp") or {}).get("iters_mean", 0) or 0))'
for rep in 1 2 3; do for L in "$@"; do
  WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so $B $ARGS 2>/dev/null | python -c "$pick" "$L" "$ARGS"
done; done

#!/bin/bash
# quick check: GPU tests + the headline numbers
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED" | tail -8
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-34s %8.1f M/s %8.4f ms/step fused %s sweep %s qp %s lane %s rnea %s it %.2f max %s ok %.4f" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us")), (d.get("qp") or {}).get("iters_mean", 0) or 0, (d.get("qp") or {}).get("iters_max"), (d.get("qp") or {}).get("status_ok_frac", 0)))'
for rep in 1 2; do
  $B --steps 500 --warmup 50 | python -c "$pick" "cfg2 n4096"
  $B --steps 500 --warmup 50 --config 3 | python -c "$pick" "cfg3 n4096"
  $B --steps 200 --warmup 20 --batch 8192 | python -c "$pick" "cfg2 n8192"
  $B --steps 100 --warmup 10 --batch 16384 | python -c "$pick" "cfg2 n16384"
  $B --steps 100 --warmup 10 --batch 32768 | python -c "$pick" "cfg2 n32768"
  $B --steps 100 --warmup 10 --batch 32768 --config 4 | python -c "$pick" "cfg4 f32 n32768"
  $B --steps 50 --warmup 5 --batch 262144 | python -c "$pick" "cfg2 n262144"
  $B --steps 500 --warmup 50 --config 4 | python -c "$pick" "cfg4 f32 n4096"
  python bench.py --config 5 --steps 50 --warmup 5 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("cfg5 h20 n1024: %.1f M steps/s, %.2f us/tick" % (d["value"]/1e6, d["us_per_tick"]))'
done
[ -f wbc_quadruped_dob_amd/lib_fstamp/libwbc_hip.so ] && WBC_LIB=$R/wbc_quadruped_dob_amd/lib_fstamp/libwbc_hip.so python tools/fused_stamp.py 2>&1 | sed -n 1,26p

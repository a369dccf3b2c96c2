#!/bin/bash
# round 3, GPU run 7: fp32 solvers on the structured QP body (fp64 arithmetic) vs the orthogonal-factor fp32 body
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03_run7"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -m pytest tests -q -m gpu > "$O/pytest_gpu_full.log" 2>&1; grep -E "passed|failed|FAILED" "$O/pytest_gpu_full.log" | tail -8
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-34s %8.1f M/s %8.4f ms/step fused %s sweep %s qp %s lane %s rnea %s it %.2f max %s ok %.4f" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us")), (d.get("qp") or {}).get("iters_mean", 0) or 0, (d.get("qp") or {}).get("iters_max"), (d.get("qp") or {}).get("status_ok_frac", 0)))'
for rep in 1 2; do
for L in lib lib_s1; do
  export WBC_LIB=$R/wbc_quadruped_dob_amd/$L/libwbc_hip.so
  for n in 4096 8192 16384 32768 65536 131072 262144; do
    st=$(( 3000000 / n + 20 ))
    $B --steps $st --warmup 10 --batch $n --config 4 | python -c "$pick" "$L cfg4 f32 n$n" >> "$O/ab.log"
  done
  WBC_QP_LANE=-1 $B --steps 50 --warmup 5 --batch 262144 --config 4 | python -c "$pick" "$L cfg4 f32 n262144 nolane" >> "$O/ab.log"
  WBC_QP_LANE=-1 $B --steps 50 --warmup 5 --batch 131072 --config 4 | python -c "$pick" "$L cfg4 f32 n131072 nolane" >> "$O/ab.log"
  $B --steps 100 --warmup 10 --batch 32768 --config 2 --dtype f32 | python -c "$pick" "$L cfg2 f32 n32768" >> "$O/ab.log"
  $B --steps 500 --warmup 50 --config 2 --dtype f32 | python -c "$pick" "$L cfg2 f32 n4096" >> "$O/ab.log"
  python bench.py --config 5 --steps 50 --warmup 5 --dtype f32 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "cfg5 f32 h20 n1024: %.1f M steps/s, %.2f us/tick" % (d["value"]/1e6, d["us_per_tick"]))' $L >> "$O/ab.log"
done
done
unset WBC_LIB
cat "$O/ab.log"
python tools/f32_error_survey.py > "$O/f32_error_survey.log" 2>&1; tail -15 "$O/f32_error_survey.log"

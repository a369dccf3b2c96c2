#!/bin/bash
# round 3, GPU run 4: trimmed structured QP loop, rnea role split over two wavefronts; single-wave issue probe
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03_run4"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
./tools/issue_probe.bin > "$O/issue_probe.log" 2>&1; cat "$O/issue_probe.log"
python -m pytest tests -q -m gpu > "$O/pytest_gpu_full.log" 2>&1; tail -5 "$O/pytest_gpu_full.log"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-34s %8.1f M/s %8.4f ms/step fused %s sweep %s qp %s lane %s rnea %s it %.2f max %s ok %.4f" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us")), (d.get("qp") or {}).get("iters_mean", 0) or 0, (d.get("qp") or {}).get("iters_max"), (d.get("qp") or {}).get("status_ok_frac", 0)))'
for rep in 1 2; do
for L in lib lib_nosplit; do
  export WBC_LIB=$R/wbc_quadruped_dob_amd/$L/libwbc_hip.so
  $B --steps 500 --warmup 50 | python -c "$pick" "$L cfg2 n4096" >> "$O/ab.log"
  $B --steps 500 --warmup 50 --config 3 | python -c "$pick" "$L cfg3 n4096" >> "$O/ab.log"
  $B --steps 500 --warmup 50 --batch 2048 | python -c "$pick" "$L cfg2 n2048" >> "$O/ab.log"
  $B --steps 200 --warmup 20 --batch 6144 | python -c "$pick" "$L cfg2 n6144" >> "$O/ab.log"
  $B --steps 200 --warmup 20 --batch 8192 | python -c "$pick" "$L cfg2 n8192" >> "$O/ab.log"
  $B --steps 200 --warmup 20 --batch 12288 | python -c "$pick" "$L cfg2 n12288" >> "$O/ab.log"
  WBC_FUSED_MAX=16384 $B --steps 200 --warmup 20 --batch 12288 | python -c "$pick" "$L cfg2 n12288 fused" >> "$O/ab.log"
  $B --steps 100 --warmup 10 --batch 32768 | python -c "$pick" "$L cfg2 n32768" >> "$O/ab.log"
  python bench.py --config 5 --steps 50 --warmup 5 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "cfg5 h20 n1024: %.1f M steps/s, %.2f us/tick" % (d["value"]/1e6, d["us_per_tick"]))' $L >> "$O/ab.log"
done
done
unset WBC_LIB
cat "$O/ab.log"
WBC_LIB=$R/wbc_quadruped_dob_amd/lib_fstamp/libwbc_hip.so python tools/fused_stamp.py > "$O/fused_timeline.txt" 2>&1
head -32 "$O/fused_timeline.txt"

#!/bin/bash
# round 3, GPU run 2: full GPU test-suite; fused-tick timelines and QP segment stamps (new vs round-2 forms); A/B of the three libs
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03_run2"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -m pytest tests -q -m gpu > "$O/pytest_gpu_full.log" 2>&1; tail -25 "$O/pytest_gpu_full.log"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-34s %8.1f M/s %8.4f ms/step fused %s sweep %s qp %s lane %s rnea %s it %.2f max %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us")), (d.get("qp") or {}).get("iters_mean", 0) or 0, (d.get("qp") or {}).get("iters_max")))'
for rep in 1 2 3; do
for L in lib lib_norinv lib_nozq; do
  export WBC_LIB=$R/wbc_quadruped_dob_amd/$L/libwbc_hip.so
  $B --steps 500 --warmup 50 | python -c "$pick" "$L cfg2 n4096" >> "$O/ab.log"
  $B --steps 500 --warmup 50 --config 3 | python -c "$pick" "$L cfg3 n4096" >> "$O/ab.log"
  $B --steps 500 --warmup 50 --batch 2048 | python -c "$pick" "$L cfg2 n2048" >> "$O/ab.log"
done
done
unset WBC_LIB
cat "$O/ab.log"
for L in fstamp fstamp_old; do
  echo "=== $L" >> "$O/fused_timeline.txt"
  WBC_LIB=$R/wbc_quadruped_dob_amd/lib_$L/libwbc_hip.so python tools/fused_stamp.py >> "$O/fused_timeline.txt" 2>&1
done
for L in qstamp qstamp_old; do
  echo "=== $L" >> "$O/qp_segments.txt"
  WBC_LIB=$R/wbc_quadruped_dob_amd/lib_$L/libwbc_hip.so python tools/qp_stamp.py >> "$O/qp_segments.txt" 2>&1
done
cat "$O/fused_timeline.txt" "$O/qp_segments.txt"
python bench.py --steps 20 --warmup 5 --large-batch 0 --no-cpu > "$O/bench_latency.json" 2> "$O/bench_latency.err"
python - <<PY
import json
d=json.load(open("$O/bench_latency.json"))
print(json.dumps(d["qp_latency"], indent=1)); print(json.dumps(d["roofline"], indent=1)); print(json.dumps(d.get("device"), indent=1)); print(d["roofline_whole_path_bytes"])
PY

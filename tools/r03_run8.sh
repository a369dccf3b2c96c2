#!/bin/bash
# round 3, GPU run 8: keep_structural, trimmed loop, defaults; full tests; default bench with all legs
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03_run8"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -m pytest tests -q -m gpu > "$O/pytest_gpu_full.log" 2>&1; grep -E "passed|failed|FAILED" "$O/pytest_gpu_full.log" | tail -8
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-34s %8.1f M/s %8.4f ms/step fused %s sweep %s qp %s lane %s rnea %s it %.2f max %s ok %.4f" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us")), (d.get("qp") or {}).get("iters_mean", 0) or 0, (d.get("qp") or {}).get("iters_max"), (d.get("qp") or {}).get("status_ok_frac", 0)))'
for rep in 1 2; do
for KS in 0 1; do
  export WBC_KEEP_STRUCTURAL=$KS
  $B --steps 500 --warmup 50 | python -c "$pick" "keep=$KS cfg2 n4096" >> "$O/ab.log"
  $B --steps 200 --warmup 20 --batch 8192 | python -c "$pick" "keep=$KS cfg2 n8192" >> "$O/ab.log"
  $B --steps 100 --warmup 10 --batch 24576 | python -c "$pick" "keep=$KS cfg2 n24576" >> "$O/ab.log"
  $B --steps 100 --warmup 10 --batch 32768 | python -c "$pick" "keep=$KS cfg2 n32768" >> "$O/ab.log"
  $B --steps 100 --warmup 10 --batch 32768 --config 4 | python -c "$pick" "keep=$KS cfg4 f32 n32768" >> "$O/ab.log"
  $B --steps 50 --warmup 5 --batch 65536 | python -c "$pick" "keep=$KS cfg2 n65536" >> "$O/ab.log"
  $B --steps 50 --warmup 5 --batch 262144 | python -c "$pick" "keep=$KS cfg2 n262144" >> "$O/ab.log"
  $B --steps 50 --warmup 5 --batch 262144 --config 3 | python -c "$pick" "keep=$KS cfg3 n262144" >> "$O/ab.log"
  $B --steps 50 --warmup 5 --batch 262144 --config 4 | python -c "$pick" "keep=$KS cfg4 f32 n262144" >> "$O/ab.log"
  python bench.py --config 5 --steps 50 --warmup 5 --batch 32768 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "cfg5 h20 n32768: %.1f M steps/s, %.2f us/tick" % (d["value"]/1e6, d["us_per_tick"]))' keep=$KS >> "$O/ab.log"
done
done
unset WBC_KEEP_STRUCTURAL
cat "$O/ab.log"
python bench.py --steps 20 --warmup 5 > "$O/bench_default_driver_flags.json" 2> "$O/bench_default.err"
python - <<PY
import json
d=json.load(open("$O/bench_default_driver_flags.json"))
print("value %.1f M/s ms/step %.4f" % (d["value"]/1e6, d["ms_per_step"]))
for k in ("roofline","roofline_large_batch","qp_latency","device","cpu_baseline","roofline_whole_path_bytes"):
    print(k, json.dumps(d.get(k))[:1500])
PY

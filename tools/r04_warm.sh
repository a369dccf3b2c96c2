#!/bin/bash
# round 4: rollouts with / without the warm start of the QP (configs[4]), the GPU test suite, rocprofv3 of the rollout kernel
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04_warm"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -m pytest tests -q -m gpu -x > "$O/pytest_gpu_full.log" 2>&1; tail -5 "$O/pytest_gpu_full.log"
pick='import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; print("%-34s %8.1f M steps/s  %7.2f us/tick  launch %s us  frac %s  iters(last tick) %s" % (sys.argv[1], d["value"]/1e6, d["us_per_tick"], r.get("avg_launch_us"), r.get("frac"), d["qp"].get("iters_mean_last_tick")))'
for w in 1 0; do
  for n in 1024 128 4096 32768; do
    WBC_ROLLOUT_WARM=$w python bench.py --config 5 --steps 50 --warmup 5 --batch $n --no-cpu 2>> "$O/bench.err" | tee "$O/bench_cfg5_h20_n${n}_warm${w}.json" | python -c "$pick" "cfg5 n$n warm=$w"
  done
  WBC_ROLLOUT_WARM=$w python bench.py --config 5 --tracking --steps 50 --warmup 5 --no-cpu 2>> "$O/bench.err" | tee "$O/bench_cfg5_tracking_warm${w}.json" | python -c "$pick" "cfg5 tracking n1024 warm=$w"
done
python bench.py --config 5 --steps 100 --warmup 10 > "$O/bench_cfg5_h20_n1024.json" 2>> "$O/bench.err"
python -c "
import json; d=json.load(open('$O/bench_cfg5_h20_n1024.json')); print(json.dumps({k: d[k] for k in ('value','us_per_tick','roofline','cpu_baseline')}, indent=1)[:3000])"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_cfg5 -- python3 "$R/bench.py" --config 5 --steps 50 --warmup 5 --no-cpu > /dev/null 2>> "$O/rocprof.err"
find "$O" -name "*kernel_trace.csv" -delete
cat "$O"/stats_cfg5_kernel_stats.csv | head -8

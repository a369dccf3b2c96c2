#!/bin/bash
# round 4: warm set-up that drops rows with negative multipliers -- tests, closed loops, rollouts
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04_drop"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_bench_contract.py tests/test_gpu_warm.py tests/test_gpu_parity.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
for spec in "2 4096" "3 4096" "3 8192" "4 8192"; do set -- $spec
  python bench.py --config $1 --batch $2 --steps 50 --warmup 5 --no-cpu --no-latency --large-batch 0 --closed-loop > "$O/bench_closed_loop_cfg$1_n$2.json" 2>> "$O/bench.err"
  python -c "
import json; d=json.load(open('$O/bench_closed_loop_cfg$1_n$2.json')); c=d['closed_loop']
print('cfg$1 n=$2: value %.1f M/s | closed loop cold %.2f us (kernels %.1f, iters %.2f) warm %.2f us (kernels %.1f, iters %.2f) speedup %.3f' % (d['value']/1e6, c['cold']['us_per_tick'], c['cold']['kernels_sum_us'], c['cold']['qp_iters_mean'], c['warm']['us_per_tick'], c['warm']['kernels_sum_us'], c['warm']['qp_iters_mean'], c['speedup_wall']))"
done
pick='import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; print("%-34s %8.1f M steps/s  %7.2f us/tick  launch %s us" % (sys.argv[1], d["value"]/1e6, d["us_per_tick"], r.get("avg_launch_us")))'
for n in 1024 128; do
  python bench.py --config 5 --steps 50 --warmup 5 --batch $n --no-cpu 2>> "$O/bench.err" | python -c "$pick" "cfg5 n$n"
done
python bench.py --config 5 --tracking --steps 50 --warmup 5 --no-cpu 2>> "$O/bench.err" | python -c "$pick" "cfg5 tracking n1024"
python tools/warm_loop.py 1024 4096 8192 2>> "$O/bench.err" | tee "$O/warm_loop.log"

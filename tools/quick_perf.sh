#!/bin/bash
# quick A/B numbers on the GPU box: default bench, N = 262144, rollouts, cfg 3 / cfg 4 at 32768
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; print("%-28s %10.1f M steps/s  %8.4f ms/step  fused %s  sweep %s  qp %s  rnea %s  iters %.2f" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], k.get("fused_tick_us"), k.get("dyn_sweep_us"), k.get("qp_us"), k.get("rnea_step_us"), (d.get("qp") or {}).get("iters_mean", 0) or 0))'
$B --steps 200 --warmup 20 | python -c "$pick" "cfg2 n4096 f64"
$B --steps 200 --warmup 20 --config 3 | python -c "$pick" "cfg3 n4096 f64 obs"
$B --steps 50 --warmup 5 --batch 262144 | python -c "$pick" "cfg2 n262144 f64"
$B --steps 50 --warmup 5 --batch 262144 --config 3 | python -c "$pick" "cfg3 n262144 f64 obs"
$B --steps 100 --warmup 10 --batch 32768 | python -c "$pick" "cfg2 n32768 f64"
$B --steps 100 --warmup 10 --batch 32768 --config 4 | python -c "$pick" "cfg4 n32768 f32 obs"
$B --steps 50 --warmup 5 --batch 262144 --config 4 | python -c "$pick" "cfg4 n262144 f32 obs"
$B --steps 100 --warmup 10 --batch 4096 --no-mats | python -c "$pick" "cfg2 n4096 nomats"
$B --steps 50 --warmup 5 --batch 262144 --no-mats | python -c "$pick" "cfg2 n262144 nomats"
python bench.py --config 5 --steps 50 --warmup 5 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("cfg5 h20 n1024: %.1f M steps/s, %.2f us/tick" % (d["value"]/1e6, d["us_per_tick"]))'
python bench.py --config 5 --steps 50 --warmup 5 --batch 128 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("cfg5 h20 n128: %.1f M steps/s, %.2f us/tick" % (d["value"]/1e6, d["us_per_tick"]))'

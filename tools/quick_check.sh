#!/bin/bash
# quick check: warm tests, rollouts, the headline tick, a closed loop of warm ticks (tools/quick_check.sh [pytest -k expression])
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/quick_check"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -m pytest tests/test_gpu_warm.py tests/test_gpu_parity.py -q -x -k "${1:-warm or rollout}" > "$O/pytest.log" 2>&1; tail -3 "$O/pytest.log"
pick='import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; print("%-34s %8.1f M steps/s  %7.2f us/tick  launch %s us" % (sys.argv[1], d["value"]/1e6, d["us_per_tick"], r.get("avg_launch_us")))'
for n in 1024 128; do
  python bench.py --config 5 --steps 50 --warmup 5 --batch $n --no-cpu 2>> "$O/bench.err" | python -c "$pick" "cfg5 n$n"
done
python bench.py --config 5 --tracking --steps 50 --warmup 5 --no-cpu 2>> "$O/bench.err" | python -c "$pick" "cfg5 tracking n1024"
python bench.py --steps 300 --warmup 30 --no-cpu --no-latency --large-batch 0 2>> "$O/bench.err" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("cfg2 n4096: %.1f M steps/s, fused %.2f us" % (d["value"]/1e6, d["kernels"]["fused_tick_us"]))'
python tools/warm_loop.py 2>> "$O/bench.err"

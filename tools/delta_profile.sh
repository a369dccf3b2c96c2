#!/bin/bash
# A short visit after a change that touches one plan only (round 6: the pair tick): GPU tests, smoke, the C++ consumers, the driver's bench line, the steps/s-vs-N sweep and the
# rocprofv3 kernel statistics of the changed plan.  tools/final_profile.sh remains the full visit.   usage (on the GPU box): bash tools/delta_profile.sh <tag>
set -u
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; T="${1:-delta}"; O="$R/gpurun_out/$T"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -m pytest tests -q -m gpu > "$O/pytest_gpu_full.log" 2>&1; grep -E "passed|failed|error" "$O/pytest_gpu_full.log" | tail -3 > "$O/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1
./tools/abi_smoke.bin > "$O/abi_smoke.log" 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver_flags_steps20.json" 2>> "$O/bench.err"
for n in 6144 8192; do
  python bench.py --steps 200 --warmup 20 --batch $n --no-cpu --no-latency --large-batch 0 --no-closed-loop > "$O/bench_cfg2_n$n.json" 2>> "$O/bench.err"
done
bash tools/n_sweep.sh 2>/dev/null > "$O/n_sweep.csv"
timeout 600 python tools/soak.py 800 71 f64 2>&1 | tail -1 > "$O/soak.log"
cd /tmp
for n in 6144 8192; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_n$n -- python3 "$R/bench.py" --steps 200 --warmup 20 --no-cpu --no-latency --large-batch 0 --no-closed-loop --batch $n > "$O/bench_under_rocprof_n$n.json" 2>> "$O/rocprof.err"
done
find "$O" -name "*kernel_trace.csv" -delete
find "$O" -name "*agent_info.csv" -delete
cat "$O/pytest_gpu.log" "$O/smoke.log" "$O/abi_smoke.log" "$O/soak.log"; head -3 "$O"/stats_n8192*kernel_stats.csv; cat "$O/n_sweep.csv"

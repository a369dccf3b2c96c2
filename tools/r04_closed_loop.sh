#!/bin/bash
# round 4: the bench's closed-loop leg and the large closed-loop table, re-taken on their own (same files as tools/final_profile.sh writes)
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/final"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_bench_contract.py tests/test_gpu_warm.py -q 2>&1 | grep -E "passed|failed" | tail -2
for spec in "2 4096" "3 8192" "3 65536" "3 262144" "4 262144"; do set -- $spec
  python bench.py --config $1 --batch $2 --steps 50 --warmup 5 --no-cpu --no-latency --large-batch 0 --closed-loop > "$O/bench_closed_loop_cfg$1_n$2.json" 2>> "$O/bench.err"
  python -c "
import json; d=json.load(open('$O/bench_closed_loop_cfg$1_n$2.json')); c=d['closed_loop']
print('cfg$1 n=$2: value %.1f M/s | closed loop cold %.2f us (%.1f M/s, kernels %.1f) warm %.2f us (%.1f M/s, kernels %.1f) speedup %.3f' % (d['value']/1e6, c['cold']['us_per_tick'], c['cold']['value']/1e6, c['cold']['kernels_sum_us'], c['warm']['us_per_tick'], c['warm']['value']/1e6, c['warm']['kernels_sum_us'], c['speedup_wall']))"
done
WARM_LOOP_LANE=1 timeout 1200 python tools/warm_loop.py 16384 32768 49152 65536 131072 262144 > "$O/warm_loop_large.log" 2>> "$O/bench.err"
cut -d'|' -f1 "$O/warm_loop_large.log"

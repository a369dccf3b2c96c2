#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05n; mkdir -p $O
pick='import sys,json; d=json.loads(sys.stdin.read()); print("%-34s %8.1f M/s %8.4f ms/step  blocks %d min %.4f med %.4f max %.4f" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], d["timing"]["blocks"], d["timing"]["block_ms_min"], d["timing"]["block_ms_median"], d["timing"]["block_ms_max"]))'
for rep in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-latency --large-batch 0 2>/dev/null | python -c "$pick" "default steps20 rep$rep"
  HSA_ENABLE_INTERRUPT=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-latency --large-batch 0 2>/dev/null | python -c "$pick" "HSA_ENABLE_INTERRUPT=0 steps20 rep$rep"
  python bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu --no-latency --large-batch 0 2>/dev/null | python -c "$pick" "default steps300 rep$rep"
  HSA_ENABLE_INTERRUPT=0 python bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu --no-latency --large-batch 0 2>/dev/null | python -c "$pick" "HSA_ENABLE_INTERRUPT=0 steps300 rep$rep"
done > $O/ab_host_wait.log 2>&1
cat $O/ab_host_wait.log

#!/bin/bash
# A/B of library builds over the tick workloads (configs 2, 3 at 4 096 states, fp32 shard at 32 768, fp64 at 262 144) and the rollouts:
#   tools/ab_ticks.sh <tag> libA libB ...      -> gpurun_out/<tag>.log
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
tag="$1"; shift
{
tools/ab_libs.sh tests "tests -m gpu -x" "$1"
for r in 1 2; do tools/ab_libs.sh "--steps 300 --warmup 30 --no-cpu --no-latency --large-batch 0 --no-closed-loop" "$@"; done
tools/ab_libs.sh "--config 3 --steps 300 --warmup 30 --no-cpu --no-latency --large-batch 0 --no-closed-loop" "$@"
tools/ab_libs.sh "--config 4 --batch 32768 --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --no-closed-loop" "$@"
tools/ab_libs.sh "--batch 262144 --steps 50 --warmup 5 --no-cpu --no-latency --large-batch 0 --no-closed-loop" "$@"
tools/ab_libs.sh "--config 5 --steps 100 --warmup 10 --no-closed-loop" "$@"
tools/ab_libs.sh "--config 5 --dtype f32 --steps 100 --warmup 10 --no-closed-loop" "$@"
tools/ab_libs.sh "--config 5 --tracking --steps 100 --warmup 10 --no-closed-loop" "$@"
} > gpurun_out/$tag.log 2>&1
grep -E "passed|failed|rep 2|^E " gpurun_out/$tag.log | tail -60

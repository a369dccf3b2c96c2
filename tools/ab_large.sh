#!/bin/bash
# A/B of library builds over the large-batch workloads (262 144 and 65 536 states, with / without M, h, Jc): tools/ab_large.sh <tag> libA libB
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
tag="$1"; shift
{
for c in 2 3 4; do tools/ab_libs.sh "--config $c --batch 262144 --steps 50 --warmup 5 --no-cpu --no-latency --large-batch 0 --no-closed-loop" "$@"; done
for c in 2 3 4; do tools/ab_libs.sh "--config $c --batch 65536 --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --no-closed-loop" "$@"; done
tools/ab_libs.sh "--config 2 --batch 262144 --steps 50 --warmup 5 --no-cpu --no-latency --large-batch 0 --no-closed-loop --no-mats" "$@"
tools/ab_libs.sh "--config 2 --batch 16384 --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --no-closed-loop" "$@"
} > gpurun_out/$tag.log 2>&1
grep -E "rep 2" gpurun_out/$tag.log | tail -40

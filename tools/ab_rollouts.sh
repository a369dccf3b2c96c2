#!/bin/bash
# rollouts A/B of two library builds over the five rollout workloads: tools/ab_rollouts.sh <tag> libA libB
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
tag="$1"; shift
{
tools/ab_libs.sh tests "tests -m gpu -k rollout" "$1"
for r in 1 2; do tools/ab_libs.sh "--config 5 --steps 100 --warmup 10 --no-closed-loop" "$@"; done
tools/ab_libs.sh "--config 5 --dtype f32 --steps 100 --warmup 10 --no-closed-loop" "$@"
tools/ab_libs.sh "--config 5 --batch 128 --steps 100 --warmup 10 --no-closed-loop" "$@"
tools/ab_libs.sh "--config 5 --tracking --steps 100 --warmup 10 --no-closed-loop" "$@"
WBC_ROLLOUT_WARM=0 tools/ab_libs.sh "--config 5 --steps 100 --warmup 10 --no-closed-loop" "$@"
} > gpurun_out/$tag.log 2>&1
grep -E "passed|failed|rep 2|^E " gpurun_out/$tag.log | tail -60

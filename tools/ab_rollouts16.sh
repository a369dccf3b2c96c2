#!/bin/bash
# rollouts of more than 1 024 robots (16-state workgroups): tools/ab_rollouts16.sh <tag> libA libB
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
tag="$1"; shift
{
tools/ab_libs.sh tests "tests -m gpu -k rollout" "$1"
for n in 2048 4000 8192; do tools/ab_libs.sh "--config 5 --batch $n --steps 40 --warmup 5 --no-cpu --no-closed-loop" "$@"; done
tools/ab_libs.sh "--config 5 --batch 2048 --dtype f32 --steps 40 --warmup 5 --no-cpu --no-closed-loop" "$@"
tools/ab_libs.sh "--config 5 --batch 2048 --tracking --steps 40 --warmup 5 --no-cpu --no-closed-loop" "$@"
WBC_ROLLOUT_WARM=0 tools/ab_libs.sh "--config 5 --batch 2048 --steps 40 --warmup 5 --no-cpu --no-closed-loop" "$@"
} > gpurun_out/$tag.log 2>&1
grep -E "passed|failed|rep 2|^E " gpurun_out/$tag.log | tail -60

#!/bin/bash
# A/B of the four-wavefront layout of the 4-state rollout workgroups (WBC_RO_MERGE): tests, then alternating bench lines
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
tools/ab_libs.sh tests "tests -m gpu -k rollout" lib lib_m0
for r in 1 2; do tools/ab_libs.sh "--config 5 --steps 100 --warmup 10 --no-closed-loop" lib lib_m0; done
tools/ab_libs.sh "--config 5 --dtype f32 --steps 100 --warmup 10 --no-closed-loop" lib lib_m0
tools/ab_libs.sh "--config 5 --batch 128 --steps 100 --warmup 10 --no-closed-loop" lib lib_m0
tools/ab_libs.sh "--config 5 --tracking --steps 100 --warmup 10 --no-closed-loop" lib lib_m0
WBC_ROLLOUT_WARM=0 tools/ab_libs.sh "--config 5 --steps 100 --warmup 10 --no-closed-loop" lib lib_m0
} > gpurun_out/r05o_merge.log 2>&1
grep -E "passed|failed|rep|^E " gpurun_out/r05o_merge.log | tail -60

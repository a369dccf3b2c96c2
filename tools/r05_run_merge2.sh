#!/bin/bash
# rollouts: lib (current) against lib_m0 (eight-wavefront layout) and lib_p2 (current + the fp32 shortenings of the integrator's phase 2 also in fp64)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
tools/ab_libs.sh tests "tests -m gpu -k rollout" lib lib_p2
for r in 1 2; do tools/ab_libs.sh "--config 5 --steps 100 --warmup 10 --no-closed-loop" lib lib_p2 lib_m0; done
tools/ab_libs.sh "--config 5 --dtype f32 --steps 100 --warmup 10 --no-closed-loop" lib lib_m0
tools/ab_libs.sh "--config 5 --batch 128 --steps 100 --warmup 10 --no-closed-loop" lib lib_p2 lib_m0
tools/ab_libs.sh "--config 5 --tracking --steps 100 --warmup 10 --no-closed-loop" lib lib_p2 lib_m0
WBC_ROLLOUT_WARM=0 tools/ab_libs.sh "--config 5 --steps 100 --warmup 10 --no-closed-loop" lib lib_p2 lib_m0
} > gpurun_out/r05p_merge2.log 2>&1
grep -E "passed|failed|rep 2|^E " gpurun_out/r05p_merge2.log | tail -60

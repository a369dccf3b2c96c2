#!/bin/bash
# A/B of the lane-split rnea role (RS_LANE2) in the 4-state rollouts: tests first, then three alternations of the bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
tools/ab_libs.sh tests "tests -m gpu -k rollout" lib lib_l0
for r in 1 2 3; do tools/ab_libs.sh "--config 5 --steps 100 --warmup 10 --no-closed-loop" lib lib_l0; done
} > gpurun_out/r05n_lane2.log 2>&1
tail -40 gpurun_out/r05n_lane2.log

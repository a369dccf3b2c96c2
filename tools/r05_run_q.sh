#!/bin/bash
# A/B of the integrator's phase 2 in the rollouts: q0 = as before, q1 = + quaternion series, q2 = + unit quaternion before the barrier, lib = + W rl = Mb (A rl)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
tools/ab_libs.sh tests "tests -m gpu -k rollout" lib
for r in 1 2 3; do tools/ab_libs.sh "--config 5 --steps 100 --warmup 10 --no-closed-loop" lib_q0 lib_q1 lib_q2 lib; done
tools/ab_libs.sh "--config 5 --dtype f32 --steps 100 --warmup 10 --no-closed-loop" lib_q0 lib
tools/ab_libs.sh "--config 5 --batch 128 --steps 100 --warmup 10 --no-closed-loop" lib_q0 lib
} > gpurun_out/r05n_phase2.log 2>&1
tail -50 gpurun_out/r05n_phase2.log

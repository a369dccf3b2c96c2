#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05i; mkdir -p $O
python -m pytest tests -q -m gpu -k "rollout or reference or scenario or warm or integr" > $O/pytest_gpu_rollout.log 2>&1; tail -5 $O/pytest_gpu_rollout.log
bash tools/ab_libs.sh "--config 5 --steps 100 --warmup 10" lib lib_u0 lib_s0 > $O/ab_rollout_phase2_n1024.log 2>&1
bash tools/ab_libs.sh "--config 5 --dtype f32 --steps 100 --warmup 10" lib lib_u0 lib_s0 > $O/ab_rollout_phase2_f32.log 2>&1
bash tools/ab_libs.sh "--config 5 --tracking --steps 100 --warmup 10" lib lib_u0 > $O/ab_rollout_phase2_tracking.log 2>&1
cat $O/ab_*.log

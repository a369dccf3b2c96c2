#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05d; mkdir -p $O
bash tools/ab_libs.sh "--config 5 --steps 100 --warmup 10" lib lib_ro_j1 lib_ro_j2 lib_ro_j3 lib_ro_i1 lib_ro_i2 lib_ro_i3 > $O/ab_rollout_roles_n1024.log 2>&1
bash tools/ab_libs.sh "--config 5 --tracking --steps 100 --warmup 10" lib lib_ro_j2 lib_ro_i2 > $O/ab_rollout_roles_tracking.log 2>&1
bash tools/ab_libs.sh "--config 5 --batch 128 --steps 100 --warmup 10" lib lib_ro_j2 lib_ro_i2 > $O/ab_rollout_roles_n128.log 2>&1
cat $O/ab_*.log

#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05g; mkdir -p $O
python -m pytest tests -q -m gpu -k "rollout or reference or scenario or warm" > $O/pytest_gpu_rollout.log 2>&1; tail -5 $O/pytest_gpu_rollout.log
bash tools/ab_libs.sh "--config 5 --steps 100 --warmup 10" lib lib_ro_a lib_ro_b > $O/ab_rollout_reslds_n1024.log 2>&1
bash tools/ab_libs.sh "--config 5 --tracking --steps 100 --warmup 10" lib lib_ro_a lib_ro_b > $O/ab_rollout_reslds_tracking.log 2>&1
bash tools/ab_libs.sh "--config 5 --batch 128 --steps 100 --warmup 10" lib lib_ro_a lib_ro_b > $O/ab_rollout_reslds_n128.log 2>&1
bash tools/ab_libs.sh "--config 5 --batch 4096 --steps 50 --warmup 10" lib lib_ro_a lib_ro_b > $O/ab_rollout_reslds_n4096.log 2>&1
bash tools/ab_libs.sh "--config 5 --dtype f32 --steps 100 --warmup 10" lib lib_ro_a lib_ro_b > $O/ab_rollout_reslds_f32.log 2>&1
WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_rstamp/libwbc_hip.so python tools/rollout_stamp.py 1024 4 > $O/rollout_timeline_spw4.txt 2>&1
cat $O/ab_*.log $O/rollout_timeline_spw4.txt

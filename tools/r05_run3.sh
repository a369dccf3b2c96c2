#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c; mkdir -p $O
WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_rstamp/libwbc_hip.so python tools/rollout_stamp.py 1024 4 > $O/rollout_timeline_spw4.txt 2>&1
WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_rstamp/libwbc_hip.so python tools/rollout_stamp.py 1024 16 > $O/rollout_timeline_spw16.txt 2>&1
python -m pytest tests -q -m gpu > $O/pytest_gpu_full.log 2>&1; tail -15 $O/pytest_gpu_full.log > $O/pytest_gpu.log
bash tools/ab_libs.sh "--config 5 --steps 100 --warmup 10" lib lib_ro_a > $O/ab_rollout_n1024.log 2>&1
bash tools/ab_libs.sh "--config 5 --dtype f32 --steps 100 --warmup 10" lib lib_ro_a > $O/ab_rollout_f32.log 2>&1
cat $O/rollout_timeline_spw4.txt $O/rollout_timeline_spw16.txt $O/pytest_gpu.log $O/ab_*.log

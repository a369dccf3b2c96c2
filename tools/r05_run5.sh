#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05e; mkdir -p $O
WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_rstamp/libwbc_hip.so python tools/rollout_stamp.py 1024 4 > $O/rollout_timeline_spw4.txt 2>&1
cat $O/rollout_timeline_spw4.txt

#!/bin/bash
# A/B of warm per-lane variants (library builds under wbc_quadruped_dob_amd/): tools/r04_warmlane_ab.sh "sizes" lib ...
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04_warmlane
sizes="$1"; shift
for L in "$@"; do
  echo "== $L"
  WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so WARM_LOOP_LANE=1 timeout 900 python tools/warm_loop.py $sizes 2>> gpurun_out/r04_warmlane/err.log | cut -d'|' -f1,3 | tee -a gpurun_out/r04_warmlane/ab.txt
done

#!/bin/bash
# Diagnostic: the persistent rollout with roles knocked out (libraries built with -DWBC_RO_KNOCK=<bits>: 1 integrator, 2 observer joint rows, 4 mass_jac,
# 8 observer base rows; results are garbage, only the time per tick means something): what is left tells which chain a tick waits for.
#   for k in 1 2 3 5 9 15; do make -C wbc_quadruped_dob_amd/csrc -j8 LIBDIR=../lib_k$k EXTRA=-DWBC_RO_KNOCK=$k; done;  tools/ro_knock.sh
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
libs="lib"; for k in 1 2 3 5 9 15; do [ -d wbc_quadruped_dob_amd/lib_k$k ] && libs="$libs lib_k$k"; done
bash tools/ab_libs.sh "--config 5 --steps 100 --warmup 10" $libs

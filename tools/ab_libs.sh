#!/bin/bash
# A/B of library builds on one device: tools/ab_libs.sh "<bench args>" libdirA libdirB ...   (directories under wbc_quadruped_dob_amd/, e.g. lib lib_remap)
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
args="$1"; shift
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; print("%-40s %8.1f M/s %8.4f ms/step fused %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], k.get("fused_tick_us")))'
for rep in 1 2 3; do for L in "$@"; do
  WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so python bench.py --no-cpu --no-latency --large-batch 0 $args 2>/dev/null | python -c "$pick" "$L [$args] rep $rep"
done; done

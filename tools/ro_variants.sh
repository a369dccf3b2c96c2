#!/bin/bash
# rollout role placements (diagnostic libraries lib_ro_*): plain and tracking rollouts of 1 024 / 128 robots per library
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
pick='import sys,json; d=json.loads(sys.stdin.read()); print("%-26s %8.1f M steps/s  %7.2f us/tick" % (sys.argv[1], d["value"]/1e6, d["us_per_tick"]))'
for rep in 1 2; do
for L in lib $(ls wbc_quadruped_dob_amd | grep lib_ro_); do
  export WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so
  python bench.py --config 5 --steps 50 --warmup 5 --no-cpu 2>/dev/null | python -c "$pick" "$L plain n1024"
  python bench.py --config 5 --steps 50 --warmup 5 --no-cpu --tracking 2>/dev/null | python -c "$pick" "$L tracking n1024"
  python bench.py --config 5 --steps 50 --warmup 5 --no-cpu --batch 128 2>/dev/null | python -c "$pick" "$L plain n128"
done
done

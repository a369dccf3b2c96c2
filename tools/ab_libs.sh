#!/bin/bash
# A/B of library builds on one device (directories under wbc_quadruped_dob_amd/, e.g. lib lib_variant; a variant is built with
#   make -C wbc_quadruped_dob_amd/csrc -j8 LIBDIR=../lib_variant EXTRA=-DSOME_MACRO=1 ).  Replaces round 4's one-off r04_*.sh scripts.
#   tools/ab_libs.sh "<bench args>" libA libB ...          bench.py lines: steps/s, ms per step, the tick's kernels; with --closed-loop the cold / warm loop,
#                                                           with --config 5 the rollout's time per tick
#   tools/ab_libs.sh loops "<sizes>" libA libB ...         tools/warm_loop.py with WARM_LOOP_LANE=1: cold / warm without the per-lane pair / warm per-lane pair / warm
#                                                           one-wavefront kernel forced, per size and workload
#   tools/ab_libs.sh tests "<pytest args>" libA ...        the GPU tests against a library build
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
what="$1"; shift
if [ "$what" = "loops" ]; then
  sizes="$1"; shift
  for L in "$@"; do
    echo "== $L"
    WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so WARM_LOOP_LANE=1 timeout 1500 python tools/warm_loop.py $sizes 2>/dev/null
  done
  exit 0
fi
if [ "$what" = "tests" ]; then
  targs="$1"; shift
  for L in "$@"; do
    echo "== $L"
    WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so timeout 1500 python -m pytest $targs -q 2>&1 | grep -E "passed|failed|^E " | tail -8
  done
  exit 0
fi
args="$what"
pick='
import sys, json
d = json.loads(sys.stdin.read()); k = d.get("kernels") or {}
f = lambda x: "-" if x is None else "%.1f" % x
line = "%-44s %8.1f M/s %8.4f ms/step" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"])
if "us_per_tick" in d: line += "  %.2f us/tick" % d["us_per_tick"]
if k: line += "  fused %s sweep %s front2 %s qp %s lane %s" % tuple(f(k.get(x)) for x in ("fused_tick_us", "dyn_sweep_us", "rnea_step_us", "qp_us", "qp_lane_us"))
c = d.get("closed_loop")
if c and "cold" in c: line += "  | closed loop cold %.2f us (kernels %.1f) warm %.2f us (kernels %.1f)" % (c["cold"]["us_per_tick"], c["cold"]["kernels_sum_us"], c["warm"]["us_per_tick"], c["warm"]["kernels_sum_us"])
print(line)'
for rep in 1 2 3; do for L in "$@"; do
  WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so python bench.py --no-cpu --no-latency --large-batch 0 $args 2>/dev/null | python -c "$pick" "$L [$args] rep $rep"
done; done

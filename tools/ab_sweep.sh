#!/bin/bash
# One parametrised A/B sweep over kernel-selection switches (replaces round 3's tools/r03_*.sh one-offs):
#   tools/ab_sweep.sh "<configs>" "<batch sizes>" "<ENV=VAL[,ENV=VAL...]:label>" ...
# every variant is a set of WBC_* variables (mapped onto wbc_solver_options by the Python binding) and a label; "-:label" = defaults.
#   tools/ab_sweep.sh "2 3" "65536 98304" "WBC_QP_LANE=-1:tiles" "WBC_QP_LANE=1:lane" "-:default"
# Prints one line per (config, batch, variant): steps/s, ms per step, per-kernel times of the sampled ticks.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cfgs="$1"; sizes="$2"; shift 2
B="python bench.py --no-cpu --no-latency --large-batch 0 ${AB_EXTRA:-}"   # (AB_EXTRA: more bench flags, e.g. --no-mats)
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-44s %8.1f M/s %8.4f ms/step  fused %s sweep %s qp %s lane %s front2 %s  iters %.2f" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us")), d["qp"]["iters_mean"]))'
for n in $sizes; do
  st=$(( 3000000 / n + 20 ))
  for c in $cfgs; do
    for v in "$@"; do
      envs="${v%%:*}"; label="${v##*:}"
      if [ "$envs" = "-" ]; then envs=""; fi
      env $(echo "$envs" | tr ',' ' ') $B --steps $st --warmup 10 --batch $n --config $c 2>/dev/null | python -c "$pick" "cfg$c n$n $label"
    done
  done
done

#!/bin/bash
# A/B on one device: the fp64 observer-off cold tick as 32-state twelve-wavefront workgroups (fused_pair_kernel, WBC_FUSED_PAIR=1) against the one-launch tick
# (WBC_FUSED_PAIR=-1 WBC_FUSED_MAX=65536) and the default plan without the pair (WBC_FUSED_PAIR=-1).   tools/ab_pair.sh [sizes...]
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu --no-latency --large-batch 0 --no-closed-loop"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.2f" % x; print("%-22s n=%-6s %8.1f M steps/s %8.4f ms/step  one-launch %s us  iters %.2f ok %.3f" % (sys.argv[1], sys.argv[2], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), (d.get("qp") or {}).get("iters_mean", 0) or 0, (d.get("qp") or {}).get("status_ok_frac", -1)))'
sizes="${@:-2048 4096 4100 4608 5000 5120 6000 6144 7168 7500 8192 9216 10240 12288 14336 16384 20480 24576}"
for n in $sizes; do
  for rep in 1 2; do
    WBC_FUSED_PAIR=1 $B --steps $(( 1000000 / n + 20 )) --warmup 10 --batch $n 2>/dev/null | python -c "$pick" "pair" "$n"
    WBC_FUSED_PAIR=-1 WBC_FUSED_MAX=65536 $B --steps $(( 1000000 / n + 20 )) --warmup 10 --batch $n 2>/dev/null | python -c "$pick" "one-launch (16-state)" "$n"
    WBC_FUSED_PAIR=-1 $B --steps $(( 1000000 / n + 20 )) --warmup 10 --batch $n 2>/dev/null | python -c "$pick" "default without pair" "$n"
  done
done

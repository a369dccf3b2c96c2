#!/bin/bash
# A/B of two builds of the library over the mid-range sizes where qp_tile_kernel runs (LIBS="lib lib_other")
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
LIBS="${LIBS:-lib lib_nopre}"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-34s %8.1f M/s %8.4f ms/step sweep %s qp %s lane %s rnea %s it %.2f ok %.4f" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us")), f(k.get("rnea_step_us")), (d.get("qp") or {}).get("iters_mean", 0) or 0, (d.get("qp") or {}).get("status_ok_frac", 0)))'
for n in ${SIZES:-16384 24576 32768}; do
  for L in $LIBS; do
    export WBC_LIB=$R/wbc_quadruped_dob_amd/$L/libwbc_hip.so
    st=$(( 3000000 / n + 20 ))
    $B --steps $st --warmup 10 --batch $n | python -c "$pick" "$L cfg2 f64 n$n"
    $B --steps $st --warmup 10 --batch $n --config 3 | python -c "$pick" "$L cfg3 f64 n$n"
    [ -n "${F32:-}" ] && $B --steps $st --warmup 10 --batch $n --config 4 | python -c "$pick" "$L cfg4 f32 n$n"
  done
done
for n in ${BIG:-131072}; do
  for L in $LIBS; do
    export WBC_LIB=$R/wbc_quadruped_dob_amd/$L/libwbc_hip.so
    [ -n "${F32:-}" ] && $B --steps 40 --warmup 10 --batch $n --config 4 | python -c "$pick" "$L cfg4 f32 n$n"
    WBC_QP_LANE=-1 $B --steps 40 --warmup 10 --batch $n | python -c "$pick" "$L cfg2 f64 n$n nolane"
  done
done

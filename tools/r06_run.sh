#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_run.log; : > $L
B="python bench.py --no-cpu --no-latency --large-batch 0 --no-closed-loop"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.2f" % x; print("%-10s %-44s %9.1f M steps/s  %8.4f ms/step  fused %s" % (sys.argv[1], sys.argv[2], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us"))))'
export WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_exp/libwbc_hip.so
for rep in 1 2; do
for n in 32768 49152 65536 98304 131072 262144; do
  st=$(( 3000000 / n + 10 ))
  A="--steps $st --warmup 5 --batch $n --config 4"
  for t in 128 96 64; do
    WBC_TT_STATES=$t $B $A 2>/dev/null | python -c "$pick" "tile$t" "$A" >> $L
  done
done; done
cat $L

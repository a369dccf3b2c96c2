#!/bin/bash
# fp32 tiles after the scalar laundering of the QP weights (159 registers: THREE workgroups per CU, 768 resident): tile sizes around N / 768
# against the one-wave kernel (-1) and the previous defaults, trot batch (configs[3] arithmetic) and standing batch
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.1f" % x; print("%-36s %8.1f M/s %8.4f ms/step sweep %s qp %s rnea %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("rnea_step_us"))))'
run() { n=$1; t=$2; st=$(( 3000000 / n + 20 )); WBC_QP_TILE=$t $B --steps $st --warmup 10 --batch $n --config 4 | python -c "$pick" "cfg4 f32 n$n tile $t"; }
for n in 16384 24576; do for t in -1 32; do run $n $t; done; done
for t in -1 36 40 64; do run 28672 $t; done
for t in -1 44 48 64; do run 32768 $t; done
for t in 56 64 80; do run 40960 $t; done
for t in 64 96; do run 49152 $t; done
for t in 88 96 128 0; do run 65536 $t; done
for t in 104 112 0; do run 81920 $t; done
for t in 128 0; do run 98304 $t; done
WBC_QP_TILE=-1 $B --steps 100 --warmup 10 --batch 24576 --dtype f32 | python -c "$pick" "cfg2 f32 n24576 tile -1"
WBC_QP_TILE=32 $B --steps 100 --warmup 10 --batch 24576 --dtype f32 | python -c "$pick" "cfg2 f32 n24576 tile 32"
WBC_QP_TILE=44 $B --steps 100 --warmup 10 --batch 32768 --dtype f32 | python -c "$pick" "cfg2 f32 n32768 tile 44"
WBC_QP_TILE=64 $B --steps 100 --warmup 10 --batch 32768 --dtype f32 | python -c "$pick" "cfg2 f32 n32768 tile 64"

#!/bin/bash
# round 6, first visit: tile timeline at configs[3]'s shard + baseline numbers of the sizes this round works on
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_first.log; : > $L
WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_tstamp/libwbc_hip.so timeout 300 python tools/tile_stamp.py 32768 f32 4 >> $L 2>&1
WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_tstamp/libwbc_hip.so timeout 300 python tools/tile_stamp.py 32768 f64 3 >> $L 2>&1
B="python bench.py --no-cpu --no-latency --large-batch 0 --no-closed-loop"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; print("%-28s %10.1f M steps/s  %8.4f ms/step  fused %s  sweep %s  qp %s  iters %.2f" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], k.get("fused_tick_us"), k.get("dyn_sweep_us"), k.get("qp_us"), (d.get("qp") or {}).get("iters_mean", 0) or 0))'
for rep in 1 2; do
$B --steps 100 --warmup 10 --batch 32768 --config 4 | python -c "$pick" "cfg4 n32768 f32 obs" >> $L
$B --steps 200 --warmup 20 | python -c "$pick" "cfg2 n4096 f64" >> $L
$B --steps 200 --warmup 20 --batch 6144 | python -c "$pick" "cfg2 n6144 f64" >> $L
$B --steps 200 --warmup 20 --batch 8192 | python -c "$pick" "cfg2 n8192 f64" >> $L
done
cat $L

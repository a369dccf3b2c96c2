#!/bin/bash
# control-steps/s vs batch size: fp64 observer off / on and fp32 observer on, default dispatch
export TMPDIR=/tmp
B="python bench.py --no-cpu --no-latency --large-batch 0"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; print("%s,%d,%.1f,%.5f,%s,%s,%s,%s" % (sys.argv[1], d["config"]["batch_per_gpu"], d["value"]/1e6, d["ms_per_step"], k.get("fused_tick_us"), k.get("dyn_sweep_us"), k.get("qp_us"), k.get("rnea_step_us")))'
echo "workload,batch,Msteps_per_s,ms_per_step,fused_tick_us,dyn_sweep_us,qp_us,rnea_step_us"
for n in 1024 2048 4096 6144 8192 12288 16384 24576 32768 49152 65536 98304 131072 196608 262144; do
  st=$(( 2000000 / n + 20 ))
  $B --steps $st --warmup 10 --batch $n | python -c "$pick" "cfg2_f64_obs_off"
  $B --steps $st --warmup 10 --batch $n --config 3 | python -c "$pick" "cfg3_f64_obs_on"
  $B --steps $st --warmup 10 --batch $n --config 4 | python -c "$pick" "cfg4_f32_obs_on"
done

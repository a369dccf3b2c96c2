set -u
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "fp32_observer_off" 2>&1 | tail -12
B="python bench.py --no-cpu --no-latency --large-batch 0 --no-closed-loop --dtype f32"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d.get("kernels") or {}; f=lambda x: "-" if x is None else "%.2f" % x; print("%-8s %-10s %9.1f M steps/s  %8.4f ms/step  fused %s  sweep %s  qp %s  lane %s" % (sys.argv[1], sys.argv[2], d["value"]/1e6, d["ms_per_step"], f(k.get("fused_tick_us")), f(k.get("dyn_sweep_us")), f(k.get("qp_us")), f(k.get("qp_lane_us"))))'
{
echo "# fp32 observer OFF (configs[1] inputs in fp32): tile tick forced (WBC_TILE_TICK=1 WBC_FUSED_MAX=0) against the default plans"
for n in 8192 9216 10240 12288 16384 24576 32768 49152 65536 98304 131072 262144; do
  st=200; [ $n -gt 40000 ] && st=60; [ $n -gt 140000 ] && st=30
  $B --steps $st --warmup 10 --batch $n 2>/dev/null | python -c "$pick" $n default
  WBC_TILE_TICK=1 WBC_FUSED_MAX=0 $B --steps $st --warmup 10 --batch $n 2>/dev/null | python -c "$pick" $n forced
done
} > gpurun_out/r06q_tile_tick_f32_noobs.log
cat gpurun_out/r06q_tile_tick_f32_noobs.log

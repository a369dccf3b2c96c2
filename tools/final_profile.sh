#!/bin/bash
# End-of-round evidence: GPU tests, smoke, C++ ABI consumer, bench lines for configs 2/3/4, rocprofv3 kernel stats, PMC traffic.
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/final"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -m pytest tests -q -m gpu > "$O/pytest_gpu_full.log" 2>&1; grep -E "passed|failed|error" "$O/pytest_gpu_full.log" | tail -3 > "$O/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1
./tools/abi_smoke.bin > "$O/abi_smoke.log" 2>&1
./tools/fma_probe.bin > "$O/fma_probe.log" 2>&1
./tools/bw_probe.bin > "$O/bw_probe.log" 2>&1
./tools/issue_probe.bin > "$O/issue_probe.log" 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver_flags_steps20.json" 2>> "$O/bench.err"
WBC_KEEP_STRUCTURAL=1 python bench.py --steps 50 --warmup 5 --batch 262144 --no-cpu --no-latency --large-batch 0 > "$O/bench_cfg2_n262144_keep_structural.json" 2>> "$O/bench.err"
WBC_KEEP_STRUCTURAL=1 python bench.py --steps 300 --warmup 30 --no-cpu --no-latency --large-batch 0 > "$O/bench_cfg2_n4096_keep_structural.json" 2>> "$O/bench.err"
WBC_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 20 --warmup 5 --no-latency --large-batch 0 > "$O/bench_scale_legs_1rank.json" 2>> "$O/bench.err"
# the same legs with the hipGraph forms of the gather (opt-in: RCCL capture beside the watchdog thread can stall; bounded by a timeout)
WBC_BENCH_GRAPH_GATHER=1 WBC_BENCH_FORCE_DIST=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-latency --large-batch 0 > "$O/bench_scale_legs_1rank_graph.json" 2>> "$O/bench.err"
# configs[4] rollouts without the warm start (A/B of wbc_solver_options.rollout_warm)
WBC_ROLLOUT_WARM=0 python bench.py --config 5 --steps 100 --warmup 10 --no-cpu > "$O/bench_cfg5_h20_n1024_cold.json" 2>> "$O/bench.err"
WBC_ROLLOUT_WARM=0 python bench.py --config 5 --tracking --steps 100 --warmup 10 --no-cpu > "$O/bench_cfg5_tracking_h20_n1024_cold.json" 2>> "$O/bench.err"
python tools/warm_loop.py 1024 4096 8192 > "$O/warm_loop.log" 2>> "$O/bench.err"
# large closed loops: cold tick / warm one-wavefront kernel / warm per-lane pair (the planner's warm thresholds come from this table)
WARM_LOOP_LANE=1 timeout 900 python tools/warm_loop.py 16384 32768 49152 65536 131072 262144 > "$O/warm_loop_large.log" 2>> "$O/bench.err"
# the closed-loop leg of the bench line (cold against warm ticks of a drifting batch), small and large batches
for spec in "2 4096" "3 8192" "3 65536" "3 262144" "4 262144"; do set -- $spec
  python bench.py --config $1 --batch $2 --steps 50 --warmup 5 --no-cpu --no-latency --large-batch 0 --closed-loop > "$O/bench_closed_loop_cfg$1_n$2.json" 2>> "$O/bench.err"
done
timeout 900 python tools/soak.py 1500 51 f64 2>&1 | tail -1 > "$O/soak.log"; timeout 900 python tools/soak.py 600 52 f32 2>&1 | tail -1 >> "$O/soak.log"
python tools/warm_timing.py > "$O/warm_timing.log" 2>> "$O/bench.err"
bash tools/ab_sweep.sh "2 3" "49152 65536 98304 114688" "-:default" "WBC_QP_LANE=1:lane" > "$O/midrange_f64.log" 2>&1
bash tools/ab_sweep.sh "4" "98304 163840 229376" "-:default" "WBC_QP_LANE=1:lane" > "$O/midrange_f32.log" 2>&1
python bench.py --steps 300 --warmup 30 > "$O/bench_cfg2_n4096.json" 2> "$O/bench.err"
python bench.py --steps 300 --warmup 30 --config 3 --no-cpu --no-latency --large-batch 0 > "$O/bench_cfg3_n4096.json" 2>> "$O/bench.err"
python bench.py --steps 100 --warmup 10 --config 4 --batch 32768 --no-cpu --no-latency --large-batch 0 > "$O/bench_cfg4_f32_n32768.json" 2>> "$O/bench.err"
python bench.py --steps 50 --warmup 5 --batch 262144 --no-cpu --no-latency --large-batch 0 > "$O/bench_cfg2_n262144.json" 2>> "$O/bench.err"
for n in 6144 8192; do python bench.py --steps 200 --warmup 20 --batch $n --no-cpu --no-latency --large-batch 0 --no-closed-loop > "$O/bench_cfg2_n$n.json" 2>> "$O/bench.err"; done   # the pair tick (fused_pair_kernel)
python bench.py --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --no-mats > "$O/bench_cfg2_n4096_nomats.json" 2>> "$O/bench.err"
# tau, f only (no M / h / Jc buffers) at the large batch: what a controller that needs only the torques gets
for c in 2 3 4; do python bench.py --config $c --steps 50 --warmup 5 --batch 262144 --no-cpu --no-latency --large-batch 0 --no-mats > "$O/bench_cfg${c}_n262144_nomats.json" 2>> "$O/bench.err"; done
python bench.py --config 5 --steps 100 --warmup 10 > "$O/bench_cfg5_h20_n1024.json" 2>> "$O/bench.err"
python bench.py --config 5 --steps 100 --warmup 10 --batch 128 > "$O/bench_cfg5_h20_n128.json" 2>> "$O/bench.err"
python bench.py --config 5 --steps 50 --warmup 5 --batch 2048 --no-cpu > "$O/bench_cfg5_h20_n2048.json" 2>> "$O/bench.err"   # 16 states per workgroup (eight-wavefront layout)
python bench.py --config 5 --dtype f32 --steps 100 --warmup 10 --no-cpu > "$O/bench_cfg5_f32_h20_n1024.json" 2>> "$O/bench.err"
python bench.py --config 5 --tracking --steps 100 --warmup 10 > "$O/bench_cfg5_tracking_h20_n1024.json" 2>> "$O/bench.err"
python bench.py --config 5 --steps 20 --warmup 3 --batch 32768 > "$O/bench_cfg5_h20_n32768.json" 2>> "$O/bench.err"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu --no-latency --large-batch 0 > "$O/bench_torchrun_1rank.json" 2>> "$O/bench.err"
python bench.py --steps 50 --warmup 5 --batch 262144 --config 3 --no-cpu --no-latency --large-batch 0 > "$O/bench_cfg3_n262144.json" 2>> "$O/bench.err"
python bench.py --steps 50 --warmup 5 --batch 262144 --config 4 --no-cpu --no-latency --large-batch 0 > "$O/bench_cfg4_f32_n262144.json" 2>> "$O/bench.err"
python bench.py --steps 100 --warmup 10 --batch 32768 --no-cpu --no-latency --large-batch 0 > "$O/bench_cfg2_n32768.json" 2>> "$O/bench.err"
python bench.py --gpus 2 --steps 20 --warmup 5 > "$O/bench_gpus2_bare.json" 2> "$O/bench_gpus2_bare.err"; echo "exit $?" >> "$O/bench_gpus2_bare.err"
python bench.py --gpus 2 --single-process --steps 100 --warmup 10 > "$O/bench_single_process_2shards.json" 2>> "$O/bench.err"
# host time of the one-process path: 8 shards (on however many devices are visible), issue threads against serial issue (host_issue in the line)
python bench.py --gpus 8 --single-process --batch 4096 --steps 200 --warmup 20 > "$O/bench_single_process_8shards_b4096.json" 2>> "$O/bench.err"
python bench.py --gpus 8 --single-process --batch 512 --steps 200 --warmup 20 > "$O/bench_single_process_8shards_b512.json" 2>> "$O/bench.err"
# the two-role front half against the all-in-one observer sweep and the two kernels (fp32 shard of configs[3]; fp64 beside the one-launch tick)
bash tools/ab_sweep.sh "4" "16384 24576 32768" "WBC_OBS_COLAUNCH=-1:allinone" "-:tworoles" "WBC_OBS_COLAUNCH=-1,WBC_OBS_SPLIT_MIN=0:twokernels" > "$O/ab_colaunch_f32.log" 2>&1
bash tools/ab_sweep.sh "3" "13312 14336" "WBC_OBS_COLAUNCH=-1:allinone" "-:tworoles" > "$O/ab_colaunch_f64.log" 2>&1
# role timeline of the last tick of a persistent rollout (make -C wbc_quadruped_dob_amd/csrc -j8 LIBDIR=../lib_rstamp EXTRA="-DWBC_FUSED_STAMP -DWBC_RO_STAMP_ALT")
if [ -f wbc_quadruped_dob_amd/lib_rstamp/libwbc_hip.so ]; then
  WBC_LIB=$R/wbc_quadruped_dob_amd/lib_rstamp/libwbc_hip.so python tools/rollout_stamp.py 1024 4 > "$O/rollout_timeline.txt" 2>> "$O/bench.err"
fi
./tools/mfma_probe.bin > "$O/mfma_probe.log" 2>&1
bash tools/n_sweep.sh 2>/dev/null > "$O/n_sweep.csv"
# role timeline of the fused tick (diagnostic build, made beforehand: make -C wbc_quadruped_dob_amd/csrc -j8 LIBDIR=../lib_fstamp EXTRA=-DWBC_FUSED_STAMP)
if [ -f wbc_quadruped_dob_amd/lib_fstamp/libwbc_hip.so ]; then
  WBC_LIB=$R/wbc_quadruped_dob_amd/lib_fstamp/libwbc_hip.so python tools/fused_stamp.py > "$O/fused_timeline.txt" 2>> "$O/bench.err"
fi
# timeline of the staged QP tiles as their own launch (WBC_TILE_TICK=-1; make ... LIBDIR=../lib_tstamp EXTRA=-DWBC_TILE_STAMP): first ticks and the bench's steady state
if [ -f wbc_quadruped_dob_amd/lib_tstamp/libwbc_hip.so ]; then
  WBC_TILE_TICK=-1 WBC_LIB=$R/wbc_quadruped_dob_amd/lib_tstamp/libwbc_hip.so python tools/tile_stamp.py 32768 f32 4 128 12 5 > "$O/tile_timeline.txt" 2>> "$O/bench.err"
  WBC_TILE_TICK=-1 WBC_LIB=$R/wbc_quadruped_dob_amd/lib_tstamp/libwbc_hip.so python tools/tile_stamp.py 32768 f32 4 128 12 2000 >> "$O/tile_timeline.txt" 2>> "$O/bench.err"
fi
if [ -f wbc_quadruped_dob_amd/lib_qstamp/libwbc_hip.so ]; then
  WBC_LIB=$R/wbc_quadruped_dob_amd/lib_qstamp/libwbc_hip.so python tools/qp_stamp.py > "$O/qp_segments.txt" 2>> "$O/bench.err"
fi
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_cfg5 -- python3 "$R/bench.py" --config 5 --steps 50 --warmup 5 > /dev/null 2>> "$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_qp_general -- python3 "$R/tools/qp_general_profile.py" > "$O/qp_general.json" 2>> "$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_cfg5trk -- python3 "$R/bench.py" --config 5 --tracking --steps 50 --warmup 5 > /dev/null 2>> "$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_n4096 -- python3 "$R/bench.py" --steps 300 --warmup 30 --no-cpu --no-latency --large-batch 0 > "$O/bench_under_rocprof_n4096.json" 2> "$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_n8192 -- python3 "$R/bench.py" --steps 200 --warmup 20 --no-cpu --no-latency --large-batch 0 --no-closed-loop --batch 8192 > "$O/bench_under_rocprof_n8192.json" 2>> "$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_n262144 -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu --no-latency --large-batch 0 --batch 262144 > "$O/bench_under_rocprof_n262144.json" 2>> "$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_n32768 -- python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --batch 32768 > "$O/bench_under_rocprof_n32768.json" 2>> "$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_cfg4_n262144 -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu --no-latency --large-batch 0 --batch 262144 --config 4 > "$O/bench_under_rocprof_cfg4_n262144.json" 2>> "$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_dyn_f32 -- python3 "$R/tools/dyn_only.py" 262144 f32 > /dev/null 2>> "$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_dyn_f32_n32768 -- python3 "$R/tools/dyn_only.py" 32768 f32 > /dev/null 2>> "$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_cfg4_n32768 -- python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --batch 32768 --config 4 > "$O/bench_under_rocprof_cfg4_n32768.json" 2>> "$O/rocprof.err"
WBC_TILE_TICK=-1 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_cfg4_n32768_two_launch -- python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --batch 32768 --config 4 > "$O/bench_under_rocprof_cfg4_n32768_two_launch.json" 2>> "$O/rocprof.err"
# the default bench command itself (incl. its N = 262 144 characterisation legs: tick sweep and the dynamics stage alone)
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o stats_default -- python3 "$R/bench.py" --steps 300 --warmup 30 --no-cpu --no-latency > "$O/bench_under_rocprof_default.json" 2>> "$O/rocprof.err"
find "$O" -name "*kernel_trace.csv" -delete
cd "$R"
bash tools/pmc_profile.sh > "$O/pmc.log" 2>&1
cp gpurun_out/pmc/summary.json "$O/pmc_summary.json"
bash tools/sq_profile.sh 262144 sq_n262144 "1 2 3" > "$O/sq_n262144.log" 2>&1; cp gpurun_out/sq_n262144/summary.json "$O/sq_counters_n262144.json"
bash tools/sq_profile.sh 4096 sq_n4096 "1 2 3" > "$O/sq_n4096.log" 2>&1; cp gpurun_out/sq_n4096/summary.json "$O/sq_counters_n4096.json"
cat "$O/pytest_gpu.log" "$O/smoke.log" "$O/abi_smoke.log" "$O/fma_probe.log"
for f in "$O"/bench_cfg[234]_n*[0-9].json; do echo "== $f"; python3 -c "
import json,sys
r=json.load(open('$f')); k=r['kernels']; print(r['config']['workload'])
us=lambda x: 'n/a' if x is None else '%.1f us'%x
print('  ms/step %.4f  steps/s %.4e  fused %s  dyn %s  qp %s  roofline.frac %s'%(r['ms_per_step'],r['value'],us(k.get('fused_tick_us')),us(k['dyn_sweep_us']),us(k['qp_us']),r['roofline']['frac']))"; done
cat "$O"/stats_n4096_kernel_stats.csv "$O"/stats_n262144_kernel_stats.csv

#!/bin/bash
# tools/collect_profiles.sh r01g : copy what tools/final_profile.sh left under gpurun_out/final/ into profiles/<tag>_* (tracked)
set -eu
T="$1"; O=gpurun_out/final; P=profiles
for f in bench_cfg2_n262144 bench_cfg2_n4096 bench_cfg2_n4096_nomats bench_cfg3_n4096 bench_cfg4_f32_n32768 bench_cfg5_h20_n1024 \
         bench_cfg5_h20_n128 bench_cfg5_h20_n32768 bench_cfg5_tracking_h20_n1024 bench_under_rocprof_n262144 bench_under_rocprof_n4096 bench_torchrun_1rank; do
  cp "$O/$f.json" "$P/${T}_$f.json"
done
for f in abi_smoke bw_probe fma_probe pytest_gpu mfma_probe; do cp "$O/$f.log" "$P/${T}_$f.log"; done
for f in bench_cfg3_n262144 bench_cfg4_f32_n262144 bench_cfg2_n32768 bench_single_process_2shards bench_under_rocprof_n32768 bench_under_rocprof_cfg4_n32768 sq_counters_n262144 sq_counters_n4096; do
  [ -f "$O/$f.json" ] && cp "$O/$f.json" "$P/${T}_$f.json"
done
for f in bench_driver_flags_steps20 bench_cfg2_n262144_keep_structural bench_cfg2_n4096_keep_structural bench_scale_legs_1rank bench_under_rocprof_cfg4_n262144; do
  [ -s "$O/$f.json" ] && cp "$O/$f.json" "$P/${T}_$f.json"
done
for f in bench_scale_legs_1rank_graph bench_cfg5_h20_n1024_cold bench_cfg5_tracking_h20_n1024_cold bench_cfg5_h20_n2048 bench_cfg5_f32_h20_n1024 qp_general; do
  [ -s "$O/$f.json" ] && cp "$O/$f.json" "$P/${T}_$f.json"
done
for f in "$O"/bench_cfg?_n262144_nomats.json; do [ -s "$f" ] && cp "$f" "$P/${T}_$(basename "$f")"; done
for f in "$O"/bench_closed_loop_*.json; do [ -s "$f" ] && cp "$f" "$P/${T}_$(basename "$f")"; done
[ -f "$O/stats_qp_general_kernel_stats.csv" ] && cp "$O/stats_qp_general_kernel_stats.csv" "$P/${T}_kernel_stats_qp_general.csv"
for f in issue_probe tile_sweep midrange midrange_f64 midrange_f32 warm_loop warm_loop_large warm_timing soak; do [ -f "$O/$f.log" ] && cp "$O/$f.log" "$P/${T}_$f.log"; done
[ -f "$O/stats_cfg4_n262144_kernel_stats.csv" ] && cp "$O/stats_cfg4_n262144_kernel_stats.csv" "$P/${T}_kernel_stats_cfg4_f32_n262144.csv"
[ -f "$O/stats_dyn_f32_kernel_stats.csv" ] && cp "$O/stats_dyn_f32_kernel_stats.csv" "$P/${T}_kernel_stats_dyn_alone_f32_n262144.csv"
[ -f "$O/stats_dyn_f32_n32768_kernel_stats.csv" ] && cp "$O/stats_dyn_f32_n32768_kernel_stats.csv" "$P/${T}_kernel_stats_dyn_alone_f32_n32768.csv"
[ -f "$O/bench_gpus2_bare.err" ] && cp "$O/bench_gpus2_bare.err" "$P/${T}_bench_gpus2_bare.log"
[ -f "$O/n_sweep.csv" ] && cp "$O/n_sweep.csv" "$P/${T}_n_sweep.csv"
[ -f "$O/qp_segments.txt" ] && cp "$O/qp_segments.txt" "$P/${T}_qp_segments.txt"
[ -f "$O/stats_n32768_kernel_stats.csv" ] && cp "$O/stats_n32768_kernel_stats.csv" "$P/${T}_kernel_stats_cfg2_n32768.csv"
[ -f "$O/stats_cfg4_n32768_kernel_stats.csv" ] && cp "$O/stats_cfg4_n32768_kernel_stats.csv" "$P/${T}_kernel_stats_cfg4_f32_n32768.csv"
[ -f "$O/fused_timeline.txt" ] && cp "$O/fused_timeline.txt" "$P/${T}_fused_timeline.txt"
[ -f "$O/tile_timeline.txt" ] && cp "$O/tile_timeline.txt" "$P/${T}_tile_timeline.txt"
[ -f "$O/stats_cfg4_n32768_two_launch_kernel_stats.csv" ] && cp "$O/stats_cfg4_n32768_two_launch_kernel_stats.csv" "$P/${T}_kernel_stats_cfg4_f32_n32768_two_launch.csv"
[ -f "$O/rollout_timeline.txt" ] && cp "$O/rollout_timeline.txt" "$P/${T}_rollout_timeline.txt"
for f in bench_single_process_8shards_b4096 bench_single_process_8shards_b512; do [ -s "$O/$f.json" ] && cp "$O/$f.json" "$P/${T}_$f.json"; done
for f in ab_colaunch_f32 ab_colaunch_f64; do [ -f "$O/$f.log" ] && cp "$O/$f.log" "$P/${T}_$f.log"; done
cp "$O/pmc_summary.json" "$P/${T}_pmc_summary.json"; cp "$O/pmc_summary.json" "$P/pmc_latest.json"
cp "$O/stats_n4096_kernel_stats.csv" "$P/${T}_kernel_stats_cfg2_n4096.csv"
cp "$O/stats_n262144_kernel_stats.csv" "$P/${T}_kernel_stats_cfg2_n262144.csv"
cp "$O/stats_cfg5_kernel_stats.csv" "$P/${T}_kernel_stats_cfg5_h20_n1024.csv"
cp "$O/stats_cfg5trk_kernel_stats.csv" "$P/${T}_kernel_stats_cfg5_tracking_h20_n1024.csv"
[ -f "$O/stats_default_kernel_stats.csv" ] && cp "$O/stats_default_kernel_stats.csv" "$P/${T}_kernel_stats_default_bench.csv" && cp "$O/bench_under_rocprof_default.json" "$P/${T}_bench_under_rocprof_default.json"
for f in "$O"/bench_*.json; do [ -s "$f" ] && python3 -c "
import json,sys; r=json.load(open('$f')); print('%-46s ms/step %.5f  value %.4e' % ('$(basename $f)', r['ms_per_step'], r['value']))"; done
grep -h "wbc::" "$P/${T}_kernel_stats_default_bench.csv" "$P/${T}_kernel_stats_cfg2_n4096.csv" "$P/${T}_kernel_stats_cfg2_n262144.csv" "$P/${T}_kernel_stats_cfg5_h20_n1024.csv" | awk -F'",' '{split($1,a,"("); print a[1], $2}' | cut -c1-140
for f in bench_cfg2_n6144 bench_cfg2_n8192 bench_under_rocprof_n8192; do [ -s "$O/$f.json" ] && cp "$O/$f.json" "$P/${T}_$f.json"; done
[ -f "$O/stats_n8192_kernel_stats.csv" ] && cp "$O/stats_n8192_kernel_stats.csv" "$P/${T}_kernel_stats_cfg2_n8192.csv"
[ -f "$O/smoke.log" ] && cp "$O/smoke.log" "$P/${T}_smoke.log"
true

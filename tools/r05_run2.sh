#!/bin/bash
# round 5, GPU visit 2: full GPU tests, rollout variants, the two-role front half, host-issue numbers with the same backend either way
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05b; mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest_gpu_full.log 2>&1; tail -15 $O/pytest_gpu_full.log > $O/pytest_gpu.log
bash tools/ab_libs.sh "--config 5 --steps 100 --warmup 10" lib lib_ro_a lib_ro_b lib_ro_d > $O/ab_rollout_n1024.log 2>&1
bash tools/ab_libs.sh "--config 5 --batch 128 --steps 100 --warmup 10" lib lib_ro_a lib_ro_b > $O/ab_rollout_n128.log 2>&1
bash tools/ab_libs.sh "--config 5 --tracking --steps 100 --warmup 10" lib lib_ro_a lib_ro_b > $O/ab_rollout_tracking.log 2>&1
bash tools/ab_libs.sh "--config 5 --dtype f32 --steps 100 --warmup 10" lib lib_ro_a > $O/ab_rollout_f32.log 2>&1
# the two-role front half (obs_colaunch): fp32 either side of and inside its window, fp64 forced
bash tools/ab_sweep.sh "4" "12290 16384 20480 24576 28672 32768" "WBC_OBS_COLAUNCH=-1:allinone" "-:tworoles" "WBC_OBS_COLAUNCH=-1,WBC_OBS_SPLIT_MIN=0:twokernels" > $O/ab_colaunch_f32.log 2>&1
bash tools/ab_sweep.sh "3" "12800 13312 14336 16384" "-:default" "WBC_OBS_COLAUNCH=1:tworoles" > $O/ab_colaunch_f64.log 2>&1
bash tools/ab_libs.sh "--config 4 --batch 262144 --steps 50 --warmup 5" lib lib_obspk > $O/ab_obspk_n262144.log 2>&1
bash tools/ab_libs.sh "--config 4 --batch 65536 --steps 50 --warmup 5" lib lib_obspk > $O/ab_obspk_n65536.log 2>&1
for b in 4096 512; do
  python bench.py --gpus 8 --single-process --batch $b --steps 200 --warmup 20 > $O/bench_single_process_8shards_b$b.json 2>> $O/bench.err
done
python bench.py --gpus 2 --single-process --batch 2048 --steps 200 --warmup 20 > $O/bench_single_process_2shards_b2048.json 2>> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O" -o stats_cfg5 -- python3 "$GRAFT_REPO_ROOT/bench.py" --config 5 --steps 50 --warmup 5 > /dev/null 2>> "$GRAFT_REPO_ROOT/$O/rocprof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O" -o stats_cfg4_n32768 -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --batch 32768 --config 4 > "$GRAFT_REPO_ROOT/$O/bench_under_rocprof_cfg4_n32768.json" 2>> "$GRAFT_REPO_ROOT/$O/rocprof.err"
cd "$GRAFT_REPO_ROOT"
find $O -name "*kernel_trace.csv" -delete
cat $O/pytest_gpu.log $O/ab_*.log
head -8 $O/stats_cfg5_kernel_stats.csv $O/stats_cfg4_n32768_kernel_stats.csv
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05b/bench_single*.json")):
    try:
        d=json.load(open(f)); print(f, "value %.1f M" % (d["value"]/1e6), "gather %.1f M" % (d["with_tau_allgather"]["value"]/1e6), "serial gather %.1f M" % (d["with_tau_allgather"]["serial"]["value"]/1e6), json.dumps(d["host_issue"]))
    except Exception as e: print(f, "ERR", e)
PY

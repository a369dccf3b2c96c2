#!/bin/bash
# round 5, GPU visit 1: tests at the new code, A/B of the joint-index argument and of the ordered r_prev read, host-issue numbers
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05a; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/pytest_gpu_full.log 2>&1; tail -5 $O/pytest_gpu_full.log > $O/pytest_gpu.log
bash tools/ab_libs.sh "--batch 4096 --steps 300 --warmup 30" lib lib_jidx0 > $O/ab_jidx_cfg2_n4096.log 2>&1
bash tools/ab_libs.sh "--config 3 --batch 4096 --steps 300 --warmup 30" lib lib_jidx0 lib_noorder > $O/ab_jidx_cfg3_n4096.log 2>&1
bash tools/ab_libs.sh "--config 5 --steps 100 --warmup 10" lib lib_jidx0 lib_noorder > $O/ab_jidx_cfg5.log 2>&1
bash tools/ab_libs.sh "--config 4 --batch 32768 --steps 100 --warmup 10" lib lib_jidx0 > $O/ab_jidx_cfg4_n32768.log 2>&1
bash tools/ab_libs.sh "--batch 262144 --steps 50 --warmup 5" lib lib_jidx0 > $O/ab_jidx_cfg2_n262144.log 2>&1
for b in 4096 512; do
  python bench.py --gpus 8 --single-process --batch $b --steps 200 --warmup 20 > $O/bench_single_process_8shards_b$b.json 2>> $O/bench.err
done
python bench.py --gpus 2 --single-process --batch 2048 --steps 200 --warmup 20 > $O/bench_single_process_2shards_b2048.json 2>> $O/bench.err
python bench.py --gpus 2 --single-process --batch 4096 --steps 200 --warmup 20 > $O/bench_single_process_2shards_b4096.json 2>> $O/bench.err
python bench.py --steps 300 --warmup 30 > $O/bench_cfg2_n4096.json 2>> $O/bench.err
cat $O/pytest_gpu.log $O/ab_*.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05a/bench_single*.json")):
    try:
        d=json.load(open(f)); print(f, "value %.1f M" % (d["value"]/1e6), "gather %.1f M" % (d["with_tau_allgather"]["value"]/1e6), "serial gather %.1f M" % (d["with_tau_allgather"]["serial"]["value"]/1e6), json.dumps(d["host_issue"]))
    except Exception as e: print(f, "ERR", e)
d=json.load(open("gpurun_out/r05a/bench_cfg2_n4096.json")); print("default: %.1f M" % (d["value"]/1e6), json.dumps(d.get("closed_loop"))[:1500])
PY

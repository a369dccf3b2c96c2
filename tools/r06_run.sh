#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_run.log; : > $L
./tools/bw_probe.bin 2>&1 | head -12 >> $L
python bench.py --steps 20 --warmup 5 --no-cpu --no-latency --large-batch 0 --no-closed-loop 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("default", d["value"]/1e6, r["kernel"][:40], r["frac"], r["pattern_ceiling"], r["frac_of_pattern_ceiling"])' >> $L
python bench.py --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --no-closed-loop --config 4 --batch 32768 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("cfg4", d["value"]/1e6, r["kernel"][:40], r["frac"], r["pattern_ceiling"], r["frac_of_pattern_ceiling"])' >> $L
timeout 900 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_round6.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -4 >> $L
cat $L

#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_run.log; : > $L
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu 2>&1 | grep -E "passed|failed" >> $L
for n in 32768 16384 24576; do tools/ab_r06.sh "--steps 200 --warmup 20 --batch $n --config 4" lib_prev lib >> $L 2>&1; done
WBC_TILE_TICK=-1 tools/ab_r06.sh "--steps 200 --warmup 20 --batch 32768 --config 4" lib_prev lib >> $L 2>&1
cat $L

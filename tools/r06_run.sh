#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_run.log; : > $L
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "two_pass" 2>&1 | grep -E "passed|failed|Error" >> $L
for n in 4608 5120 6144 7168 8192; do tools/ab_r06.sh "--steps 300 --warmup 30 --batch $n" lib_prev lib 2>&1 | head -4 >> $L; done
tools/ab_r06.sh "--steps 300 --warmup 30 --batch 8192 --no-mats" lib_prev lib 2>&1 | head -4 >> $L
tools/ab_r06.sh "--steps 300 --warmup 30 --batch 6144 --dtype f32" lib_prev lib 2>&1 | head -4 >> $L
cat $L

#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r06_run.log; : > $L
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 >> $L
echo "=== staged tiles, first ticks" >> $L
WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_tstamp/libwbc_hip.so timeout 300 python tools/tile_stamp.py 32768 f32 4 128 12 5 >> $L 2>&1
echo "=== staged tiles, bench steady state" >> $L
WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_tstamp/libwbc_hip.so timeout 300 python tools/tile_stamp.py 32768 f32 4 128 12 2000 >> $L 2>&1
tools/ab_r06.sh "--steps 100 --warmup 10 --batch 32768 --config 4" lib_base lib >> $L 2>&1
tools/ab_r06.sh "--steps 100 --warmup 10 --batch 16384 --config 4" lib_base lib >> $L 2>&1
tools/ab_r06.sh "--steps 100 --warmup 10 --batch 49152 --config 4" lib_base lib >> $L 2>&1
cat $L

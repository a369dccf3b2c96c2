#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05j; mkdir -p $O
bash tools/ab_libs.sh "--config 5 --steps 100 --warmup 10" lib lib_rg > $O/ab_roleunguard_cfg5.log 2>&1
bash tools/ab_libs.sh "--batch 4096 --steps 300 --warmup 30" lib lib_rg > $O/ab_roleunguard_cfg2_n4096.log 2>&1
bash tools/ab_libs.sh "--config 3 --batch 4096 --steps 300 --warmup 30" lib lib_rg > $O/ab_roleunguard_cfg3_n4096.log 2>&1
bash tools/ab_libs.sh "--config 4 --batch 4096 --steps 300 --warmup 30" lib lib_rg > $O/ab_roleunguard_cfg4_n4096.log 2>&1
bash tools/ab_libs.sh "--batch 1000 --steps 300 --warmup 30" lib lib_rg > $O/ab_roleunguard_cfg2_n1000.log 2>&1
bash tools/ab_libs.sh tests "tests -m gpu -x" lib_rg > $O/tests_roleunguard.log 2>&1
cat $O/*.log

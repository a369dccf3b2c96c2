#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05l; mkdir -p $O
bash tools/ab_libs.sh "--config 4 --batch 32768 --steps 100 --warmup 10" lib lib_p1 > $O/ab_prio_cfg4_n32768.log 2>&1
bash tools/ab_libs.sh "--batch 4096 --steps 300 --warmup 30" lib lib_p1 lib_p2 > $O/ab_prio_cfg2_n4096.log 2>&1
bash tools/ab_libs.sh "--config 3 --batch 4096 --steps 300 --warmup 30" lib lib_p1 lib_p2 > $O/ab_prio_cfg3_n4096.log 2>&1
bash tools/ab_libs.sh "--config 5 --steps 100 --warmup 10" lib lib_p1 > $O/ab_prio_cfg5.log 2>&1
cat $O/*.log

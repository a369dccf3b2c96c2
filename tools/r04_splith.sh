#!/bin/bash
# A/B: warm fused tick (observer off, fp64, M/h/Jc written) with the bias forces h on a seventh wavefront (lib_splith_warm) against the default
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for L in lib lib_splith_warm; do
  for n in 1024 4096 8192; do
  WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so python bench.py --config 2 --batch $n --steps 100 --warmup 10 --no-cpu --no-latency --large-batch 0 --closed-loop 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['closed_loop']
print('%-18s n=%5d value %.1f M/s | closed loop cold %.2f us (kernel %.2f) warm %.2f us (kernel %.2f)' % ('$L', $n, d['value']/1e6, c['cold']['us_per_tick'], c['cold']['kernels_sum_us'], c['warm']['us_per_tick'], c['warm']['kernels_sum_us']))"
  done
done; done
WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_splith_warm/libwbc_hip.so timeout 600 python -m pytest tests/test_gpu_warm.py -q -x 2>&1 | grep -E "passed|failed" | tail -2
for L in lib lib_splith_warm; do echo "== $L"; WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so python tools/warm_loop.py 1024 4096 8192 2>/dev/null | grep "cfg2" | cut -c1-40,150-330; done

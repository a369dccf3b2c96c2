#!/bin/bash
# A/B: list kernel started from the per-lane kernel's last faces (lib_wl_fromlane) against the default
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print("%-40s %8.1f M/s %8.4f ms/step  lane %s list %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], k.get("qp_lane_us"), k.get("qp_us")))'
for rep in 1 2; do for L in lib lib_wl_fromlane; do for spec in "2 262144" "3 262144" "4 262144" "2 131072" "3 131072"; do set -- $spec
  WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so python bench.py --config $1 --batch $2 --steps 30 --warmup 5 --no-cpu --no-latency --large-batch 0 2>/dev/null | python -c "$pick" "$L cfg$1 n$2"
done; done; done
WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_wl_fromlane/libwbc_hip.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_warm.py -q -x 2>&1 | grep -E "passed|failed" | tail -2
for L in lib lib_wl_fromlane; do echo "== $L"; WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so WARM_LOOP_LANE=1 timeout 600 python tools/warm_loop.py 65536 262144 2>/dev/null | python -c "
import re,sys
for l in sys.stdin:
    m = re.match(r'(cfg\d \w+ obs\d n=\s*\d+).*?cold\s+([\d.]+).*?warm per-lane\s+([\d.]+).*?kernels cold (\{[^}]*\}).*warmlane (\{[^}]*\})', l)
    if m: print(m.group(1), 'cold', m.group(2), m.group(4), '| warm', m.group(3), m.group(5))"
done

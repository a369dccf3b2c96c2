#!/bin/bash
# A/B: fp32 per-lane QP kernel at three wavefronts per SIMD (lib_ql_f32w3: 168 registers, 20 bytes of scratch) against two (172 registers)
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
pick='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print("%-40s %8.1f M/s %8.4f ms/step  lane %s list %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], k.get("qp_lane_us"), k.get("qp_us")))'
for rep in 1 2 3; do for L in lib lib_ql_f32w3; do for n in 262144 229376; do
  WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so python bench.py --config 4 --batch $n --steps 30 --warmup 5 --no-cpu --no-latency --large-batch 0 2>/dev/null | python -c "$pick" "$L cfg4 n$n"
done; done; done
for L in lib lib_ql_f32w3; do echo "== $L"; WBC_LIB=$PWD/wbc_quadruped_dob_amd/$L/libwbc_hip.so WARM_LOOP_LANE=1 timeout 600 python tools/warm_loop.py 36864 65536 131072 262144 2>/dev/null | grep "cfg4" | python -c "
import re,sys
for l in sys.stdin:
    m = re.match(r'(cfg\d \w+ obs\d n=\s*\d+).*?cold\s+([\d.]+).*?warm per-lane\s+([\d.]+).*?kernels cold (\{[^}]*\}).*warmlane (\{[^}]*\})', l)
    if m: print(m.group(1), 'cold', m.group(2), m.group(4), '| warm', m.group(3), m.group(5))"
done

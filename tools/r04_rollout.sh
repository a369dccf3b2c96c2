#!/bin/bash
# round 4: rollout kernel check -- parity tests that touch rollouts, then the configs[4] lines
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04_rollout"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_warm.py tests/test_gpu_parity.py tests/test_gpu_scenarios.py -q -x -k "rollout or tracking or warm" 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
pick='import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; print("%-34s %8.1f M steps/s  %7.2f us/tick  launch %s us" % (sys.argv[1], d["value"]/1e6, d["us_per_tick"], r.get("avg_launch_us")))'
for rep in 1 2; do
for n in 1024 128 4096; do
  python bench.py --config 5 --steps 50 --warmup 5 --batch $n --no-cpu 2>> "$O/bench.err" | python -c "$pick" "cfg5 n$n"
done
python bench.py --config 5 --tracking --steps 50 --warmup 5 --no-cpu 2>> "$O/bench.err" | python -c "$pick" "cfg5 tracking n1024"
WBC_ROLLOUT_WARM=0 python bench.py --config 5 --steps 50 --warmup 5 --no-cpu 2>> "$O/bench.err" | python -c "$pick" "cfg5 n1024 cold"
done
python bench.py --config 5 --dtype f32 --steps 50 --warmup 5 --no-cpu 2>> "$O/bench.err" | python -c "$pick" "cfg5 f32 n1024"
[ -f wbc_quadruped_dob_amd/lib_fstamp/libwbc_hip.so ] && WBC_LIB=$R/wbc_quadruped_dob_amd/lib_fstamp/libwbc_hip.so python tools/fused_stamp.py 2>> "$O/bench.err" | tail -14

#!/bin/bash
# round 4: the warm per-lane QP kernel against the warm one-wavefront kernel and the cold tick, closed loops of drifting states
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04_warmlane"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_warm.py -q -x > "$O/pytest.log" 2>&1; tail -5 "$O/pytest.log"
WARM_LOOP_LANE=1 timeout 1500 python tools/warm_loop.py ${1:-10240 12288 16384 24576 32768 65536 131072 262144} 2>> "$O/err.log" | tee "$O/warm_loop.txt"

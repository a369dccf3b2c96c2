#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, bench, rocprof kernel trace.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
python bench.py --steps 200 --warmup 20 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit: $?" >> gpurun_out/bench.err
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.json
tail -5 gpurun_out/bench.err

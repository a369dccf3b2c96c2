#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, bench (default, bare --gpus 2, single-process 2 shards), rocprof.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit: $?" >> gpurun_out/bench.err
python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/bench_gpus2.json 2> gpurun_out/bench_gpus2.err
echo "bench --gpus 2 exit: $?" >> gpurun_out/bench_gpus2.err
python bench.py --gpus 2 --single-process --steps 20 --warmup 5 > gpurun_out/bench_sp2.json 2> gpurun_out/bench_sp2.err
echo "bench --gpus 2 --single-process exit: $?" >> gpurun_out/bench_sp2.err
python bench.py --steps 50 --warmup 5 --batch 262144 --no-cpu --no-latency --large-batch 0 > gpurun_out/bench_262144.json 2>> gpurun_out/bench.err
if [ "${PROFILE:-1}" = "1" ]; then
  rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r02 -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 200 --warmup 20 --no-cpu --no-latency --large-batch 0 > "$GRAFT_REPO_ROOT/gpurun_out/prof/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof/rocprof.err" )
  find gpurun_out/prof -name "*kernel_trace.csv" -size +2M -delete
fi
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.json gpurun_out/bench_gpus2.err gpurun_out/bench_sp2.json gpurun_out/bench_sp2.err gpurun_out/bench_262144.json
tail -5 gpurun_out/bench.err

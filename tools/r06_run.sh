set -u
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_pytest_gpu.log 2>&1; grep -n "passed\|failed" gpurun_out/r06_pytest_gpu.log | tail -3
timeout 900 python tools/soak.py 300 72 f32 2>&1 | tail -1

set -u
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_pytest_gpu.log 2>&1; grep -n "passed\|failed" gpurun_out/r06_pytest_gpu.log | tail -3
timeout 600 python tools/soak.py 400 91 f64 2>&1 | tail -1
timeout 600 python tools/soak.py 200 92 f32 2>&1 | tail -1

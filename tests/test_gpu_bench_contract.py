"""GPU: `bench.py` keeps the driver's contract -- one JSON line on stdout with the fixed keys, the `roofline` and
`cpu_baseline` objects, numbers that agree with each other -- when it is started the way the driver starts it
(`python bench.py --gpus 1 --steps K --warmup W`) and as the one-rank `torch.distributed.run` job of the N > 1 recipe.
bench.py runs as a CHILD process (it initialises the GPU itself)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"]


def one_line(cmd):
    run = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # ONE line, nothing else on stdout
    return json.loads(lines[0])


def check_common(d, steps, warmup):
    assert d["metric"] == "WBC control-steps/sec (batched DogBot)" and d["unit"] == "control-steps/s"
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "configs[1]" in d["config"]["workload"] and d["config"]["batch_per_gpu"] == 4096
    assert "model" not in d["config"]
    # value = states of the K ticks / the time of the K ticks
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert 1e7 < d["value"] < 1e9
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) <= 1e-6 * r["achieved"]
    assert r["launches_timed"] >= 10                      # measured live, over the timed region
    assert r["avg_launch_us"] * 1e-3 <= d["ms_per_step"] * 1.25   # the dominant kernel fits into a step (event spans read a little long)
    assert "traffic" in r and "NOT collected inside this run" in r["traffic_source"]


def test_driver_command_line():
    d = one_line([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--large-batch", "0", "--no-latency"])
    for k in KEYS:
        assert k in d, k
    check_common(d, 20, 5)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_one_rank_of_the_multi_gpu_recipe():
    env_port = str(29600 + os.getpid() % 300)
    d = one_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                  "--master-port", env_port, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu",
                  "--large-batch", "0", "--no-latency"])
    for k in KEYS[:-1]:
        assert k in d, k
    check_common(d, 20, 5)

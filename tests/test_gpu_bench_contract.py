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
    assert r["avg_launch_us"] * 1e-3 <= d["ms_per_step"] * (1 + 1e-9)   # the span the roofline is priced on fits inside a step
    assert r["event_span_us"] >= r["avg_launch_us"] - 1e-9 and abs(r["step_period_us"] - d["ms_per_step"] * 1e3) < 1e-6
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


def test_scale_legs_of_the_multi_gpu_line_on_one_rank():
    """The N > 1 line carries BASELINE.json's 8-GPU configs as extra legs (scale_config3: 32 768 fp32 states per GPU with the
    observer on; scale_config5: horizon-20 rollouts, 1 024 and 128 per GPU) and the >= 5 ms blocks of the headline config.
    WBC_BENCH_FORCE_DIST=1 makes the one-rank job of this box initialise RCCL and take that path (rccl_ranks = 1)."""
    env = dict(os.environ, WBC_BENCH_FORCE_DIST="1")
    port = str(29900 + os.getpid() % 90)
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", port, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu", "--large-batch", "0", "--no-latency"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    check_common(d, 20, 5)
    assert d["rccl_ranks"] == 1 and d["with_tau_allgather"]["value"] > 0
    lb = d["value_long_blocks"]
    assert lb["steps_per_block"] >= 20 and lb["block_ms_median"] >= 4.0 and 0.7 < lb["value"] / d["value"] < 1.5
    c3 = d["scale_config3"]
    assert "error" not in c3, c3
    assert "configs[3]" in c3["workload"] and c3["dtype"] == "f32" and c3["rccl_ranks"] == 1
    assert c3["ms_per_step"] * c3["steps_per_block"] >= 4.0 and 1e8 < c3["value"] < 5e9
    assert 0 < c3["with_tau_allgather"]["value"] <= c3["value"] * 1.05 and c3["status_ok_frac_rank0"] > 0.99
    c5 = d["scale_config5"]
    for k, n in (("per_gpu_1024", 1024), ("total_1024_over_8", 128)):
        assert "error" not in c5[k], c5[k]
        assert "configs[4]" in c5[k]["workload"] and abs(c5[k]["value"] - 20 * n / (c5[k]["ms_per_rollout"] * 1e-3)) <= 1e-6 * c5[k]["value"]
        assert 5.0 < c5[k]["us_per_tick"] < 200.0
        r = c5[k]["roofline"]
        assert r and r["bound"] == "valu_f64" and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        assert r["launches_timed"] >= 5 and r["us_per_tick"] <= c5[k]["us_per_tick"] * 1.02 and r["bytes"]["achieved"] > 0


def test_config5_line_carries_roofline_and_cpu_baseline():
    """`bench.py --config 5` (BASELINE.json configs[4]: horizon-20 rollouts of 1 024 robots): both measurement objects, priced on the
    rollout launch's own start / stop events and the oracle's instrumented operation count; the CPU leg is the oracle's rollout()."""
    d = one_line([sys.executable, "bench.py", "--config", "5", "--steps", "5", "--warmup", "2"])
    for k in KEYS:
        assert k in d, k
    assert "configs[4]" in d["config"]["workload"] and d["config"]["batch_per_gpu"] == 1024 and d["config"]["horizon"] == 20
    assert abs(d["value"] - 20 * 1024 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "valu_f64" and r["unit"] == "TFLOP/s" and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["launches_timed"] >= 10 and r["avg_launch_us"] * 1e-3 <= d["ms_per_step"] * 1.02       # the launch fits inside a step
    assert abs(r["achieved"] - r["flops_per_tick"] * 20 * 1024 / (r["avg_launch_us"] * 1e-6) / 1e12) <= 1e-6 * r["achieved"]
    assert r["bytes"]["bound"] == "hbm" and r["bytes"]["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["flops_per_tick"] > 1e4
    assert c["qp_iters_per_tick"] < c["qp_iters_per_tick_cold"]
    assert d["qp"]["status_ok_frac_last_tick"] == 1.0


def test_closed_loop_leg_reports_cold_and_warm_ticks_of_a_drifting_batch():
    """--closed-loop adds the dependent-tick leg BESIDE `value` (which stays the cold tick on standing inputs): same batch, inputs rewritten
    every tick, wbc_step_batch against wbc_step_batch_warm; both loops solve every QP, the warm one in fewer iterations."""
    d = one_line([sys.executable, "bench.py", "--steps", "50", "--warmup", "5", "--large-batch", "0", "--no-latency", "--no-cpu", "--closed-loop"])
    check_common(d, 50, 5)
    cl = d["closed_loop"]
    for tag in ("cold", "warm"):
        leg = cl[tag]
        assert leg["status_ok_frac"] > 0.999 and leg["us_per_tick"] > leg["drift_kernels_alone_us_per_tick"] * 0.5
        assert abs(leg["value"] - 4096 / (leg["us_per_tick"] * 1e-6)) <= 1e-6 * leg["value"]
        assert leg["plan"]["fused"] == 1 and "fused_us" in leg["tick_kernels_us"]
    assert cl["warm"]["plan"]["qp_warm"] == 1 and cl["cold"]["plan"]["qp_warm"] == 0
    assert cl["warm"]["qp_iters_mean"] < 0.3 * cl["cold"]["qp_iters_mean"]
    assert cl["warm"]["tick_kernels_us"]["fused_us"] < cl["cold"]["tick_kernels_us"]["fused_us"]

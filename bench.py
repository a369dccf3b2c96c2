#!/usr/bin/env python3
"""Benchmark of the WBC per-tick hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config {2,3,4}] [--no-cpu]

One "step" = one control tick (dynamics sweep -> observer -> GRF QP -> torque map) over one batch of
synthetic states already resident in HBM.  Default workload = BASELINE.json configs[1]:
batch 4096 DogBot-like states per GPU, 4-contact stance, observer off, fp64 (the real DogBot URDF is
absent; a synthetic quadruped of the same topology stands in -- see DESIGN.md).
For N > 1: one rank per GPU over RCCL; the batch shards with no data-path collective (weak scaling: per-GPU batch
fixed).  Under torch.distributed.run (the driver's form) this process is one rank.  Started bare
(`python bench.py --gpus N`, no WORLD_SIZE in the environment) it launches the N ranks itself as a child
`python -m torch.distributed.run ...` BEFORE touching the GPU, forwards rank 0's JSON line and exits with the
children's status (non-zero with a clear message when the box has fewer than N GPUs).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VALU_F64_PEAK_TFLOPS = 78.6  # fp64 vector FMA; tools/fma_probe.hip measures it on the box
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Algorithmic words per state of the kernel the roofline is quoted on:
#   fused dyn_sweep (WBC_SWEEP=fused): SURVEY.md 8(d) dynamics-sweep stage, in q19+v18(+mask) -> out M171+h18+Jc216 = 443
#   split default: mass_jac_kernel, in q19 -> out M171+Jc216 = 406 (h and the step workspace belong to rnea_step_kernel)
DYN_WORDS_FUSED = 443
DYN_WORDS_MASS_JAC = 406


def dyn_words(split):
    return DYN_WORDS_MASS_JAC if split else DYN_WORDS_FUSED


def dyn_kernel_name(split):
    return "mass_jac_kernel" if split else "dyn_sweep_kernel"


def tick_sweep_symbol(dtype, obs, n):
    """rocprofv3 name prefix of the front-half kernel that writes M, h, Jc in a two-kernel tick -- asked of the library's own planner (wbc_plan_tick), not
    restated: front 4 = sweep_obs_kernel (observer update + observer-free sweep as two roles of one launch), MODE 11 = MATS | STEP | NOB (observer off, or
    the observer as its own kernel in front), MODE 7 = MATS | STEP | OBS (all-in-one observer sweep)"""
    import wbc_quadruped_dob_amd as W
    sc = "double" if dtype == "f64" else "float"
    pl = W.plan_tick(n, dtype, obs)
    if pl["front"] == 4:
        return "sweep_obs_kernel<%s," % sc
    return "dyn_sweep_kernel<%s, %d," % (sc, 7 if (obs and pl["front"] == 0) else 11)


def tick_sweep_name(dtype, obs, n, split):
    """the name the roofline object carries for that kernel"""
    if split:
        return "mass_jac_kernel"
    return "sweep_obs_kernel (observer update + observer-free dynamics sweep as the two roles of one launch)" if tick_sweep_symbol(dtype, obs, n).startswith("sweep_obs") else "dyn_sweep_kernel"


print_line = lambda obj: print(json.dumps(obj))   # replaced in main() once descriptor 1 has been pointed at stderr


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks as a child process.  Nothing in THIS process
    initialises the GPU (device_count() does not, on this image) and nothing is exec'ed: the child is a subprocess and
    its exit status becomes ours."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this box: nothing measured\n" % (args.gpus, have))
        return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def timed_blocks(step, steps, dist, torch, min_total_s=0.05, max_blocks=400, min_blocks=7, device="cuda", sync=None):
    """Times blocks of EXACTLY `steps` ticks, each bracketed by barrier + synchronize on both sides, until at least
    `min_total_s` has been measured AND at least `min_blocks` blocks exist; returns the per-block seconds (max over ranks).
    The ranks leave the opening barrier together and do not talk to each other inside a block (no data-path collective), so
    the job's time for the block is the slowest rank's t1 - t0 with t1 read after that rank's synchronize: the MAX over
    ranks below.  The closing barrier follows the clock read -- its own latency (a one-element RCCL all-reduce plus a host
    wait, tens of microseconds) is not part of the K ticks and would be 10 % of a 20-tick block.
    A 20-tick block at the bench default lasts 0.5 ms -- one scheduler hiccup moves a single sample by > 5 %, the median of
    ~100 blocks does not.  (min_blocks: a one-off stall of ~70 ms was seen in about one in twelve fp32 runs, always in the first
    or second block; with one or two blocks it WAS the median.)
    device / sync: where the agreement tensors live and how a rank waits for its device ("cuda" / torch.cuda.synchronize; the gloo rehearsal of the N > 1
    path, tests/test_sharding_gloo.py, passes "cpu" and a no-op -- the SAME function times its stand-in ticks)."""
    sync = sync or torch.cuda.synchronize
    times = []
    total = 0.0
    while len(times) < max_blocks and (total < min_total_s or len(times) < min_blocks):
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        dt = time.perf_counter() - t0     # this rank's K ticks, finished on its device
        if dist is not None:
            dist.barrier()
        if dist is not None:   # every rank takes the same decision: the slowest rank's clock
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        times.append(dt)
        # a stalled block (host hiccup: > 3 x the fastest block so far) is kept as a sample but does not count towards the measured total -- round 5: one
        # 70 ms stall had ended a run after three blocks, whose median then sat 5 % off the twelve-block runs beside it (profiles/r05m_bench_cfg4_f32_n32768.json)
        total += dt if dt <= 3.0 * min(times) else 0.0
    return times


def long_blocks_of(step, k0, dist, torch, np, min_block_s=5e-3, device="cuda", sync=None):
    """timed_blocks with a block length that is raised until the median block lasts at least min_block_s (a first estimate of the
    step time taken right after set-up can be off by an order of magnitude: module load, clocks)"""
    k = max(1, int(k0))
    if dist is not None:   # k0 comes from each rank's own clock: every rank must run the SAME number of steps (callers follow the blocks
        from wbc_quadruped_dob_amd.sharding import agree_on_steps   # with per-step collectives; a different count per rank would hang them)
        k = agree_on_steps(k, dist, device)
    for _ in range(4):
        bl = timed_blocks(step, k, dist, torch, device=device, sync=sync)
        el = float(np.median(bl))
        if el >= min_block_s:
            break
        k = int(np.ceil(k * min_block_s / max(el, 1e-7) * 1.15))
    return k, bl, el


def measure_job(step, steps, n, world, dist, torch, np, device="cuda", sync=None, after_blocks=None):
    """The timed region of the contract and its N > 1 companion: blocks of EXACTLY `steps` ticks (median block = `elapsed`), and -- with a process group --
    the same ticks again in blocks of >= 5 ms (rank skew at the barriers < 1 % of a block), reported beside `value`.  after_blocks(): called between the two
    (main() closes its sampled-timing window there).  Returns (blocks, elapsed, long_blocks)."""
    blocks = timed_blocks(step, steps, dist, torch, device=device, sync=sync)
    if after_blocks is not None:
        after_blocks()
    elapsed = float(np.median(blocks))   # seconds per block of exactly `steps` ticks
    long_blocks = None
    if dist is not None:
        # N > 1: the driver's K may be 20 ticks = 0.4 ms per block, where one late rank moves the maximum by several per cent.
        k_long, lb, el_long = long_blocks_of(step, max(steps, int(np.ceil(5e-3 / max(elapsed / steps, 1e-7)))), dist, torch, np, device=device, sync=sync)
        long_blocks = {"value": k_long * n * world / el_long, "ms_per_step": el_long / k_long * 1e3, "steps_per_block": k_long,
                       "blocks": len(lb), "block_ms_median": el_long * 1e3,
                       "note": "blocks of >= 5 ms so that rank skew at the barriers is < 1 % of a block; `value` keeps the contract's K"}
    return blocks, elapsed, long_blocks


def contract_fields(steps, warmup, n, world, elapsed, dtype, workload, want_mats):
    """The driver's contract: `value` = units all ranks processed / the slowest rank's time for exactly `steps` ticks (median block)."""
    return {
        "metric": "WBC control-steps/sec (batched DogBot)",
        "value": steps * n * world / elapsed,
        "unit": "control-steps/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic",
        "config": {"workload": workload, "batch_per_gpu": n, "parallelism": "batch-sharded x%d, no data-path collective" % world,
                   "writes_M_h_Jc": want_mats},
    }


def job_fields(blocks, steps, elapsed, long_blocks, gather_res, world, dist):
    """what the line says about the timing itself and the N > 1 extras"""
    res = {"timing": {"blocks": len(blocks), "steps_per_block": steps, "block_ms_min": min(blocks) * 1e3,
                      "block_ms_median": elapsed * 1e3, "block_ms_max": max(blocks) * 1e3,
                      "note": "value and ms_per_step are the MEDIAN block of exactly `steps` ticks (barrier + synchronize on both "
                              "sides of every block, max over ranks); blocks repeat until >= 50 ms are measured"},
           "with_tau_allgather": gather_res,
           "rccl_ranks": world if dist is not None else None}
    if long_blocks is not None:
        res["value_long_blocks"] = long_blocks
    return res


def leg_fields(workload, k, units_per_step, world, el, bl, gather, dtype, unit_ms_key="ms_per_step"):
    """one extra leg of an N > 1 line (scale_legs): k steps per block, each step `units_per_step` control-steps on every one of `world` ranks, median block `el` s"""
    return {"workload": workload, "value": k * units_per_step * world / el, "unit": "control-steps/s", unit_ms_key: el / k * 1e3,
            ("steps_per_block" if unit_ms_key == "ms_per_step" else "rollouts_per_block"): k, "blocks": len(bl), "with_tau_allgather": gather,
            "rccl_ranks": world, "dtype": dtype}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000, help="timed ticks (2000 x ~25 us: long enough that rank skew at N>1 is noise)")
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=4096, help="states per GPU")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument("--horizon", type=int, default=20, help="config 5: ticks per rollout")
    ap.add_argument("--dtype", default=None, choices=["f64", "f32"])
    ap.add_argument("--tracking", action="store_true", help="config 5: CoM planner in the loop (reference kernel every tick)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU-oracle baseline leg")
    ap.add_argument("--no-mats", action="store_true", help="do not write M,h,Jc to HBM (fused-only variant)")
    ap.add_argument("--sample-every", type=int, default=47,
                    help="HIP-event instrumentation period inside the timed region (a sampled launch costs ~2 us more: every 10-th "
                         "took 2 %% off `value`; a prime, so that with short blocks the samples fall on every position of a block)")
    ap.add_argument("--large-batch", type=int, default=262144, help="extra roofline characterisation batch (0 = skip)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-state latency leg (p50 over 1000 ticks)")
    ap.add_argument("--closed-loop", action="store_true",
                    help="extra leg: dependent ticks of a DRIFTING batch, cold start against wbc_step_batch_warm, reported beside `value` (opt-in: its "
                         "cold ticks launch the same kernel symbols as the timed region and would mix into a rocprofv3 --stats summary of the command)")
    ap.add_argument("--closed-loop-only", action="store_true",
                    help="(internal) print ONLY the closed-loop leg of this config / batch -- the bench's own batch and the same batch with +-40 N "
                         "lateral commands -- as one JSON line: what the default run starts as a CHILD process")
    ap.add_argument("--no-closed-loop", action="store_true", help="default run: do not start the closed-loop child")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 without torchrun: ONE process drives N devices through the C-ABI's wbc_multi_* path")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.single_process:
        sys.exit(self_launch(args))
    if args.single_process:
        # several shards on ONE device (what a 1-GPU box can run of this path) share the device's hardware queues -- four by default: two shard streams
        # that land on the same queue run their ticks one after the other (156 against 296 M steps/s for the same two 2 048-state ticks, round 5).  One
        # queue per stream for this mode; with one shard per device (the real use) it changes nothing.  Must be set before the runtime comes up.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    # ONE line on stdout, whatever the libraries underneath print: RCCL writes its version banner (and, with NCCL_DEBUG set on
    # the box, more) to file descriptor 1 when the process group comes up.  From here on descriptor 1 IS stderr; the JSON
    # line goes to the saved original through emit().
    sys.stdout.flush()
    _real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(_real_stdout, (json.dumps(obj) + "\n").encode())
    global print_line
    print_line = emit

    import numpy as np
    import torch
    import wbc_quadruped_dob_amd as W
    from wbc_quadruped_dob_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("WBC_BENCH_FORCE_DIST"):   # (the variable: a one-rank job that still takes the N > 1 path, for the tests)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.single_process:
        return multi_capi_bench(args, W, synth, torch, np)
    if args.closed_loop_only:
        return closed_loop_only(args, W, synth, torch, np)
    if world != args.gpus:
        sys.stderr.write("bench.py: WORLD_SIZE=%d but --gpus %d\n" % (world, args.gpus))
        sys.exit(2)
    torch.cuda.set_device(local_rank)

    if args.config == 5:
        return rollout_bench(args, W, synth, torch, np, dist, world, rank, local_rank)
    dtype = args.dtype or ("f32" if args.config == 4 else "f64")
    obs = 0 if args.config == 2 else 1
    td = torch.float64 if dtype == "f64" else torch.float32
    n = args.batch
    model = load_model(W, torch, dist, rank, local_rank)
    P = synth.default_params(observer_order=obs, dtype=dtype)
    solver = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=local_rank, max_batch=n)
    B = synth.make_batch(args.config, n, model.total_mass, rank=rank)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
    inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
    mask = torch.from_numpy(B["mask"]).cuda()
    integ = rr = None
    if obs:
        integ = solver.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone()
        rr = torch.zeros_like(integ)
    out = {}
    want_mats = not args.no_mats
    split = os.environ.get("WBC_SWEEP", "fused") == "split"

    # argument structs validated and built once (Solver.prepare_step): a tick is then one C call from Python
    tick, out0 = solver.prepare_step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask,
                                     inp["tau_prev"], inp["f_prev"], integ, rr, out=out, want_mats=want_mats)
    out.update(out0)

    def step():
        tick()
        return out

    step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # HIP events bracket both kernels of every SAMPLE-th tick of the timed region (on the launch stream): recording
    # events around every launch costs ~15 us per tick at this batch size, which would be a quarter of the step.
    sample = max(1, args.sample_every)
    solver.enable_timing(sample)
    tmbox = {}

    def close_sampling():
        tmbox["tm"] = solver.collect_timing()
        solver.enable_timing(0)
    # (tests/test_sharding_gloo.py drives measure_job / gather_leg / contract_fields / job_fields with eight gloo ranks and stand-in ticks)
    blocks, elapsed, long_blocks = measure_job(step, args.steps, n, world, dist, torch, np, after_blocks=close_sampling)
    tm = tmbox["tm"]
    status = out["status"].cpu().numpy()
    iters = out["iters"].cpu().numpy()
    # the optional consumer-side collective (every rank receives all torques), reported BESIDE `value`
    gather_res = None
    if dist is not None and not os.environ.get("WBC_BENCH_NO_GATHER"):
        try:
            nbytes = 12 * n * (8 if dtype == "f64" else 4)

            def make_tick(tau_view):   # the same tick, its torques written straight into this rank's block of a gather buffer
                o = dict(out)
                o["tau"] = tau_view
                tk, o2 = solver.prepare_step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask,
                                             inp["tau_prev"], inp["f_prev"], integ, rr, out=o, want_mats=want_mats)

                def st():
                    tk()
                    return o2
                return st
            gather_res = gather_leg(make_tick, dist, args.steps, n, world, rank, nbytes, td, torch)
        except Exception as e:  # never lose the main line to the optional leg
            gather_res = {"error": repr(e)[:200]}

    # cost of an event pair with nothing between them, on the same stream (reported for reference: the per-kernel times
    # are the dispatches' own start/stop events unless WBC_TIMING=pair, see docs/DESIGN_R04.md section 6)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    for e0, e1 in ev:
        step()
        e0.record()
        e1.record()
    torch.cuda.synchronize()
    ev_overhead_us = float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])) * 1e3

    if rank == 0:
        ts = 8 if dtype == "f64" else 4
        fused = tm.get("fused_launches", 0) > 0   # the whole tick as ONE kernel: fused_tick_kernel (small batches) or tile_tick_kernel (wbc_tick_plan.fused = 2)
        tile_tick = fused and solver.plan_tick(n, want_mats=want_mats)["fused"] == 2
        pair_tick = fused and solver.plan_tick(n, want_mats=want_mats)["fused"] == 3   # fused_pair_kernel: the one-launch tick as 32-state workgroups (ABI 9)
        dyn_s = (tm["fused_ms"] * 1e-3 / tm["fused_launches"]) if fused else tm["dyn_ms"] * 1e-3 / max(1, tm["dyn_launches"])
        qp_s = tm["qp_ms"] * 1e-3 / max(1, tm["qp_launches"])
        qpl_s = tm.get("qp_lane_ms", 0.0) * 1e-3 / max(1, tm.get("qp_lane_launches", 0))
        rnea_s = tm["rnea_ms"] * 1e-3 / max(1, tm["rnea_launches"])
        # algorithmic words per state of the timed launch: the dynamics stage alone is 443 (in q 19 + v 18,
        # out M 171 + h 18 + Jc 216, + 1); the fused tick launch does the WHOLE tick: in 78 (q, v, w_des, vdot_des, normals,
        # mu, mask) + out 24 (tau, f) + M/h/Jc 405 = 507; observer on adds 60 in (state, tau_prev, f_prev) + 36 out = 603
        words = (507 + (96 if obs else 0) - (0 if want_mats else 405)) if fused else dyn_words(split)
        dyn_bytes = words * ts * n
        # The span the roofline is priced on must fit inside a step.  The sampled launches carry their own start / stop events
        # (hipExtLaunchKernelGGL), which makes THEM ~2 us slower than the un-sampled launches that `value` is made of (round 2:
        # 22.1 us sampled vs 21.05 us per step vs 19.96 us by rocprofv3).  Back-to-back launches of one kernel per step on one
        # stream have period = kernel duration + dispatch gap >= kernel duration, so for the one-launch tick the un-perturbed
        # bound is the step period itself; for a multi-kernel tick the sampled spans are scaled down when their sum exceeds it.
        period_s = elapsed / args.steps
        event_span_s = dyn_s
        spans_sum = dyn_s + (0.0 if fused else qp_s + qpl_s + rnea_s)
        span_scale = min(1.0, period_s / spans_sum) if spans_sum > 0 else 1.0
        dyn_s = dyn_s * span_scale
        achieved = dyn_bytes / dyn_s / 1e9 if ((want_mats or fused) and dyn_s > 0) else None
        # SURVEY.md 8(d) whole-path algorithmic bytes: in 78 words + out 24 (observer: + 60 in, + 36 out)
        path_words = 102 + (96 if obs else 0)
        path_gbs = path_words * ts * (args.steps * n / elapsed) / 1e9
        devinfo = device_probe(torch) if world == 1 else None
        # rows of the priced launch: whole tick = in 78 (+ 60 with the observer) / out 24 + 405 (+ 36); the dynamics stage alone = in 38, out 405
        rows_in = (78 + (60 if obs else 0)) if fused else 38
        rows_out = words - rows_in
        pattern = pattern_ceiling(ts, rows_in, rows_out, n) if world == 1 else None
        res = contract_fields(args.steps, args.warmup, n, world, elapsed, dtype,
                              "configs[%d]: batch=%d states/GPU, %s, observer %s, %s; synthetic quadruped URDF (DogBot URDF absent)"
                              % (args.config - 1, n, "4-contact stance" if args.config == 2 else "mixed 2/3/4-foot trot masks", "on" if obs else "off", dtype), want_mats)
        res.update({
            "roofline": {"kernel": (("tile_tick_kernel (sweep + observer roles of a 64 / 96 / 128-state workgroup, then the staged QP tile of the same states: one launch)" if tile_tick
                                     else ("fused_pair_kernel (two 16-state tick workgroups as one twelve-wavefront workgroup of 32 states)" if pair_tick else "fused_tick_kernel")) if fused else tick_sweep_name(dtype, obs, n, split)),
                         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic": pmc_traffic((("tile_tick" if tile_tick else "fused_tick") if fused else tick_sweep_symbol(dtype, obs, n)), n, dtype),
                         "traffic_source": PMC_SOURCE,
                         "algorithmic_words_per_state": words,
                         "algorithmic_bytes_per_launch": dyn_bytes, "avg_launch_us": dyn_s * 1e6,
                         "event_span_us": event_span_s * 1e6, "step_period_us": period_s * 1e6,
                         "avg_launch_source": ("sampled dispatch spans (start/stop events of the dispatch) scaled by %.3f so that the "
                                               "tick's kernels fit inside the measured step period: a sampled launch is slower than "
                                               "the un-sampled ones `value` is made of" % span_scale),
                         # the yardstick: tools/bw_probe.bin moving EXACTLY this launch's rows (reads and writes, leg-major lanes, no arithmetic) on this device, now
                         "pattern_ceiling": pattern,
                         "frac_of_pattern_ceiling": (achieved / pattern["gbs"]) if (achieved and pattern and pattern.get("gbs")) else None,
                         "launches_timed": tm["fused_launches"] if fused else tm["dyn_launches"],
                         "event_pair_overhead_us": ev_overhead_us,
                         "note": ("HIP start/stop events of the dispatch itself (hipExtLaunchKernelGGL) on the launch stream, every "
                                  "%d-th tick of the timed region" % sample) +
                                 ("; the whole tick is ONE launch (a workgroup per CU runs the dynamics roles, then the staged QP tile of its own states), so `achieved` = the "
                                  "algorithmic bytes of the whole tick (inputs incl. observer state + tau, f + M, h, Jc + new observer state) over that launch" if tile_tick else "") +
                                 ("; at this batch the whole tick is ONE launch (dynamics + GRF QP as wavefront roles of a "
                                  "workgroup, latency-bound: one workgroup per CU), so `achieved` = the algorithmic bytes of the whole tick "
                                  "(inputs + tau, f + M, h, Jc) over that launch -- see roofline_dyn_sweep_alone for the 443-word "
                                  "sweep kernel by itself at this batch and roofline_large_batch for the HBM-bound regime" if (fused and not tile_tick) else "")},
            "kernels": {"dyn_sweep_us": None if fused else dyn_s * 1e6, "fused_tick_us": dyn_s * 1e6 if fused else None,
                        "rnea_step_us": rnea_s * 1e6 if tm["rnea_launches"] else None,
                        "qp_us": None if fused else qp_s * 1e6,
                        "qp_lane_us": (qpl_s * 1e6) if tm.get("qp_lane_launches", 0) else None,
                        "qp_us_per_state_amortized": None if fused else qp_s * 1e6 / n,
                        "qp_kernel": os.environ.get("WBC_QP_KERNEL", "group16"),
                        "sweep": ("one launch: rnea_step | mass_jac | qp_group16 as wavefront roles (fused_tick_kernel)" if fused
                                  else ("split: mass_jac on a 2nd stream || rnea_step -> qp" if split else "dyn_sweep -> qp") if want_mats
                                  else "rnea_step (no CRBA, no M/h/Jc) -> qp")},
            "qp": {"status_ok_frac": float((status == 0).mean()), "iters_mean": float(iters.mean()),
                   "iters_max": int(iters.max())},
            "roofline_whole_path_bytes": {"bound": "hbm", "achieved": path_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": path_gbs / HBM_PEAK_GBS / world, "bytes_per_step": path_words * ts,
                                          "note": "SURVEY.md 8(d) whole-path algorithmic bytes per control step (inputs + tau, f%s) x "
                                                  "control-steps/s, per GPU against 8 TB/s: the path is latency / issue-bound long before "
                                                  "it is bound by these bytes" % (" + observer state" if obs else "")},
            "device": devinfo,
        })
        res.update(job_fields(blocks, args.steps, elapsed, long_blocks, gather_res, world, dist))
        if not args.no_latency and world == 1:
            res["qp_latency"] = qp_latency(W, synth, torch, np, model, B, P, dtype, td, obs)
        if not args.no_latency and world == 1:
            res["qp_dense_general"] = qp_dense_general(W, torch, dtype, with_cpu=not args.no_cpu)
        if fused and world == 1:
            res["roofline_dyn_sweep_alone"] = sweep_alone_roofline(solver, torch, inp, n, dtype, ts)
        if args.large_batch and world == 1:
            res["roofline_large_batch"] = large_batch_roofline(W, synth, torch, np, model, args, dtype, td, obs, split)
        if args.large_batch and world == 1 and not args.no_latency and args.config == 2:   # (the full default line only)
            res["baseline_configs"] = baseline_configs_leg(W, synth, torch, np, model, args)
        lb_ = res.get("roofline_large_batch")
        if lb_ and isinstance(res.get("device"), dict) and dtype == "f64" and not obs and args.large_batch == 262144:
            # the pool's devices fall into two classes on the HBM-bound tick at identical clocks (docs/DESIGN_R04.md 6.0: sweep 182-187 us on
            # four of five devices probed, 208-213 us on the fifth): which kind this run drew, by the sweep's own time
            res["device"]["pool_class"] = "fast" if lb_["avg_launch_us"] <= 196.0 else "slow"
            res["device"]["pool_class_basis"] = "dyn_sweep<double, 11> at 262 144 states: %.1f us (<= 196 us = fast)" % lb_["avg_launch_us"]
        if args.closed_loop and world == 1:
            try:
                res["closed_loop"] = closed_loop_leg(W, torch, np, model, P, B, dtype, td, obs, n, local_rank, want_mats)
            except Exception as e:  # never lose the main line to the optional leg
                res["closed_loop"] = {"error": repr(e)[:200]}
        elif world == 1 and not args.no_closed_loop and not args.no_latency:
            # The full line carries the closed loop of its own batch (`value` is the cold tick on STANDING inputs, the friendliest case
            # of its kernel -- the same batch ticked as a control loop, and with harder commands, belongs beside it).  It runs in a CHILD process
            # started without the profiler's preload: its cold ticks launch the same kernel symbols as the timed region and would otherwise mix
            # into a rocprofv3 --stats summary of this command.
            res["closed_loop"] = closed_loop_child(args)
        if not args.no_cpu and world == 1:
            res["cpu_baseline"] = cpu_baseline(B, P, dtype, n)
            fl = res["cpu_baseline"].get("flops_per_step")
            if fl:
                # whole path (sweep + QP) against the fp64 vector-FMA peak (the fused path sits above the
                # fp64 ridge, so this -- not HBM -- is its ceiling); 78.6 TFLOP/s = 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz
                tf = fl * res["value"] / 1e12
                res["roofline_whole_path"] = {"bound": "valu_f64", "achieved": tf, "peak": VALU_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                              "frac": tf / VALU_F64_PEAK_TFLOPS, "flops_per_step": fl,
                                              "note": "flops = instrumented oracle op count x control-steps/s at the bench batch"}
                lb = res.get("roofline_large_batch")
                if lb:
                    tfl = fl * lb["steps_per_s"] / 1e12
                    lb["whole_path_tflops"] = tfl
                    lb["whole_path_frac_of_valu_f64_peak"] = tfl / VALU_F64_PEAK_TFLOPS
    # N > 1: BASELINE.json's 8-GPU configs as extra legs of the SAME line (the driver runs one command per N)
    legs = None
    if dist is not None and not os.environ.get("WBC_BENCH_NO_SCALE_LEGS"):
        torch.cuda.empty_cache()
        legs = scale_legs(args, W, synth, torch, np, dist, world, rank, local_rank, model)
    if rank == 0:
        if legs:
            res.update(legs)
        print_line(res)
    if dist is not None:
        dist.destroy_process_group()


def load_model(W, torch, dist, rank, local_rank):
    """The robot model of this rank.  N > 1: rank 0 reads the description, the flat arrays (< 4 kB) are broadcast once and
    every other rank builds its model from them (wbc_model_from_flat)."""
    if dist is None:
        return W.Model.from_urdf(W.SYNTHETIC_URDF)
    from wbc_quadruped_dob_amd.sharding import broadcast_model
    flat = W.Model.from_urdf(W.SYNTHETIC_URDF).flat() if rank == 0 else None
    return W.Model.from_flat(broadcast_model(flat, dist, device=torch.device("cuda", local_rank)))


def closed_loop_child(args):
    import subprocess
    env = {k: v for k, v in os.environ.items() if not (k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE") or k.startswith("ROCP") or k.startswith("ROCPROF"))}
    cmd = [sys.executable, os.path.abspath(__file__), "--closed-loop-only", "--config", str(args.config), "--batch", str(args.batch)]
    if args.dtype:
        cmd += ["--dtype", args.dtype]
    if args.no_mats:
        cmd += ["--no-mats"]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"error": "child exited %d: %s" % (p.returncode, p.stderr[-200:])}
        return json.loads(lines[-1])
    except Exception as e:  # never lose the main line to this leg
        return {"error": repr(e)[:200]}


def closed_loop_only(args, W, synth, torch, np):
    dtype = args.dtype or ("f32" if args.config == 4 else "f64")
    obs = 0 if args.config == 2 else 1
    td = torch.float64 if dtype == "f64" else torch.float32
    n = args.batch
    model = W.Model.from_urdf(W.SYNTHETIC_URDF)
    P = synth.default_params(observer_order=obs, dtype=dtype)
    B = synth.make_batch(args.config, n, model.total_mass, rank=0)
    res = closed_loop_leg(W, torch, np, model, P, B, dtype, td, obs, n, 0, not args.no_mats)
    Bh = {k: v.copy() for k, v in B.items()}
    Bh["w_des"][:, 0:2] += np.random.default_rng(1).uniform(-40, 40, (n, 2))   # (tools/warm_loop.py's batch: the same mean iteration count, harder worst cases)
    hard = closed_loop_leg(W, torch, np, model, P, Bh, dtype, td, obs, n, 0, not args.no_mats)
    res["hard_commands"] = {k: hard[k] for k in ("cold", "warm", "speedup_wall")}
    res["hard_commands"]["note"] = "the same loop with +-40 N lateral commands added to w_des: the tick lasts as long as its hardest QP"
    res["process"] = "child of the bench run (no profiler preload)"
    print_line(res)


def closed_loop_leg(W, torch, np, model, P, B, dtype, td, obs, n, device, want_mats, ticks=200):
    """Dependent ticks: the joint angles and the commanded wrench of every state move between ticks (two elementwise kernels), so the tick's
    inputs are never those of the previous tick.  The same loop twice -- wbc_step_batch (cold start) and wbc_step_batch_warm (every QP starts
    from the active set the previous tick ended on, carried in one int32 buffer).  `value` of the main line stays the cold tick on standing
    inputs; this leg says what a control loop over the same batch gets."""
    import time
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
    gen = torch.Generator(device="cuda").manual_seed(3)
    dq = 1e-3 * torch.randn((12, n), dtype=td, device="cuda", generator=gen)
    dw = 0.2 * torch.randn((6, n), dtype=td, device="cuda", generator=gen)
    dqn, dwn = -dq, -dw
    res = {}
    for tag, warm in (("cold", False), ("warm", True)):
        solver = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=device, max_batch=n)
        inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
        mask = torch.from_numpy(B["mask"]).cuda()
        integ = rr = None
        if obs:
            integ = solver.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone()
            rr = torch.zeros_like(integ)
        tick, out = solver.prepare_step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"],
                                        inp["f_prev"], integ, rr, want_mats=want_mats, warm=warm)
        qj = inp["q"][7:]

        def loop(k, with_tick=True):
            for i in range(k):
                qj.add_(dq if i % 2 == 0 else dqn)
                inp["w_des"].add_(dw if i % 2 == 0 else dwn)
                if with_tick:
                    tick()
        loop(20)
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            loop(ticks)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        t0 = time.perf_counter()
        loop(ticks, with_tick=False)
        torch.cuda.synchronize()
        drift = time.perf_counter() - t0
        solver.enable_timing(7)
        loop(210)
        torch.cuda.synchronize()
        tm = solver.collect_timing()
        solver.enable_timing(0)
        kern = {k[:-3] + "_us": v * 1e3 / max(1, tm[k[:-3] + "_launches"]) for k, v in tm.items() if k.endswith("_ms") and v > 0}
        res[tag] = {"us_per_tick": best / ticks * 1e6, "value": n * ticks / best, "drift_kernels_alone_us_per_tick": drift / ticks * 1e6,
                    "tick_kernels_us": kern, "kernels_sum_us": sum(kern.values()), "qp_iters_mean": float(out["iters"].double().mean()),
                    "status_ok_frac": float((out["status"] == 0).double().mean()), "plan": solver.plan_tick(n, want_mats=want_mats, warm=warm)}
    res["speedup_wall"] = res["cold"]["us_per_tick"] / res["warm"]["us_per_tick"]
    res["unit"] = "control-steps/s"
    res["note"] = ("best of 3 blocks of %d dependent ticks, wall time incl. the two drift kernels per tick (timed alone beside it); tick_kernels_us: the "
                   "tick's own kernels by their dispatch events, every 7th tick; plan.qp_warm = 0: at this size the cold tiles are the faster QP kernels "
                   "and the warm tick only carries the sets (DESIGN.md 4.4)" % ticks)
    return res


def gather_leg(make_tick, dist, steps, n, world, rank, nbytes, td, torch, device="cuda", sync=None):
    """SURVEY.md 8e "steps/s with and without the all-gather": the consumer-side collective of every tick.  The ticks write their
    torques straight into this rank's block of the gather buffer (in-place all-gather: no second copy of the local block), two
    buffers alternate.  `value` = K ticks + K gathers captured ONCE as a hipGraph and replayed (what the device needs; a tick of
    4 096 states is shorter than the Python call of one collective) -- the better of "gather behind the tick on its stream" and
    "gather on a side stream beside the next tick"; the same graph without the collectives and the eager (host-issued) forms are
    reported beside it."""
    from wbc_quadruped_dob_amd.sharding import (agree_on_steps, gather_buffers, graph_steps_with_gather, timed_steps_with_gather,
                                                timed_steps_with_overlapped_gather)
    get = lambda o: o["tau"]
    sync = sync or torch.cuda.synchronize
    flats, views = gather_buffers(world, rank, 12, n, td, device)
    step_fns = tuple(make_tick(v) for v in views)
    el_serial, _ = timed_steps_with_gather(step_fns[0], get, dist, steps, sync, flat=flats[0])
    res = {"collective": "all_gather_into_tensor(tau) of every step, RCCL, %d B per rank per step, in place (the tick writes tau into its block of "
                         "the gather buffer); two buffers alternate" % nbytes,
           "eager_serial": {"value": steps * n * world / el_serial, "ms_per_step": el_serial / steps * 1e3,
                            "note": "host-issued, the gather on the tick's stream behind every tick"}}
    try:
        el_o, _ = timed_steps_with_overlapped_gather(step_fns, get, dist, steps, sync, flats=flats)
        res["eager_overlapped"] = {"value": steps * n * world / el_o, "ms_per_step": el_o / steps * 1e3,
                                   "note": "host-issued, side stream beside the next tick: bound by the Python cost of the collective calls"}
    except Exception as e:
        res["eager_overlapped"] = {"error": repr(e)[:200]}
    # The hipGraph forms are OPT-IN (WBC_BENCH_GRAPH_GATHER=1): capturing RCCL collectives beside the process group's watchdog thread
    # stalled one run in three on this stack for the 10 minutes of the group's time-out (and aborted it in the default capture mode) --
    # not something the driver's scaling command should risk.  profiles/r04*_bench_scale_legs_1rank_graph.json is such an opt-in run.
    graph_forms = (("graph_ticks_only", False, False), ("graph_serial", False, True), ("graph_overlapped", True, True)) \
        if os.environ.get("WBC_BENCH_GRAPH_GATHER") == "1" else ()
    # graphs long enough (>= 5 ms) that the replay's own launch cost is < 1 % of what it times
    kg = max(steps, int(5e-3 / max(el_serial / steps, 1e-6)) + 1)
    kg = agree_on_steps(kg, dist, device) if graph_forms else kg
    for key, ov, ga in graph_forms:
        try:
            el, _ = graph_steps_with_gather(step_fns, get, dist, kg, replays=5, overlapped=ov, gather=ga, flats=flats)
            res[key] = {"value": kg * n * world / el, "ms_per_step": el / kg * 1e3,
                        "note": "%d ticks%s captured once as a hipGraph, replayed 5 times" % (kg, " + %d gathers" % kg if ga else " (no gather: the baseline)")}
        except Exception as e:
            res[key] = {"error": repr(e)[:300]}
    cands = [k for k in ("graph_serial", "graph_overlapped") if "value" in res.get(k, {})]
    if cands:
        bk = max(cands, key=lambda k: res[k]["value"])
        res["value"], res["ms_per_step"], res["value_is"] = res[bk]["value"], res[bk]["ms_per_step"], bk
        if "value" in res.get("graph_ticks_only", {}):
            res["frac_of_graph_ticks_only"] = res["value"] / res["graph_ticks_only"]["value"]
    else:
        ek = max([k for k in ("eager_serial", "eager_overlapped") if "value" in res.get(k, {})], key=lambda k: res[k]["value"])
        res["value"], res["ms_per_step"], res["value_is"] = res[ek]["value"], res[ek]["ms_per_step"], ek
        res["note"] = ("host-issued ticks and collectives: at this tick length the Python call of a collective costs more than the tick; the device-side "
                       "cost (the same sequence captured as a hipGraph) is measured with WBC_BENCH_GRAPH_GATHER=1")
    return res


def pattern_ceiling(scalar_bytes, rows_in, rows_out, n):
    """What this device's memory system delivers on the launch's own footprint: tools/bw_probe.bin (built by __graft_entry__.build()) run as a CHILD process --
    `rows_in` rows read and `rows_out` rows written per state, component-major, leg-major lanes, 8 bytes per lane and row, no arithmetic.  None when the probe
    is not built.  (the yardstick used to be torch's 1 GiB copy, 4.5 TB/s -- below what the sweep's own pattern reaches.)"""
    import subprocess
    exe = os.path.join(ROOT, "tools", "bw_probe.bin")
    if not os.path.exists(exe):
        return None
    try:
        run = subprocess.run([exe, "rw", str(scalar_bytes), str(rows_in), str(rows_out), str(n)], capture_output=True, text=True, timeout=120)
        gbs, us = run.stdout.split()[:2]
        return {"gbs": float(gbs), "us": float(us), "rows_in": rows_in, "rows_out": rows_out, "states": n,
                "source": "tools/bw_probe.bin rw %d %d %d %d (child process, after the timed region): the better of 64- and 256-thread workgroups moving these rows 8 bytes per lane; a "
                          "kernel whose own store order does better than the probe's (the sweep writes its 162 structural words as 16-byte stores) can exceed it" % (scalar_bytes, rows_in, rows_out, n)}
    except Exception as e:
        return {"error": repr(e)[:200]}


def device_probe(torch):
    """What this device's memory system delivers on a plain copy (the pool's devices differ by up to 12 % on the HBM-bound
    tick at identical clocks, docs/DESIGN_R04.md 6.0): 1 GiB device-to-device copy, read + write bytes over the median of 10 copies."""
    n = 1 << 28
    try:
        a = torch.empty(n, dtype=torch.float32, device="cuda")
        b = torch.empty(n, dtype=torch.float32, device="cuda")
        a.fill_(1.0)
        for _ in range(3):
            b.copy_(a)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for e0, e1 in ev:
            e0.record(); b.copy_(a); e1.record()
        torch.cuda.synchronize()
        ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)[len(ev) // 2]
        gbs = 2.0 * 4 * n / (ms * 1e-3) / 1e9
        del a, b
        torch.cuda.empty_cache()
        return {"name": torch.cuda.get_device_name(), "hbm_copy_gbs": gbs, "hbm_copy_frac_of_peak": gbs / HBM_PEAK_GBS,
                "note": "torch 1 GiB device-to-device copy, read + write bytes, median of 10 (what 'achievable' means on this device)"}
    except Exception as e:   # never lose the main line to the probe
        return {"error": repr(e)[:200]}


def scale_legs(args, W, synth, torch, np, dist, world, rank, local_rank, model):
    """BASELINE.json configs[3] (batch 262 144 fp32 sharded over 8 GPUs = 32 768 states per GPU, tilted terrain, observer on)
    and configs[4] (horizon-20 rollouts, 1 024 per GPU and 1 024 in total = 128 per GPU) measured by the driver's own N > 1
    command: per leg steps/s over all ranks in blocks of >= 5 ms, the same with an all-gather of tau after every tick
    (configs[3]), and the RCCL rank count.  Every rank keeps its slice for all ticks: no data-path collective."""
    from wbc_quadruped_dob_amd.sharding import timed_steps_with_gather, timed_steps_with_overlapped_gather
    res = {}
    td = torch.float32
    n3 = 32768
    try:
        P = synth.default_params(observer_order=1, dtype="f32")
        solver = W.Solver(model, W.Params.from_dict(P, "f32"), dtype="f32", device=local_rank, max_batch=n3)
        B = synth.make_batch(4, n3, model.total_mass, rank=rank)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
        inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
        mask = torch.from_numpy(B["mask"]).cuda()
        integ = solver.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone()
        rr = torch.zeros_like(integ)
        tick, out = solver.prepare_step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask,
                                        inp["tau_prev"], inp["f_prev"], integ, rr, want_mats=True)

        def step():
            tick()
            return out
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        k3, bl, el = long_blocks_of(step, max(20, int(np.ceil(5e-3 / max((time.perf_counter() - t0) / 20, 1e-7)))), dist, torch, np)
        def make_tick(tau_view):
            o = dict(out)
            o["tau"] = tau_view
            tk, o2 = solver.prepare_step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask,
                                         inp["tau_prev"], inp["f_prev"], integ, rr, out=o, want_mats=True)

            def st():
                tk()
                return o2
            return st
        g3 = gather_leg(make_tick, dist, k3, n3, world, rank, 12 * n3 * 4, td, torch)
        ok = float((out["status"] == 0).double().mean().item())
        res["scale_config3"] = leg_fields("configs[3]: batch=%d fp32 = %d states per GPU x %d, tilted terrain normals + disturbances, observer on, M/h/Jc written"
                                          % (n3 * world, n3, world), k3, n3, world, el, bl, g3, "f32")
        res["scale_config3"]["status_ok_frac_rank0"] = ok
        del solver, tick, out, inp, integ, rr, make_tick
        torch.cuda.empty_cache()
    except Exception as e:   # never lose the headline to an extra leg
        res["scale_config3"] = {"error": repr(e)[:300]}
    for name, n5 in (("per_gpu_1024", 1024), ("total_1024_over_8", 128)):
        try:
            r5 = rollout_setup(args, W, synth, torch, np, model, "f64", n5, args.horizon, rank, local_rank, False)
            for _ in range(3):
                r5["one_rollout"]()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                r5["one_rollout"]()
            torch.cuda.synchronize()
            k5, bl, el = long_blocks_of(r5["one_rollout"], max(3, int(np.ceil(5e-3 / max((time.perf_counter() - t0) / 3, 1e-7)))), dist, torch, np)
            rf5 = rollout_roofline(r5["solver"], r5, torch, np, n5, args.horizon, "f64", reps=10)
            cpu5 = None
            if rank == 0:
                try:   # the timed CPU leg only on a one-rank job (the contract's "rank 0 at N = 1"); the operation count always
                    cpu5 = (cpu_baseline_rollout(r5["B"], r5["P"], n5, args.horizon, r5["tau_ext_np"]) if (world == 1 and not args.no_cpu)
                            else {"flops_per_tick": rollout_flops_only(r5, args.horizon)})
                except Exception as e:
                    cpu5 = None
            leg = leg_fields("configs[4]: horizon=%d x %d rollouts per GPU x %d GPUs, trot masks, observer on, pushes, fp64, rank-local for all ticks"
                             % (args.horizon, n5, world), k5, args.horizon * n5, world, el, bl, None, "f64", unit_ms_key="ms_per_rollout")
            leg.update({"roofline": rollout_measurement_objects(rf5, cpu5, None, n5, args.horizon, "f64", world) if cpu5 else None,
                        "cpu_baseline": cpu5 if (cpu5 and "value" in cpu5) else None,
                        "us_per_tick": el / k5 / args.horizon * 1e6,
                        "note": "no gather leg: a rollout keeps tau of 20 ticks on its rank; the consumer-side collective is configs[3]'s"})
            res.setdefault("scale_config5", {})[name] = leg
            del r5
            torch.cuda.empty_cache()
        except Exception as e:
            res.setdefault("scale_config5", {})[name] = {"error": repr(e)[:300]}
    return res


def multi_capi_bench(args, W, synth, torch, np):
    """--single-process: ONE host process, --gpus shards through the C-ABI's wbc_multi_* (what a C++ host does; shards are
    dealt round-robin over the visible devices, so on a 1-GPU box this is a functional run on device 0).  Same workload,
    timing brackets and JSON keys as the torchrun path; `value` excludes the gather, `with_tau_allgather` includes it."""
    dtype = args.dtype or ("f32" if args.config == 4 else "f64")
    obs = 0 if args.config == 2 else 1
    td = torch.float64 if dtype == "f64" else torch.float32
    k = args.gpus
    ndev = torch.cuda.device_count()
    devices = [i % ndev for i in range(k)]
    distinct = ndev >= k
    n_total = args.batch * k
    model = W.Model.from_urdf(W.SYNTHETIC_URDF)
    P = synth.default_params(observer_order=obs, dtype=dtype)
    ms = W.MultiSolver(model, W.Params.from_dict(P, dtype), dtype=dtype, devices=devices, max_batch_total=n_total,
                       gather="rccl" if distinct else "peer")
    shards = [synth.make_batch(args.config, args.batch, model.total_mass, rank=r) for r in range(k)]
    put = lambda a, d: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).to(torch.device("cuda", d))
    ins = {name: [put(B[name], d) for B, d in zip(shards, devices)] for name in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
    ins["mask"] = [torch.from_numpy(B["mask"]).to(torch.device("cuda", d)) for B, d in zip(shards, devices)]
    obs_state = None
    if obs:
        integ = []
        for i, d in enumerate(devices):
            sv = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=d, max_batch=args.batch)
            integ.append(sv.dynamics(ins["q"][i], ins["v"][i], want=("p",))["p"].clone())
        obs_state = (integ, [torch.zeros_like(x) for x in integ])
    tick, outs = ms.prepare_step(n_total, ins, obs_state, want_mats=not args.no_mats)
    tick_b, outs_b = ms.prepare_step(n_total, ins, obs_state, want_mats=not args.no_mats)   # second tau buffer for the overlapped gather
    tau_all = ms.allgather_tau(n_total, outs)
    tau_all_b = ms.allgather_tau(n_total, outs_b)
    tick_gather = ms.prepare_tick_gather(n_total, [tick.capi, tick_b.capi], [tau_all, tau_all_b])
    ms.synchronize()

    def run(mode, ms=ms, tick=tick, tick_gather=tick_gather):
        times, total, host = [], 0.0, []
        while total < 0.05 or not times:
            ms.synchronize()
            ms.host_stats(reset=True)
            t0 = time.perf_counter()
            for k in range(args.steps):
                if mode == 2:      # wbc_multi_tick_gather: gather_wait + tick + the gather of tick k beside tick k + 1, tau double-buffered, ONE call
                    tick_gather(k & 1)
                else:
                    tick()
                    if mode == 1:
                        ms.allgather_tau(n_total, outs, tau_all)
            t1 = time.perf_counter()
            ms.synchronize()
            times.append(time.perf_counter() - t0)
            total += times[-1]
            calls, sec = ms.host_stats()
            host.append((sec / args.steps * 1e6, (t1 - t0) / args.steps * 1e6))   # inside the C entry points / the whole Python loop, per tick
        return times, host

    for _ in range(args.warmup):
        tick()
    run(0)              # one untimed pass of blocks: the first measured mode otherwise carries clock / allocator warm-up (155 M against 294 M for the same ticks)
    blocks, host0 = run(0)
    sblocks, host1 = run(1)
    gblocks, host2 = run(2)
    el, gel, sel = float(np.median(blocks)), float(np.median(gblocks)), float(np.median(sblocks))
    med = lambda hs, i: float(np.median([h[i] for h in hs]))
    host_issue = {"issue_threads": ms.issue_threads,
                  "tick_call_us": med(host0, 0), "tick_loop_us": med(host0, 1),
                  "tick_gather_call_us": med(host2, 0), "tick_gather_loop_us": med(host2, 1),
                  "note": "host time per tick: *_call_us inside the C entry points (wbc_multi_host_stats), *_loop_us the Python loop around them; the device "
                          "needs ms_per_step -- the path is host-bound when the call time exceeds it"}
    # the same ticks with the issue threads forced on / off (wbc_solver_options.multi_threads = 1 / -1; auto = threads when the shards sit on more than
    # one device), same gather backend: what the caller pays per tick either way
    for tag, mt in (("threads", 1), ("serial", -1)):
        try:
            ms_v = W.MultiSolver(model, W.Params.from_dict(P, dtype), dtype=dtype, devices=devices, max_batch_total=n_total,
                                 gather="rccl" if distinct else "peer", options={"multi_threads": mt})
            tick_v, outs_v = ms_v.prepare_step(n_total, ins, obs_state, want_mats=not args.no_mats)
            tick_vb, outs_vb = ms_v.prepare_step(n_total, ins, obs_state, want_mats=not args.no_mats)
            alls_v = [ms_v.allgather_tau(n_total, outs_v), ms_v.allgather_tau(n_total, outs_vb)]
            tg_v = ms_v.prepare_tick_gather(n_total, [tick_v.capi, tick_vb.capi], alls_v)
            ms_v.synchronize()
            for _ in range(args.warmup):
                tick_v()
            bl_v, host_v = run(0, ms=ms_v, tick=tick_v, tick_gather=None)
            bl_g, host_g = run(2, ms=ms_v, tick=tick_v, tick_gather=tg_v)
            host_issue[tag] = {"issue_threads": ms_v.issue_threads, "empty_ticket_us": ms_v.probe_issue(), "tick_call_us": med(host_v, 0), "tick_loop_us": med(host_v, 1),
                               "value": args.steps * n_total / float(np.median(bl_v)),
                               "tick_gather_call_us": med(host_g, 0), "tick_gather_loop_us": med(host_g, 1),
                               "with_tau_allgather_value": args.steps * n_total / float(np.median(bl_g))}
            del ms_v, tick_v, tick_vb, tg_v
        except Exception as e:
            host_issue[tag] = {"error": repr(e)[:200]}
    ok = float(np.mean([(o["status"] == 0).double().mean().item() for o in outs]))
    print_line({
        "metric": "WBC control-steps/sec (batched DogBot)", "value": args.steps * n_total / el, "unit": "control-steps/s",
        "n_gpus": k, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": "configs[%d] per shard: batch=%d states, observer %s, %s; ONE process, %d shards through wbc_multi_* "
                               "on devices %s" % (args.config - 1, args.batch, "on" if obs else "off", dtype, k, devices),
                   "batch_per_gpu": args.batch, "parallelism": "single process, batch-sharded x%d (C-ABI wbc_multi_step_batch), no data-path collective" % k,
                   "distinct_devices": distinct},
        "timing": {"blocks": len(blocks), "steps_per_block": args.steps, "block_ms_median": el * 1e3, "block_ms_min": min(blocks) * 1e3,
                   "block_ms_max": max(blocks) * 1e3},
        "with_tau_allgather": {"value": args.steps * n_total / gel, "ms_per_step": gel / args.steps * 1e3,
                               "collective": ("RCCL ncclAllGather (ncclCommInitAll, one group call)" if distinct else "peer copies (shards share a device)")
                                             + "; wbc_multi_tick_gather: the gather on the gather streams beside the next tick, tau double-buffered, one call per tick",
                               "serial": {"value": args.steps * n_total / sel, "ms_per_step": sel / args.steps * 1e3,
                                          "collective": "wbc_multi_allgather_tau on the shard streams behind every tick"}},
        "host_issue": host_issue,
        "rccl_ranks": ms.rccl_ranks, "qp": {"status_ok_frac": ok}, "roofline": None, "cpu_baseline": None})


def rollout_setup(args, W, synth, torch, np, model, dtype, n, H, rank, local_rank, tracking):
    """buffers + the one-rollout closure of configs[4] (used by the --config 5 bench and by the N > 1 scale legs)"""
    td = torch.float64 if dtype == "f64" else torch.float32
    P = synth.default_params(observer_order=1, dtype=dtype)
    solver = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=local_rank, max_batch=n)
    B = synth.make_batch(3, n, model.total_mass, rank=rank)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
    inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu")}
    q0, v0 = inp["q"].clone(), inp["v"].clone()
    mask = torch.from_numpy(B["mask"]).cuda()
    text = np.zeros((n, 18))
    text[:, 0:3] = B["push"]
    tau_ext = dev(text)
    integ0 = solver.dynamics(q0, v0, want=("p",))["p"].clone()
    integ, rr = integ0.clone(), torch.zeros_like(integ0)
    out = dict(tau=torch.zeros((12, n), dtype=td, device="cuda"), f=torch.zeros((12, n), dtype=td, device="cuda"),
               status=torch.zeros(n, dtype=torch.int32, device="cuda"), iters=torch.zeros(n, dtype=torch.int32, device="cuda"),
               M=solver.empty(171, n), h=solver.empty(18, n), Jc=solver.empty(216, n), pf=solver.empty(12, n))
    plan = None
    if tracking:
        solver.set_ref_params(synth.default_ref_params())
        plan = dev(synth.make_plan(B, rank=rank))

    def one_rollout():
        inp["q"].copy_(q0); inp["v"].copy_(v0); integ.copy_(integ0); rr.zero_(); out["tau"].zero_(); out["f"].zero_()
        if plan is not None:
            solver.rollout_tracking(H, inp["q"], inp["v"], plan, inp["normals"], inp["mu"], mask, out, inp["w_des"],
                                    inp["vdot_des"], integ, rr, tau_ext)
        else:
            solver.rollout(H, inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, out, integ,
                           rr, tau_ext)
    return {"one_rollout": one_rollout, "out": out, "solver": solver, "B": B, "P": P, "tau_ext_np": text,
            "keep": (inp, q0, v0, mask, tau_ext, integ0, integ, rr, plan)}


# Algorithmic words per state and tick of a rollout: read q 19 + v 18 + w_des 6 + vdot_des 18 + normals 12 + mu 4 + mask 1 +
# tau_ext 18 + observer state 36 + previous tau, f 24 = 156; written q, v 37 + tau, f 24 + observer state 36 + h 18 = 115.  Round 5: the persistent
# kernel writes M and Jc (225 data words + 162 structural ones) ONCE per launch, in its last tick -- the integrator takes them from LDS -- which the
# per-tick figure below leaves out (387 / horizon words per tick more)
ROLLOUT_WORDS_PER_TICK = 156 + 115


def rollout_roofline(solver, r5, torch, np, n, H, dtype, reps=20):
    """HIP start / stop events of the rollout dispatches themselves (wbc_solver_enable_timing: on the launch stream) over `reps` rollouts:
    average launch duration of the persistent rollout_kernel, or the per-tick kernels' sum for batches beyond one launch."""
    solver.enable_timing(1)
    for _ in range(reps):
        r5["one_rollout"]()
    torch.cuda.synchronize()
    tm = solver.collect_timing()
    solver.enable_timing(0)
    if tm.get("rollout_launches", 0):
        return {"kernel": "rollout_kernel", "avg_launch_us": tm["rollout_ms"] * 1e3 / tm["rollout_launches"], "launches_timed": tm["rollout_launches"],
                "ticks_per_launch": H}
    per_rollout = sum(tm[k + "_ms"] for k in ("dyn", "qp", "rnea", "fused", "qp_lane")) * 1e3 / reps
    return {"kernel": "per-tick launches (front half, QP; the integrate kernel is not instrumented)", "avg_launch_us": per_rollout,
            "launches_timed": sum(tm[k + "_launches"] for k in ("dyn", "qp", "rnea", "fused", "qp_lane")), "ticks_per_launch": H}


def rollout_measurement_objects(rf, cpu, value, n, H, dtype, world):
    """`roofline` (whole path against the fp64 vector-FMA peak: the rollout is arithmetic / latency-bound) and the bytes the
    rollout moves against HBM, from this run's event timing (rf) and the oracle's instrumented operation count (cpu)."""
    ts = 8 if dtype == "f64" else 4
    fl = cpu["flops_per_tick"] if cpu else None
    launch_s = rf["avg_launch_us"] * 1e-6
    steps_in_launch = n * H
    tf = fl * steps_in_launch / launch_s / 1e12 if fl else None
    peak = VALU_F64_PEAK_TFLOPS * (1.0 if dtype == "f64" else 2.0)
    gbs = ROLLOUT_WORDS_PER_TICK * ts * steps_in_launch / launch_s / 1e9
    return {"kernel": rf["kernel"], "bound": "valu_f64" if dtype == "f64" else "valu_f32", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
            "frac": (tf / peak) if tf else None, "traffic": None,
            "flops_per_tick": fl, "avg_launch_us": rf["avg_launch_us"], "us_per_tick": rf["avg_launch_us"] / H, "launches_timed": rf["launches_timed"],
            "note": "ALGORITHMIC flops = the oracle's instrumented operation count per tick (control step with warm-started QP + forward dynamics + "
                    "integrator) x rollouts x ticks of one launch / that launch's average duration (HIP start / stop events of the dispatch on the "
                    "launch stream); peak = fp64 vector FMA, 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz.  The rollout is a chain of dependent ticks "
                    "on one workgroup per 4 robots: latency-bound, see DESIGN.md 4.7",
            "bytes": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                      "algorithmic_words_per_state_and_tick": ROLLOUT_WORDS_PER_TICK,
                      "note": "the state a rollout actually moves per tick (inputs; q / v / tau / f / h / observer state written back; M, Jc once per "
                              "launch, not counted) -- far from binding"}}


def rollout_bench(args, W, synth, torch, np, dist, world, rank, local_rank):
    """BASELINE.json configs[4]: MPC-style WBC-in-the-loop rollouts, horizon 20 x 1024 states per GPU (the config does
    not say whether 1024 is per GPU or total; per GPU here, --batch overrides).  One "step" = one rollout = `horizon`
    dependent ticks of {dyn_sweep, QP, forward dynamics + integrator}; value is still control-steps/s."""
    dtype = args.dtype or "f64"
    n = args.batch if args.batch != 4096 else 1024
    H = args.horizon
    model = load_model(W, torch, dist, rank, local_rank)
    r5 = rollout_setup(args, W, synth, torch, np, model, dtype, n, H, rank, local_rank, args.tracking)
    one_rollout, out = r5["one_rollout"], r5["out"]

    for _ in range(max(1, args.warmup)):
        one_rollout()
    torch.cuda.synchronize()
    blocks = timed_blocks(one_rollout, args.steps, dist, torch)
    elapsed = float(np.median(blocks))
    ok = float((out["status"].cpu().numpy() == 0).mean())
    iters_last = float(out["iters"].double().mean().item())
    rf = rollout_roofline(r5["solver"], r5, torch, np, n, H, dtype)
    if rank == 0:
        value = args.steps * H * n * world / elapsed
        cpu = None
        if not args.no_cpu and world == 1 and dtype == "f64" and not args.tracking:
            cpu = cpu_baseline_rollout(r5["B"], r5["P"], n, H, r5["tau_ext_np"])
        flops_src = cpu
        if flops_src is None:   # (N > 1, fp32 or the planner in the loop: the operation count alone, no timed CPU leg)
            try:
                flops_src = {"flops_per_tick": rollout_flops_only(r5, H)}
            except Exception:
                flops_src = None
        warm_on = os.environ.get("WBC_ROLLOUT_WARM", "1") != "0"
        print_line({
            "metric": "WBC control-steps/sec (batched DogBot)", "value": value,
            "unit": "control-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": "configs[4]: horizon=%d x batch=%d rollouts per GPU, trot masks, observer on, pushes, %s; "
                                   "one step = one rollout (%d dependent ticks incl. %sforward dynamics + integrator)"
                                   % (H, n, dtype, H, "CoM planner/reference generator + " if args.tracking else ""),
                       "batch_per_gpu": n, "horizon": H, "parallelism": "batch-sharded x%d, rank-local for all ticks" % world,
                       "qp_warm_start": ("ticks after the first start their QPs from the previous tick's active set (wbc_solver_options.rollout_warm = 1; "
                                         "results do not depend on it)" if warm_on else "off (rollout_warm = 0): every tick solves from the unconstrained minimum"),
                       "launches": ("one persistent rollout_kernel launch per rollout" if (n <= 4096 and os.environ.get("WBC_ROLLOUT_PERSISTENT", "1") != "0")
                                    else "per tick: %sdyn_sweep/fused tick, qp, integrate" % ("reference, " if args.tracking else ""))},
            "us_per_tick": elapsed / args.steps / H * 1e6, "qp": {"status_ok_frac_last_tick": ok, "iters_mean_last_tick": iters_last},
            "timing": {"blocks": len(blocks), "steps_per_block": args.steps, "block_ms_min": min(blocks) * 1e3,
                       "block_ms_median": elapsed * 1e3, "block_ms_max": max(blocks) * 1e3},
            "rccl_ranks": world if dist is not None else None,
            "roofline": rollout_measurement_objects(rf, flops_src, value, n, H, dtype, world), "cpu_baseline": cpu})
    if dist is not None:
        dist.destroy_process_group()


def rollout_flops_only(r5, H):
    import numpy as np
    import wbc_quadruped_dob_amd as W
    from oracle import oracle_py, urdf_model
    orc = oracle_py.Oracle(urdf_model.load_urdf(W.SYNTHETIC_URDF))
    B, P, te = r5["B"], r5["P"], r5["tau_ext_np"]
    m = min(B["q"].shape[0], 32)
    integ0 = orc.dynamics(B["q"][:m], B["v"][:m])["p"]
    return float(np.mean([orc.op_count_rollout(P, H, B["q"][i], B["v"][i], B["w_des"][i], B["vdot_des"][i], B["normals"][i], B["mu"][i],
                                               int(B["mask"][i]), te[i], integ0[i], warm=True)["flops_per_tick"] for i in range(m)]))


def qp_dense_general(W, torch, dtype, with_cpu=False):
    """The general dense QP kernel (wbc_qp_dense_batch: run-time sizes, one QP per wavefront, factors in LDS) on random strictly
    convex problems generated on the device: the size of the controller's own GRF QP, the size of a whole-body QP over CoM / joint accelerations
    and contact forces with torque bounds (30 variables, 58 rows, 18 of them equalities), the largest size it takes, and ONE such problem per
    launch (a single robot's tick through the general path).  Reported beside the structured path, not part of `value`."""
    td = torch.float64 if dtype == "f64" else torch.float32
    out = {}
    for n, m, meq, N in ((12, 24, 0, 4096), (30, 58, 18, 4096), (36, 64, 12, 4096), (30, 58, 18, 1)):
        gen = torch.Generator(device="cuda").manual_seed(1234 + n)
        A = torch.randn(N, n, n, dtype=torch.float64, device="cuda", generator=gen)
        H = (A @ A.transpose(1, 2)) / n + torch.eye(n, dtype=torch.float64, device="cuda")
        H = 0.5 * (H + H.transpose(1, 2))
        xf = torch.randn(N, n, dtype=torch.float64, device="cuda", generator=gen)
        Cm = torch.randn(N, m, n, dtype=torch.float64, device="cuda", generator=gen)
        slack = torch.rand(N, m, dtype=torch.float64, device="cuda", generator=gen)
        slack[:, :meq] = 0
        d = torch.einsum("kij,kj->ki", Cm, xf) - slack
        g = -torch.einsum("kij,kj->ki", H, xf + 2 * torch.randn(N, n, dtype=torch.float64, device="cuda", generator=gen))
        H, g, Cm, d = (t.to(td).contiguous() for t in (H, g, Cm, d))
        tol = 1e-9 if dtype == "f64" else 1e-3
        for _ in range(3):
            o = W.qp_dense_batch(H, g, Cm, d, meq=meq, max_iter=400, tol=tol)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            o = W.qp_dense_batch(H, g, Cm, d, meq=meq, max_iter=400, tol=tol)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        key = "n%d_m%d_meq%d%s" % (n, m, meq, "" if N > 1 else "_single")
        it_mean = float(o["iters"].double().mean())
        # work of one Goldfarb-Idnani iteration at these sizes: d = J^T n+ (2 n^2), z = J2 d2 (<= 2 n^2), r = R^-1 d1 (<= n^2), the slacks of
        # the m rows (2 m n), the Givens update of J (<= 6 n^2): ~ 11 n^2 + 2 m n flops
        fl_it = 11.0 * n * n + 2.0 * m * n
        out[key] = {"batch": N, "launch_us": us, "qps_per_s": N / us * 1e6, "iters_mean": it_mean,
                    "status_ok_frac": float((o["status"] == 0).double().mean()),
                    "us_per_iteration_of_a_wavefront": us / max(it_mean, 1e-9) if N == 1 else None,
                    "roofline": {"bound": "valu_%s" % dtype, "achieved": fl_it * it_mean * N / (us * 1e-6) / 1e12, "peak": VALU_F64_PEAK_TFLOPS * (1 if dtype == "f64" else 2),
                                 "unit": "TFLOP/s", "flops_per_iteration_estimate": fl_it,
                                 "note": "O(n^2) work per iteration spread over 64 lanes with an LDS round trip between its steps: latency-bound, one wavefront per QP"}}
        out[key]["roofline"]["frac"] = out[key]["roofline"]["achieved"] / out[key]["roofline"]["peak"]
        if with_cpu and dtype == "f64":   # the same problems through the oracle's general solver on the host cores (the cpu_baseline of THIS kernel)
            try:
                from oracle import oracle_py
                Hn, gn, Cn, dn = (t.double().cpu().numpy() for t in (H, g, Cm, d))
                t0 = time.perf_counter()
                xr, _, sr, itr = oracle_py.qp_general(Hn, gn, Cn, dn, meq=meq, max_iter=400, tol=tol) if N > 1 else oracle_py.qp_general(Hn[0], gn[0], Cn[0], dn[0], meq=meq, max_iter=400, tol=tol)
                dtc = time.perf_counter() - t0
                if N > 1:
                    t0 = time.perf_counter()
                    oracle_py.qp_general(Hn, gn, Cn, dn, meq=meq, max_iter=400, tol=tol)
                    dtc = time.perf_counter() - t0
                    agree = float(np_abs_max(o["x"].double().cpu().numpy() - xr) / max(1.0, np_abs_max(xr)))
                else:
                    reps_c = 200
                    t0 = time.perf_counter()
                    for _ in range(reps_c):
                        oracle_py.qp_general(Hn[0], gn[0], Cn[0], dn[0], meq=meq, max_iter=400, tol=tol)
                    dtc = (time.perf_counter() - t0) / reps_c
                    agree = float(np_abs_max(o["x"].double().cpu().numpy()[0] - xr) / max(1.0, np_abs_max(xr)))
                out[key]["cpu_oracle"] = {"us": dtc * 1e6, "qps_per_s": N / dtc, "threads": 8 if N > 1 else 1, "kind": "port",
                                          "max_rel_difference_of_x": agree,
                                          "note": "oracle/qp_general.hpp on the same problems (OpenMP over problems, 8 threads%s)" % ("" if N > 1 else "; one problem: one thread, incl. the ctypes call")}
            except Exception as e:
                out[key]["cpu_oracle"] = {"error": repr(e)[:200]}
    out["note"] = ("random feasible problems (H = A A^T / n + I, a third of the rows active at the optimum), same-stream back-to-back launches; the "
                   "controller's own 12-variable GRF QP goes through the structured kernels instead (kernels.qp_us)")
    return out


def np_abs_max(a):
    import numpy as np
    return float(np.abs(a).max()) if a.size else 0.0


def qp_latency(W, synth, torch, np, model, B, P, dtype, td, obs):
    """BASELINE.json metric, second half ("p50 QP us"): one state per launch, 1000 synchronous ticks.
    tick_p50 = host-observed wall time of one wbc_step_batch(N=1) + stream sync (what a 1-robot control loop sees);
    qp_kernel_p50 = HIP-event span around the QP kernel alone in those ticks (raw, includes the event-pair cost)."""
    solver = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=torch.cuda.current_device(), max_batch=1)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a[:1].T)).to(td).cuda()
    inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
    mask = torch.from_numpy(B["mask"][:1].copy()).cuda()
    integ = rr = None
    if obs:
        integ = solver.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone()
        rr = torch.zeros_like(integ)
    out = {}

    def tick():
        return solver.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"],
                           inp["f_prev"], integ, rr, out=out, want_mats=False)

    out.update(tick())
    for _ in range(20):
        tick()
    torch.cuda.synchronize()
    wall, qpk, dynk, fusk = [], [], [], []
    for _ in range(1000):                      # host-observed latency, no instrumentation in the way
        t0 = time.perf_counter()
        tick()
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
    solver.enable_timing(1)
    for _ in range(200):                       # kernel spans (HIP events on the launch stream)
        tick()
        torch.cuda.synchronize()
        tm = solver.collect_timing()
        qpk.append(tm["qp_ms"])
        dynk.append(tm["dyn_ms"] + tm["rnea_ms"])
        fusk.append(tm.get("fused_ms", 0.0))
    solver.enable_timing(0)
    fused = float(np.median(fusk)) > 0
    # BASELINE.json "p50 QP us": the GRF QP as its own kernel at N = 1, measured here with a second solver whose tick is
    # the two-kernel form (wbc_solver_options.fused_max = 0): dispatch start/stop events of the QP kernel, 300 ticks
    s2 = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=torch.cuda.current_device(), max_batch=1,
                  options={"fused_max": 0})
    out2 = {}
    integ2 = None if integ is None else integ.clone()
    rr2 = None if rr is None else torch.zeros_like(rr)

    def tick2():
        return s2.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"],
                       inp["f_prev"], integ2, rr2, out=out2, want_mats=False)

    out2.update(tick2())
    for _ in range(20):
        tick2()
    torch.cuda.synchronize()
    s2.enable_timing(1)
    qp2, front2 = [], []
    for _ in range(300):
        tick2()
        torch.cuda.synchronize()
        tm = s2.collect_timing()
        qp2.append(tm["qp_ms"])
        front2.append(tm["dyn_ms"] + tm["rnea_ms"])
    s2.enable_timing(0)
    # the host-pointer single-robot call (BASELINE.json configs[0] shape): staging copy in, tick, copy out, one sync
    h = {k: np.ascontiguousarray(B[k][0], dtype=np.float64) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
    hig = np.zeros(18) if obs else None
    hr = np.zeros(18) if obs else None
    ct = lambda: solver.compute_torques(h["q"], h["v"], h["w_des"], h["vdot_des"], h["normals"], h["mu"], int(B["mask"][0]),
                                        h["tau_prev"] if obs else None, h["f_prev"] if obs else None, hig, hr)
    for _ in range(20):
        ct()
    cwall = []
    for _ in range(500):
        t0 = time.perf_counter()
        ct()
        cwall.append(time.perf_counter() - t0)
    # the same loop on the solver's pinned image (wbc_one_map / wbc_one_tick: no staging copies), per completion mode
    one = {}
    if dtype == "f64":
        for zc, label in ((0, "copies+sync"), (1, "zerocopy+sync"), (2, "zerocopy+stream_ticket"), (3, "zerocopy+kernel_ticket")):
            try:
                sz = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=torch.cuda.current_device(), max_batch=1,
                              options={"one_zerocopy": zc})
                img = sz.one_image()
                for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev"):
                    img[k][:] = h[k]
                img["mask"][0] = int(B["mask"][0])
                if obs:
                    ig0, r0 = sz.observer_init(h["q"], h["v"])
                    img["obs_integ"][:] = ig0
                    img["obs_r"][:] = r0
                for _ in range(50):
                    sz.one_tick()
                ow = []
                for _ in range(1000):
                    t0 = time.perf_counter()
                    sz.one_tick()
                    ow.append(time.perf_counter() - t0)
                one[label] = {"p50_us": float(np.median(ow)) * 1e6, "p99_us": float(np.percentile(ow, 99)) * 1e6, "status": int(img["status"][0])}
                del sz
            except Exception as e:
                one[label] = {"error": repr(e)[:200]}
    return {"ticks": 1000, "tick_p50_us": float(np.median(wall)) * 1e6, "tick_p99_us": float(np.percentile(wall, 99)) * 1e6,
            "compute_torques_p50_us": float(np.median(cwall)) * 1e6, "compute_torques_p99_us": float(np.percentile(cwall, 99)) * 1e6,
            "tick_kernel_p50_us": float(np.median(fusk)) * 1e3 if fused else None,
            "qp_kernel_p50_us": float(np.median(qp2)) * 1e3, "qp_kernel_p99_us": float(np.percentile(qp2, 99)) * 1e3,
            "front_kernel_p50_us": float(np.median(front2)) * 1e3,
            "one_tick_on_pinned_image": one,
            "note": "N=1 per launch, synchronous wbc_step_batch without M/h/Jc outputs; tick = host wall time incl. launch + stream "
                    "sync of the default dispatch (%s); tick_kernel = its kernel span over 200 further ticks; qp_kernel / "
                    "front_kernel = the GRF QP (assembly + solve + torque map) and the rnea_step front half as their own kernels, "
                    "measured in this run on a second solver with fused_max = 0 (dispatch start/stop events, 300 ticks)"
                    % ("one fused launch: rnea_step and qp_group16 as wavefront roles" if fused else "rnea_step -> qp")}


def sweep_alone_roofline(solver, torch, inp, n, dtype, ts):
    """The 443-word dynamics stage as its own kernel (wbc_dynamics_batch: q, v -> M, h, Jc) at the bench batch: what the
    bench line's `roofline` measured before small batches ran the tick as one fused launch."""
    out = solver.dynamics(inp["q"], inp["v"], want=("M", "h", "Jc"))
    for _ in range(10):
        solver.dynamics(inp["q"], inp["v"], want=("M", "h", "Jc"), out=out)
    torch.cuda.synchronize()
    solver.enable_timing(1)
    for _ in range(100):
        solver.dynamics(inp["q"], inp["v"], want=("M", "h", "Jc"), out=out)
    torch.cuda.synchronize()
    tm = solver.collect_timing()
    solver.enable_timing(0)
    t = tm["dyn_ms"] * 1e-3 / tm["dyn_launches"]
    ach = DYN_WORDS_FUSED * ts * n / t / 1e9
    return {"kernel": "dyn_sweep_kernel (M, h, Jc only)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "avg_launch_us": t * 1e6, "launches_timed": tm["dyn_launches"],
            "traffic": pmc_traffic("dyn_sweep_kernel<%s, 1," % ("double" if dtype == "f64" else "float"), n, dtype),
            "algorithmic_words_per_state": DYN_WORDS_FUSED}


PMC_SOURCE = ("profiles/pmc_latest.json: committed rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate runs) of this "
              "workload on MI355X -- NOT collected inside this run")


def pmc_traffic(kernel, n, dtype):
    """HBM bytes per launch from the committed PMC summary of this round (separate rocprofv3 --pmc passes with the
    FETCH_SIZE x2 correction calibrated in tools/pmc_profile.sh); None when no matching profile is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        d = json.load(open(path))
        for key in (("n%d" % n, "n%d_f32" % n) if dtype != "f64" else ("n%d" % n,)):
            for k, v in d.get(key, {}).items():
                if kernel in k and ("double" in k) == (dtype == "f64"):
                    return v.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def baseline_configs_leg(W, synth, torch, np, model, args):
    """BASELINE.json's other configs beside `value` (N = 1, full line only; `value` stays configs[1]): configs[2] (4 096 states, trot masks, observer on, fp64), the per-GPU
    shard of configs[3] (32 768 fp32 states, tilted normals, observer on) and configs[4] (horizon-20 rollouts of 1 024 robots) -- each a short timed run of the same
    bracketing (timed_blocks: blocks of K ticks, barrier-free at N = 1, synchronize on both sides, median block), so that the driver's one command records them too."""
    res = {}
    for key, cfg, n, dtype, K in (("configs[2]_n4096_f64_observer_on", 3, 4096, "f64", 100), ("configs[3]_shard_n32768_f32_observer_on", 4, 32768, "f32", 50)):
        try:
            td = torch.float64 if dtype == "f64" else torch.float32
            P = synth.default_params(observer_order=1, dtype=dtype)
            solver = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=torch.cuda.current_device(), max_batch=n)
            B = synth.make_batch(cfg, n, model.total_mass, rank=0)
            dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
            inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
            mask = torch.from_numpy(B["mask"]).cuda()
            integ = solver.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone()
            rr = torch.zeros_like(integ)
            tick, out = solver.prepare_step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"], inp["f_prev"], integ, rr, want_mats=True)
            for _ in range(30):
                tick()
            bl = timed_blocks(tick, K, None, torch, min_total_s=0.03)
            el = float(np.median(bl))
            pl = solver.plan_tick(n)
            res[key] = {"value": K * n / el, "unit": "control-steps/s", "ms_per_step": el / K * 1e3, "steps_per_block": K, "blocks": len(bl), "dtype": dtype,
                        "status_ok_frac": float((out["status"] == 0).double().mean().item()),
                        "kernels": ("one launch: tile_tick_kernel" if pl["fused"] == 2 else "one launch: fused_tick_kernel" if pl["fused"] == 1 else "one launch: fused_pair_kernel" if pl["fused"] == 3 else "front %d -> qp %d" % (pl["front"], pl["qp"])),
                        "plan": pl}
            del solver, tick, out, inp, integ, rr
            torch.cuda.empty_cache()
        except Exception as e:   # never lose the headline to an extra leg
            res[key] = {"error": repr(e)[:200]}
    try:
        r5 = rollout_setup(args, W, synth, torch, np, model, "f64", 1024, args.horizon, 0, torch.cuda.current_device(), False)
        for _ in range(3):
            r5["one_rollout"]()
        bl = timed_blocks(r5["one_rollout"], 10, None, torch, min_total_s=0.03)
        el = float(np.median(bl))
        res["configs[4]_h%d_n1024_f64" % args.horizon] = {"value": 10 * args.horizon * 1024 / el, "unit": "control-steps/s", "ms_per_rollout": el / 10 * 1e3,
                                                        "us_per_tick": el / 10 / args.horizon * 1e6, "rollouts_per_block": 10, "blocks": len(bl), "dtype": "f64"}
        del r5
        torch.cuda.empty_cache()
    except Exception as e:
        res["configs[4]"] = {"error": repr(e)[:200]}
    return res


def large_batch_roofline(W, synth, torch, np, model, args, dtype, td, obs, split):
    """The same two kernels at 262144 states (the largest BASELINE.json batch): where the sweep is bandwidth-bound
    rather than launch/latency-bound.  Not part of `value`."""
    n = args.large_batch
    torch.cuda.empty_cache()   # fresh 2 MiB-aligned segments for the big buffers instead of holes of the small run's cache
    P = synth.default_params(observer_order=obs, dtype=dtype)
    solver = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=torch.cuda.current_device(), max_batch=n)
    B = synth.make_batch(args.config, n, model.total_mass, rank=0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
    inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
    mask = torch.from_numpy(B["mask"]).cuda()
    integ = rr = None
    if obs:
        integ = solver.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone()
        rr = torch.zeros_like(integ)
    out = solver.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"],
                      inp["f_prev"], integ, rr, want_mats=True)
    for _ in range(40):   # clocks and TLBs settle: the first launches at this size read 5-8 % slower
        solver.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"],
                    inp["f_prev"], integ, rr, out=out, want_mats=True)
    torch.cuda.synchronize()
    solver.enable_timing(1)
    K = 40
    t0 = time.perf_counter()
    for _ in range(K):
        solver.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"],
                    inp["f_prev"], integ, rr, out=out, want_mats=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tm = solver.collect_timing()
    solver.enable_timing(0)
    ts = 8 if dtype == "f64" else 4
    dyn_s = tm["dyn_ms"] * 1e-3 / tm["dyn_launches"]
    qp_s = tm["qp_ms"] * 1e-3 / tm["qp_launches"]
    ach = dyn_words(split) * ts * n / dyn_s / 1e9
    # the dynamics stage exactly as SURVEY.md 8(d) defines it (q, v -> M, h, Jc as its own kernel, no step prologue)
    # the same ticks with wbc_solver_options.keep_structural = 1: the 162 structural words of M and Jc per state stay as the first
    # tick wrote them (the caller's promise not to touch M / Jc between ticks); the sweep then stores 281 of the 443 words
    kept = None
    try:
        s2 = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=torch.cuda.current_device(), max_batch=n,
                      options={"keep_structural": 1})
        integ2 = None if integ is None else integ.clone()
        rr2 = None if rr is None else torch.zeros_like(rr)
        out2 = {k: out[k] for k in ("M", "h", "Jc", "pf")}
        for _ in range(20):
            o2 = s2.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"],
                         inp["f_prev"], integ2, rr2, out=out2, want_mats=True)
            out2.update(o2)
        torch.cuda.synchronize()
        s2.enable_timing(1)
        t0 = time.perf_counter()
        for _ in range(K):
            s2.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"],
                    inp["f_prev"], integ2, rr2, out=out2, want_mats=True)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        tm2 = s2.collect_timing()
        s2.enable_timing(0)
        d2 = tm2["dyn_ms"] * 1e-3 / max(1, tm2["dyn_launches"])
        kept = {"steps_per_s": K * n / el2, "ms_per_step": el2 / K * 1e3, "dyn_sweep_us": d2 * 1e6, "stored_words_per_state": 405 - 162,
                "achieved_GBps_on_281_words": 281 * ts * n / d2 / 1e9,
                "note": "wbc_solver_options.keep_structural = 1: M / Jc structural zeros and ones written by the first tick only"}
        del s2
    except Exception as e:
        kept = {"error": repr(e)[:200]}
    # the same ticks for a caller that passes no M / h / Jc buffers (tau, f only -- what a controller consumes): rnea_step front half, no CRBA,
    # no matrix stores; observer-on batches run the observer kernel + the observer-free rnea_step
    tf_only = None
    try:
        s3 = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=torch.cuda.current_device(), max_batch=n)
        integ3 = None if integ is None else integ.clone()
        rr3 = None if rr is None else torch.zeros_like(rr)
        o3 = s3.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"], inp["f_prev"], integ3, rr3,
                     want_mats=False)
        for _ in range(20):
            s3.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"], inp["f_prev"], integ3, rr3,
                    out=o3, want_mats=False)
        torch.cuda.synchronize()
        s3.enable_timing(1)
        t0 = time.perf_counter()
        for _ in range(K):
            s3.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"], inp["f_prev"], integ3, rr3,
                    out=o3, want_mats=False)
        torch.cuda.synchronize()
        el3 = time.perf_counter() - t0
        tm3 = s3.collect_timing()
        s3.enable_timing(0)
        us = lambda k: (tm3[k + "_ms"] * 1e3 / tm3[k + "_launches"]) if tm3.get(k + "_launches", 0) else None
        plan3 = s3.plan_tick(n, want_mats=False, want_pf=False)
        tf_only = {"steps_per_s": K * n / el3, "ms_per_step": el3 / K * 1e3, "plan": plan3,
                   "rnea_step_us": us("dyn") if plan3["front"] == 3 else us("rnea"), "observer_us": us("rnea") if plan3["front"] == 3 else None,
                   "qp_us": us("qp"), "qp_lane_us": us("qp_lane"), "status_ok_frac": float((o3["status"] == 0).double().mean()),
                   "whole_path_GBps": (102 + (96 if obs else 0)) * ts * K * n / el3 / 1e9,
                   "note": "out.M = out.h = out.Jc = NULL: SURVEY.md 8(d)'s whole-path bytes (inputs + tau, f [+ observer state]) are all this tick moves "
                           "besides its 66-word workspace; plan.front = 3: observer kernel + observer-free rnea_step"}
        del s3
    except Exception as e:
        tf_only = {"error": repr(e)[:200]}
    alone = sweep_alone_roofline(solver, torch, inp, n, dtype, ts)
    alone["traffic"] = pmc_traffic("dyn_sweep_kernel<%s, 1," % ("double" if dtype == "f64" else "float"), n, dtype)
    return {"batch": n, "kernel": dyn_kernel_name(split), "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "dynamics_stage_alone": alone, "keep_structural": kept, "tau_f_only": tf_only,
            "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(tick_sweep_symbol(dtype, obs, n), n, dtype),
            "algorithmic_words_per_state": dyn_words(split), "avg_launch_us": dyn_s * 1e6,
            "rnea_step_us": tm["rnea_ms"] * 1e3 / max(1, tm["rnea_launches"]), "qp_us": qp_s * 1e6,
            "qp_lane_us": (tm.get("qp_lane_ms", 0.0) * 1e3 / tm["qp_lane_launches"]) if tm.get("qp_lane_launches", 0) else None,
            "qp_note": ("GRF QP stage = qp_lane_kernel (one state per lane, semismooth Newton; qp_lane_us) + qp_list_kernel (dense active-set "
                        "solver over the states the first did not finish; qp_us)" if tm.get("qp_lane_launches", 0) else
                        "GRF QP stage = the dense active-set kernel (qp_us)"),
            "steps_per_s": K * n / el, "ms_per_step": el / K * 1e3}


def host_cpu_info():
    """What the host gives this process: logical CPUs, the affinity mask, the cgroup CPU quota, the OpenMP binding in force."""
    info = {"nproc": os.cpu_count()}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except AttributeError:
        info["affinity"] = info["nproc"]
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                quota = None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            info["cgroup_cpu_quota_source"] = path
            break
        except Exception:
            continue
    info["cgroup_cpu_quota_cores"] = quota
    info["usable_cores"] = int(min(info["affinity"], quota)) if quota else info["affinity"]
    info["OMP_PROC_BIND"] = os.environ.get("OMP_PROC_BIND")
    info["OMP_PLACES"] = os.environ.get("OMP_PLACES")
    return info


def _thread_counts(usable):
    return sorted({t for t in (2, 4, 8, 16, 32, 64, 128, usable) if 1 < t <= usable})


def cpu_baseline(B, P, dtype, n):
    """The build's CPU restatement (oracle, kind "port") timed on this box's host cores: bounded sample.  Every thread count runs
    `reps` passes over the batch inside ONE OpenMP region (static schedule, per-thread scratch on the thread's stack), so thread
    start-up and the fork / join of a region per 4 096-state call -- what made round 3's 32 threads 35 % efficient and its 256
    threads collapse -- are outside the clock; thread counts beyond the cgroup quota / affinity mask are not tried."""
    import numpy as np
    import wbc_quadruped_dob_amd as W
    os.environ.setdefault("OMP_PROC_BIND", "close")     # (read when the oracle's libgomp loads, below)
    os.environ.setdefault("OMP_PLACES", "cores")
    os.environ.setdefault("OMP_WAIT_POLICY", "active")
    from oracle import oracle_py, urdf_model
    orc = oracle_py.Oracle(urdf_model.load_urdf(W.SYNTHETIC_URDF))
    nd = np.float64 if dtype == "f64" else np.float32
    c = lambda a: np.ascontiguousarray(a, dtype=nd)
    args = [c(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu")] + [B["mask"], c(B["tau_prev"]), c(B["f_prev"])]
    host = host_cpu_info()
    usable = max(1, host["usable_cores"])

    def run(threads, budget):
        sec = orc.step_timed(P, *args, reps=1, nthreads=threads)           # warm: threads exist, pages touched
        reps = max(2, int(budget / max(sec, 1e-6)))
        sec = orc.step_timed(P, *args, reps=reps, nthreads=threads)
        return reps * n / sec, reps

    one, r1 = run(1, 4.0)
    best, bt, br = one, 1, r1
    tried = {1: one}
    for t in _thread_counts(usable):
        val, reps = run(t, 2.5)
        tried[t] = val
        if val > best:
            best, bt, br = val, t, reps
    # SURVEY.md 8d extras from the same oracle: per-QP wall time on one thread, instrumented operation count per step
    extras = {}
    if dtype == "f64":
        ns, _ = orc.qp_time(P, *[args[i] for i in (0, 1, 2, 4, 5)], B["mask"])
        extras["qp_p50_us"] = float(np.median(ns)) * 1e-3
        extras["qp_p99_us"] = float(np.percentile(ns, 99)) * 1e-3
        ns_ = min(n, 256)
        integ0 = orc.dynamics(args[0][:ns_], args[1][:ns_])["p"]
        fl = []
        for i in range(ns_):
            oc = orc.op_count(P, *[a[i] for a in args[:6]], int(B["mask"][i]), args[7][i], args[8][i],
                              integ0[i] if P["observer_order"] else None, np.zeros(18) if P["observer_order"] else None)
            fl.append(oc["flops"])
        extras["flops_per_step"] = float(np.mean(fl))
        extras["flops_note"] = ("instrumented count (add+mul+div+sqrt+trig, FMA = 2) of the oracle's scalar type over the first "
                                "%d states; dense restatement, upper bound for a structure-exploiting kernel" % ns_)
    return {"value": best, "unit": "control-steps/s", "cores": bt, "kind": "port", **extras,
            "parallel_efficiency": best / (one * bt),
            "sample": "the same %d-state batch, %d passes inside one OpenMP region on %d thread(s) (static schedule, per-thread scratch); ~4 s "
                      "single-thread + ~2.5 s per thread count tried %s; host: %d logical CPUs, %d in the affinity mask, cgroup quota %s cores; "
                      "OMP_PROC_BIND=%s OMP_PLACES=%s; g++ -O2 -march=native build of oracle/"
                      % (n, br, bt, sorted(tried), host["nproc"], host["affinity"],
                         ("%.1f" % host["cgroup_cpu_quota_cores"]) if host["cgroup_cpu_quota_cores"] else "none",
                         os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES")),
            "host": host, "single_thread_value": one, "by_threads": {str(k): v for k, v in sorted(tried.items())}}


def cpu_baseline_rollout(B, P, n, H, tau_ext):
    """configs[4] on the host cores: the oracle's rollout() (horizon H dependent ticks incl. forward dynamics + integrator, QPs warm-started
    from the previous tick like the HIP path's) over the same n rollouts, OpenMP over rollouts; bounded sample, best thread count."""
    import numpy as np
    import wbc_quadruped_dob_amd as W
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    from oracle import oracle_py, urdf_model
    orc = oracle_py.Oracle(urdf_model.load_urdf(W.SYNTHETIC_URDF))
    host = host_cpu_info()
    usable = max(1, host["usable_cores"])
    integ0 = orc.dynamics(B["q"], B["v"], nthreads=min(8, usable))["p"]

    def run(threads, budget):
        t_used, reps = 0.0, 0
        while t_used < budget:
            q, v = B["q"].copy(), B["v"].copy()
            integ, r = integ0.copy(), np.zeros((n, 18))
            t0 = time.perf_counter()
            orc.rollout(P, H, q, v, B["w_des"], B["vdot_des"], B["normals"], B["mu"], B["mask"], tau_ext=tau_ext, integ=integ, r=r,
                        nthreads=threads, warm=True)
            t_used += time.perf_counter() - t0
            reps += 1
        return reps * n * H / t_used, reps

    run(min(8, usable), 0.2)
    one, _ = run(1, 3.0)
    best, bt, br = one, 1, 0
    tried = {1: one}
    for t in _thread_counts(usable):
        val, reps = run(t, 2.0)
        tried[t] = val
        if val > best:
            best, bt, br = val, t, reps
    oc = [orc.op_count_rollout(P, H, B["q"][i], B["v"][i], B["w_des"][i], B["vdot_des"][i], B["normals"][i], B["mu"][i], int(B["mask"][i]),
                               tau_ext[i], integ0[i], warm=w) for w in (True, False) for i in range(min(n, 64))]
    k = len(oc) // 2
    return {"value": best, "unit": "control-steps/s", "cores": bt, "kind": "port", "parallel_efficiency": best / (one * bt),
            "flops_per_tick": float(np.mean([o["flops_per_tick"] for o in oc[:k]])),
            "flops_per_tick_cold_qp": float(np.mean([o["flops_per_tick"] for o in oc[k:]])),
            "qp_iters_per_tick": float(np.mean([o["iters_sum"] for o in oc[:k]])) / H,
            "qp_iters_per_tick_cold": float(np.mean([o["iters_sum"] for o in oc[k:]])) / H,
            "sample": "the oracle's rollout() over the same %d rollouts x %d ticks (QPs warm-started from the previous tick's active set), OpenMP over "
                      "rollouts, %d repetitions on %d thread(s); ~3 s single-thread + ~2 s per thread count tried %s; host: %d logical CPUs, %d in the "
                      "affinity mask, cgroup quota %s cores" % (n, H, br, bt, sorted(tried), host["nproc"], host["affinity"],
                                                                ("%.1f" % host["cgroup_cpu_quota_cores"]) if host["cgroup_cpu_quota_cores"] else "none"),
            "host": host, "single_thread_value": one, "by_threads": {str(k_): v for k_, v in sorted(tried.items())},
            "flops_note": "instrumented count (add+mul+div+sqrt+trig, FMA = 2) of the oracle's rollout over the first %d rollouts, per tick: control step "
                          "+ dense 18 x 18 forward dynamics + integrator; dense restatement, upper bound for a structure-exploiting kernel" % k}


if __name__ == "__main__":
    main()

"""CPU, world_size 2 over gloo: the N>1 path (slice ownership, ragged all-gather, status all-reduce).
The per-rank compute is the CPU oracle injected as a stand-in (allowed in tests/ only); on the GPU box
the same ShardedBatch wraps Solver.step over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wbc_quadruped_dob_amd import synth
from wbc_quadruped_dob_amd.sharding import ShardedBatch, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 4096, 262144, 262147):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import wbc_quadruped_dob_amd as W
        from oracle import oracle_py, urdf_model
        flat = urdf_model.load_urdf(W.SYNTHETIC_URDF)
        orc = oracle_py.Oracle(flat)
        P = synth.default_params(observer_order=0)
        B = synth.make_batch(3, n_total, float(flat["mass"].sum()))  # same seed on every rank = the full batch
        keys = ("q", "v", "w_des", "vdot_des", "normals", "mu")

        def step_fn(loc):  # CPU stand-in for Solver.step, component-major tensors in and out
            a = [loc[k].numpy().T.copy() for k in keys]
            o = orc.step(P, *a, loc["mask"].numpy())
            return dict(tau=torch.from_numpy(o["tau"].T.copy()), f=torch.from_numpy(o["f"].T.copy()),
                        status=torch.from_numpy(o["status"]), iters=torch.from_numpy(o["iters"]))

        sb = ShardedBatch(step_fn, n_total, dist)
        full = {k: torch.from_numpy(B[k].T.copy()) for k in keys}
        full["mask"] = torch.from_numpy(B["mask"])
        loc = {k: sb.local_slice(v).contiguous() for k, v in full.items()}
        out = sb.step(loc)
        tau_all = sb.gather(out["tau"])
        stats = sb.status_counts(out["status"], out["iters"])
        # the bench's "with all-gather" leg needs equal slices: run it on the first 500 states of each rank
        from wbc_quadruped_dob_amd.sharding import agree_on_steps, timed_steps_with_gather
        eq = {k: v[..., :500].contiguous() for k, v in loc.items()}
        # each rank proposes its own step count (as bench.py's ranks do from their own clocks): the leg must run with the agreed one
        k_steps = agree_on_steps(2 + (rank % 2), dist)
        assert k_steps == (3 if world > 1 else 2)
        secs, gathered = timed_steps_with_gather(lambda: sb.step(eq), lambda o: o["tau"], dist, k_steps)
        assert secs > 0 and tuple(gathered.shape) == (world, 12, 500)
        assert torch.equal(gathered[rank], sb.step(eq)["tau"])
        # the double-buffered form (gather of tick k beside tick k + 1): two ticks that write DIFFERENT tau buffers, the second over
        # other states, so that a gather that read the wrong buffer -- or a buffer overwritten too early -- shows
        from wbc_quadruped_dob_amd.sharding import timed_steps_with_overlapped_gather
        eq2 = dict(eq)
        eq2["w_des"] = eq["w_des"] * 1.05
        keep = {}

        def mk(inputs, tag):
            def f():
                keep[tag] = sb.step(inputs)
                return keep[tag]
            return f
        secs2, (g0, g1) = timed_steps_with_overlapped_gather((mk(eq, "a"), mk(eq2, "b")), lambda o: o["tau"], dist, 5)
        assert secs2 > 0 and tuple(g0.shape) == tuple(g1.shape) == (world, 12, 500)
        assert torch.equal(g0[rank], sb.step(eq)["tau"]) and torch.equal(g1[rank], sb.step(eq2)["tau"])   # ticks 4 (even) and 3 (odd)
        assert not torch.equal(g0[rank], g1[rank])
        if rank == 0:
            ref = orc.step(P, *[B[k] for k in keys], B["mask"])
            q.put((float(np.abs(tau_all.numpy().T - ref["tau"]).max()), stats, int((ref["status"] == 0).sum()),
                   int(ref["iters"].sum()), int(ref["iters"].max()), sb.count))
    finally:
        dist.destroy_process_group()


def _run_world(world, n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=150)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return got


@pytest.mark.timeout(120)
def test_two_rank_gloo_matches_single_process():
    n_total = 1001  # ragged: 501 + 500
    err, stats, ok, isum, imax, cnt0 = _run_world(2, n_total)
    assert err == 0.0  # same code, same inputs, same slices: bit-identical
    assert cnt0 == 501
    assert stats["ok"] == ok == n_total and stats["iters_sum"] == isum and stats["iters_max"] == imax


@pytest.mark.timeout(240)
def test_eight_rank_gloo_ragged_matches_single_process():
    """The node's shape (8 ranks, BASELINE.json configs[3] / [4]) with a ragged split: 4 003 states = 3 ranks of 501 + 5 of 500;
    the timed all-gather leg runs on the equal first 500 states of every rank (what bench.py's N > 1 legs do)."""
    n_total = 4003
    assert [shard_range(n_total, 8, r)[1] for r in range(8)] == [501] * 3 + [500] * 5
    err, stats, ok, isum, imax, cnt0 = _run_world(8, n_total)
    assert err == 0.0
    assert cnt0 == 501
    assert stats["ok"] == ok == n_total and stats["iters_sum"] == isum and stats["iters_max"] == imax


def _bcast_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import wbc_quadruped_dob_amd as W
        from wbc_quadruped_dob_amd.sharding import broadcast_model
        # only rank 0 opens the URDF; the others receive the flat arrays and build their model from them
        mine = W.Model.from_urdf(W.SYNTHETIC_URDF).flat() if rank == 0 else None
        got = broadcast_model(mine, dist)
        model = W.Model.from_flat(got)
        back = model.flat()
        ref = W.Model.from_urdf(W.SYNTHETIC_URDF).flat()      # (the check only: what this rank would have parsed itself)
        same = all(np.array_equal(np.asarray(back[k]), np.asarray(ref[k])) for k in
                   ("parent", "Rt", "rt", "axis", "mass", "com", "Ic", "foot_body", "foot_off", "gravity"))
        dims = (model.nb, model.nq, model.nv, model.nj, model.nf)
        t = torch.tensor([1 if same else 0], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            q.put((int(t.item()), dims, abs(model.total_mass - float(ref["mass"].sum()))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_model_constants_are_broadcast_once():
    """rank 0 reads the robot description, every other rank builds its model from the broadcast flat arrays (one float64
    vector of < 4 kB) -- bit-identical to parsing the file itself."""
    from wbc_quadruped_dob_amd.sharding import pack_model, unpack_model, model_vector_len
    import wbc_quadruped_dob_amd as W
    flat = W.Model.from_urdf(W.SYNTHETIC_URDF).flat()
    vec = pack_model(flat)
    assert len(vec) == model_vector_len(int(flat["nb"]), len(flat["foot_body"])) and vec.nbytes < 4096
    again = unpack_model(vec)
    assert all(np.array_equal(np.asarray(again[k]), np.asarray(flat[k])) for k in again if k != "nb")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    same, dims, dm = q.get(timeout=100)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert same == 1 and dims == (13, 19, 18, 12, 4) and dm < 1e-12


def _bench_line_worker(rank, world, port, q):
    """One rank of the rehearsal: bench.py's OWN N > 1 functions (measure_job -> timed_blocks / long_blocks_of, gather_leg, contract_fields, job_fields,
    leg_fields) over gloo, with a stand-in tick that sleeps a rank-dependent time and writes a recognisable tau."""
    import json
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        import bench
        n, steps, warmup = 96, 4, 2
        tick_s = 0.8e-3 * (1.0 + 0.25 * rank)            # the slowest rank (world - 1) sets the job's time
        tau = torch.zeros((12, n), dtype=torch.float64)
        calls = [0]

        def step():
            calls[0] += 1
            time.sleep(tick_s)
            tau.fill_(float(rank + 1))
            return {"tau": tau}
        nosync = lambda: None
        blocks, elapsed, long_blocks = bench.measure_job(step, steps, n, world, dist, torch, np, device="cpu", sync=nosync)

        def make_tick(view):
            def st():
                time.sleep(tick_s)
                view.fill_(float(rank + 1))
                return {"tau": view}
            return st
        gather = bench.gather_leg(make_tick, dist, steps, n, world, rank, 12 * n * 8, torch.float64, torch, device="cpu", sync=nosync)
        line = bench.contract_fields(steps, warmup, n, world, elapsed, "f64", "rehearsal: stand-in ticks", True)
        line.update(bench.job_fields(blocks, steps, elapsed, long_blocks, gather, world, dist))
        k_leg, bl_leg, el_leg = bench.long_blocks_of(step, 3, dist, torch, np, device="cpu", sync=nosync)
        line["scale_config3"] = bench.leg_fields("rehearsal leg", k_leg, n, world, el_leg, bl_leg, gather, "f32")
        q.put((rank, json.dumps(line), blocks, elapsed, tick_s, calls[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_rank_bench_line_rehearsal():
    """bench.py's N > 1 aggregation had only ever executed with world = 1.  Eight gloo ranks drive the very functions the driver's
    `--gpus 8` run goes through: every rank arrives at the SAME blocks (max over ranks), `value` = steps x n x 8 / the slowest rank's block, the long
    blocks and the gather leg agree with it, and rank 0's line is valid JSON of the contract's shape."""
    import json
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_line_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    lines = [json.loads(g[1]) for g in got]
    blocks0, elapsed0 = got[0][2], got[0][3]
    slowest = max(g[4] for g in got)
    n, steps = 96, 4
    for (rank, _, blocks, elapsed, tick_s, ncalls), line in zip(got, lines):
        assert blocks == blocks0 and elapsed == elapsed0                        # every rank took the slowest rank's clock: same decisions everywhere
        assert ncalls == got[0][5]                                              # ... and ran the same number of ticks (agree_on_steps)
        assert line["n_gpus"] == world and line["rccl_ranks"] == world and line["steps"] == steps and line["scaling"] == "weak" and line["higher_is_better"] is True
        assert abs(line["value"] - steps * n * world / elapsed0) < 1e-6 * line["value"]
        assert abs(line["ms_per_step"] - elapsed0 / steps * 1e3) < 1e-9
        assert line["timing"]["blocks"] == len(blocks0) >= 7 and abs(line["timing"]["block_ms_median"] - elapsed0 * 1e3) < 1e-9
    line = lines[0]
    # the job's time is the SLOWEST rank's: a block of 4 ticks lasts at least 4 x its sleep, and not much more than that
    assert steps * slowest <= elapsed0 < steps * slowest * 1.6 + 2e-3
    lb = line["value_long_blocks"]
    assert lb["steps_per_block"] >= steps and lb["block_ms_median"] >= 5.0 * 0.9 and 0.6 < lb["value"] / line["value"] < 1.6
    g = line["with_tau_allgather"]
    assert g["value_is"] in ("eager_serial", "eager_overlapped") and 0.3 < g["value"] / line["value"] < 1.3 and "value" in g["eager_serial"]
    leg = line["scale_config3"]
    assert leg["rccl_ranks"] == world and abs(leg["value"] - leg["steps_per_block"] * n * world / (leg["ms_per_step"] * 1e-3 * leg["steps_per_block"])) < 1e-6 * leg["value"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
    assert line["config"]["batch_per_gpu"] == n and "x8" in line["config"]["parallelism"]

"""Batch sharding of the hot path across the GPUs of one node.

The path is embarrassingly parallel over robot states: each rank owns a contiguous slice of the batch and
runs the same kernels on it; there is NO data-path collective.  Two optional collectives exist for
consumers that need them: an all-gather of the torques (one consumer wants every tau) and an all-reduce
of a few status counters.  On the GPU box the backend is RCCL ("nccl") over xGMI; the CPU tests run the
same code over gloo with an injected compute function.

The reference has no distributed path (single robot, single process; /root/reference/README.md:58-60).
"""
import numpy as np


def shard_range(n_total, world, rank):
    """Contiguous, balanced slices: the first (n_total % world) ranks get one extra state."""
    if world < 1 or not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


class ShardedBatch:
    """Owns this rank's slice of a component-major batch [ncomp, n_total] and the optional collectives.

    step_fn(local_inputs: dict) -> dict(tau=[nj, n_local], f=..., status=[n_local], ...) is the per-rank
    compute: Solver.step on the GPU box.  (Tests inject a CPU stand-in; the product never does.)
    """

    def __init__(self, step_fn, n_total, dist=None):
        self.step_fn = step_fn
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.n_total = n_total
        self.start, self.count = shard_range(n_total, self.world, self.rank)

    def local_slice(self, x):
        """slice a full [ncomp, n_total] (or [n_total]) array/tensor down to this rank's columns"""
        return x[..., self.start:self.start + self.count]

    def step(self, local_inputs):
        return self.step_fn(local_inputs)

    def gather(self, x_local):
        """all-gather of a [ncomp, n_local] tensor into [ncomp, n_total] on every rank (ragged-safe)."""
        import torch
        if self.dist is None:
            return x_local
        ncomp = x_local.shape[0]
        counts = [shard_range(self.n_total, self.world, r)[1] for r in range(self.world)]
        mx = max(counts)
        pad = torch.zeros((ncomp, mx), dtype=x_local.dtype, device=x_local.device)
        pad[:, :self.count] = x_local
        bufs = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(bufs, pad)
        return torch.cat([b[:, :c] for b, c in zip(bufs, counts)], dim=1)

    def status_counts(self, status_local, iters_local=None):
        """all-reduce of [n_ok, n_iter_limit, n_infeasible, sum_iters, max_iters] over ranks."""
        import torch
        st = status_local.to(torch.int64)
        it = iters_local.to(torch.int64) if iters_local is not None else torch.zeros_like(st)
        sums = torch.stack([(st == 0).sum(), (st == 1).sum(), (st == 2).sum(), it.sum()])
        mx = it.max().reshape(1) if it.numel() else torch.zeros(1, dtype=torch.int64, device=st.device)
        if self.dist is not None:
            self.dist.all_reduce(sums, op=self.dist.ReduceOp.SUM)
            self.dist.all_reduce(mx, op=self.dist.ReduceOp.MAX)
        return dict(ok=int(sums[0]), iter_limit=int(sums[1]), infeasible=int(sums[2]), iters_sum=int(sums[3]),
                    iters_max=int(mx[0]))


_FLAT_F64 = ("Rt", "rt", "axis", "mass", "com", "Ic", "foot_off", "gravity")


def pack_model(flat):
    """The flat robot model (Model.flat(): < 4 kB) as ONE float64 vector: [nb, nf, parent(nb), foot_body(nf), Rt, rt, axis, mass, com, Ic,
    foot_off, gravity] -- the integers are exact in float64."""
    nb, nf = int(flat["nb"]), len(flat["foot_body"])
    parts = [np.array([nb, nf], np.float64), np.asarray(flat["parent"], np.float64), np.asarray(flat["foot_body"], np.float64)]
    parts += [np.asarray(flat[k], np.float64).ravel() for k in _FLAT_F64]
    return np.concatenate(parts)


def unpack_model(vec):
    vec = np.asarray(vec, np.float64)
    nb, nf = int(vec[0]), int(vec[1])
    shapes = dict(Rt=(nb, 9), rt=(nb, 3), axis=(nb, 3), mass=(nb,), com=(nb, 3), Ic=(nb, 6), foot_off=(nf, 3), gravity=(3,))
    o = dict(nb=nb, parent=vec[2:2 + nb].astype(np.int32), foot_body=vec[2 + nb:2 + nb + nf].astype(np.int32))
    at = 2 + nb + nf
    for k in _FLAT_F64:
        n = int(np.prod(shapes[k]))
        o[k] = vec[at:at + n].reshape(shapes[k]).copy()
        at += n
    assert at == len(vec)
    return o


def model_vector_len(nb, nf):
    return 2 + nb + nf + nb * (9 + 3 + 3 + 1 + 3 + 6) + nf * 3 + 3


def broadcast_model(flat, dist, device=None, src=0):
    """the model constants are read ONCE (rank `src` parses the URDF) and broadcast; the other ranks build their model from
    the received flat arrays (Model.from_flat / wbc_model_from_flat) instead of each opening the file.  flat: Model.flat() on rank src,
    ignored elsewhere.  Returns the flat dict on every rank (bit-identical: one float64 vector, two collectives -- its length, then it)."""
    import torch
    if dist is None:
        return flat
    mine = dist.get_rank() == src
    vec = pack_model(flat) if mine else None
    n = torch.tensor([len(vec) if mine else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    buf = torch.from_numpy(vec).to(device) if mine else torch.empty(int(n.item()), dtype=torch.float64, device=device)
    dist.broadcast(buf, src=src)
    return unpack_model(buf.cpu().numpy())


def agree_on_steps(k, dist, device=None):
    """Every rank derives its block length from its OWN clock; legs that follow each step with a collective (the all-gather below)
    must run the same number of steps on every rank or they hang.  Returns the maximum of `k` over the ranks (k itself without a
    process group)."""
    if dist is None:
        return int(k)
    import torch
    t = torch.tensor([int(k)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def gather_buffers(world, rank, rows, n, dtype, device, count=2):
    """Buffers for the IN-PLACE all-gather: `count` flat [world * rows, n] tensors and, for each, the view of this rank's block.  A tick
    that writes its torques straight into that view (Solver.step(out={"tau": view})) makes the collective in place: the local block
    is not copied again, and a one-rank communicator has nothing left to do."""
    import torch
    flats = [torch.zeros((world * rows, n), dtype=dtype, device=device) for _ in range(count)]
    views = [f.view(world, rows, n)[rank] for f in flats]
    return flats, views


def timed_steps_with_gather(step_fn, get_tau, dist, steps, sync=lambda: None, flat=None):
    """SURVEY.md 8e "report steps/s with and without the all-gather": runs `steps` ticks where every tick is followed by
    an all-gather of this rank's tau ([nj, n_local], equal n_local on every rank) into a preallocated
    [world, nj, n_local] buffer.  Returns (max-over-ranks seconds, gathered buffer).  bench.py calls it on the GPU box
    (RCCL); tests/test_sharding_gloo.py runs the same function over gloo."""
    import time
    import torch
    world = dist.get_world_size()
    tau = get_tau(step_fn())
    # output = the ranks' blocks concatenated along dim 0 (the one layout both RCCL and gloo accept); `flat` given: the caller's buffer
    # (in place when tau is this rank's block of it, gather_buffers)
    if flat is None:
        flat = torch.empty((world * tau.shape[0],) + tuple(tau.shape[1:]), dtype=tau.dtype, device=tau.device)
    gathered = flat.view((world,) + tuple(tau.shape))
    dist.all_gather_into_tensor(flat, tau.contiguous())   # warm the communicator
    sync()
    dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        tau = get_tau(step_fn())
        dist.all_gather_into_tensor(flat, tau.contiguous())
    sync()
    dt = time.perf_counter() - t0   # read before the closing barrier: the MAX over ranks below is the job's time
    dist.barrier()
    el = torch.tensor([dt], dtype=torch.float64, device=tau.device)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item()), gathered


def timed_steps_with_overlapped_gather(step_fns, get_tau, dist, steps, sync=lambda: None, flats=None):
    """The consumer-side all-gather OFF the tick's critical path: tau is double-buffered -- step_fns = (even, odd), two ticks over the
    same inputs that write their torques into DIFFERENT buffers -- and the gather of tick k runs on a side stream while tick k + 1
    computes; tick k + 2, which overwrites tick k's buffer, waits (stream-ordered, not on the host) for that gather only.
    Order of operations per tick k, buffer b = k & 1:   wait(gather k-2)  ->  tick k writes tau_b  ->  gather k of tau_b into all_b (async).
    Returns (max-over-ranks seconds, [all_0, all_1]): all_b[r] = rank r's tau of the last tick that used buffer b.
    CPU tensors (the gloo tests) take the same path without streams: async_op works, wait() blocks the host."""
    import time
    import torch
    world = dist.get_world_size()
    taus = [get_tau(f()) for f in step_fns]
    assert taus[0].data_ptr() != taus[1].data_ptr(), "the two ticks must write their torques into different buffers"
    cuda = taus[0].is_cuda
    flat = flats if flats is not None else [torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in taus]
    gathered = [fl.view((world,) + tuple(t.shape)) for fl, t in zip(flat, taus)]
    side = torch.cuda.Stream(device=taus[0].device) if cuda else None
    evs = [torch.cuda.Event() for _ in range(2)] if cuda else None
    works = [None, None]

    def gather(b, tau):
        #  the collective reads tau on the SIDE stream: a `.contiguous()` temporary made on the compute stream would be handed back to the
        # caching allocator when this function returns and could be reused while the side stream still reads it -- tau must be the tick's own
        # (contiguous, long-lived) output buffer, and the allocator is told which other stream uses it
        assert tau.is_contiguous(), "the overlapped gather needs the tick's own contiguous tau buffer"
        if cuda:
            tau.record_stream(side)
            evs[b].record()                          # tau_b is complete on the compute stream ...
            side.wait_event(evs[b])                  # ... before the side stream reads it
            with torch.cuda.stream(side):
                works[b] = dist.all_gather_into_tensor(flat[b], tau, async_op=True)
        else:
            works[b] = dist.all_gather_into_tensor(flat[b], tau, async_op=True)

    for b in (0, 1):                                 # warm the communicator and both paths
        gather(b, taus[b])
    for b in (0, 1):
        works[b].wait()
    sync()
    dist.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(steps):
        b = k & 1
        if works[b] is not None:
            works[b].wait()                          # (GPU: the CURRENT stream waits for gather k - 2; the host does not)
        tau = get_tau(step_fns[b]())
        gather(b, tau)
    for b in (0, 1):
        if works[b] is not None:
            works[b].wait()
    sync()
    dt = time.perf_counter() - t0
    dist.barrier()
    el = torch.tensor([dt], dtype=torch.float64, device=taus[0].device)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item()), gathered


def graph_steps_with_gather(step_fns, get_tau, dist, steps, replays=20, overlapped=True, gather=True, flats=None):
    """The same double-buffered tick + gather sequence as timed_steps_with_overlapped_gather (overlapped = False: the gather on the tick's
    own stream behind every tick), CAPTURED ONCE as a hipGraph of `steps` ticks and replayed: the host issues one graph launch per
    `steps` ticks, so what is timed is what the DEVICE needs for K ticks with their K gathers -- at 4 096 states a tick lasts 14 us,
    less than the Python call of one torch.distributed collective.  gather = False: the same graph without the collectives (the
    baseline the other two are compared with).  GPU tensors only.  Returns (max-over-ranks seconds per replay,
    [all_0, all_1]); raises when the stack refuses to capture the collective (the caller falls back to the eager forms)."""
    import time
    import torch
    world = dist.get_world_size()
    taus = [get_tau(f()) for f in step_fns]
    assert taus[0].is_cuda and taus[0].data_ptr() != taus[1].data_ptr()
    flat = flats if flats is not None else [torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in taus]
    gathered = [fl.view((world,) + tuple(t.shape)) for fl, t in zip(flat, taus)]
    for b in (0, 1):
        dist.all_gather_into_tensor(flat[b], taus[b])          # communicator and buffers warm before the capture
    torch.cuda.synchronize()
    main = torch.cuda.Stream(device=taus[0].device)
    side = torch.cuda.Stream(device=taus[0].device)
    main.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        # (thread_local: the process group's watchdog thread polls events of earlier collectives while this thread captures; in the
        #  default "global" mode that foreign call invalidates the capture and the watchdog aborts the process)
        with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
            works = [None, None]
            for k in range(steps):
                b = k & 1
                if works[b] is not None:
                    works[b].wait()                              # tick k overwrites tau_b: behind gather k - 2
                tau = get_tau(step_fns[b]())
                if not gather:
                    continue
                assert tau.is_contiguous()                       # (no temporaries across streams: see timed_steps_with_overlapped_gather)
                if overlapped:
                    tau.record_stream(side)
                    ev = torch.cuda.Event()
                    ev.record()
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        works[b] = dist.all_gather_into_tensor(flat[b], tau, async_op=True)
                else:
                    dist.all_gather_into_tensor(flat[b], tau)
            for b in (0, 1):
                if works[b] is not None:
                    works[b].wait()
            main.wait_stream(side)                               # the side stream joins before the capture ends
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / replays
    dist.barrier()
    el = torch.tensor([dt], dtype=torch.float64, device=taus[0].device)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item()), gathered

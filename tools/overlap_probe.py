"""Feasibility probe: does the QP stage of one part of a large batch overlap with the dynamics sweep of another part when the two run on
different streams?  K solvers, each with its own contiguous slice of the batch and its own torch stream, ticking freely (no
synchronisation between them) against ONE solver over the whole batch.   usage: python tools/overlap_probe.py [N] [K] [config] [dtype]"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dtype = sys.argv[4] if len(sys.argv) > 4 else "f64"
td = torch.float64 if dtype == "f64" else torch.float32
m = W.Model.from_urdf(W.SYNTHETIC_URDF)
obs = 0 if cfg == 2 else 1
P = synth.default_params(observer_order=obs, dtype=dtype)
B = synth.make_batch(cfg, N, m.total_mass)
def make(lo, hi, stream):
    n = hi - lo
    s = W.Solver(m, W.Params.from_dict(P, dtype), dtype=dtype, device=0, max_batch=n)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a[lo:hi].T)).to(td).cuda()
    inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
    mask = torch.from_numpy(B["mask"][lo:hi].copy()).cuda()
    integ = rr = None
    if obs:
        integ = s.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone(); rr = torch.zeros_like(integ)
    out = {}
    def tick():
        with torch.cuda.stream(stream):
            return s.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"], inp["f_prev"], integ, rr, out=out, want_mats=True)
    out.update(tick())
    return tick
def bench(ticks, reps=40):
    for t in ticks: t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for t in ticks: t()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
one = make(0, N, torch.cuda.current_stream())
t1 = bench([one])
del one
bounds = [N * k // K for k in range(K + 1)]
streams = [torch.cuda.Stream() for _ in range(K)]
parts = [make(bounds[k], bounds[k + 1], streams[k]) for k in range(K)]
tk = bench(parts)
same = [make(bounds[k], bounds[k + 1], torch.cuda.current_stream()) for k in range(K)]
ts = bench(same)
print("N %d cfg %d %s: one solver %.1f us = %.0f M steps/s | %d slices on %d streams %.1f us = %.0f M | %d slices on one stream %.1f us = %.0f M"
      % (N, cfg, dtype, t1 * 1e6, N / t1 / 1e6, K, K, tk * 1e6, N / tk / 1e6, K, ts * 1e6, N / ts / 1e6))

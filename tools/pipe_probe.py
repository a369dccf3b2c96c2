"""Diagnostic: does a large tick gain from running as TWO half-batches on two streams, the second half a phase behind the first (its HBM-bound sweep under the
first half's latency-bound QP kernels)?   python tools/pipe_probe.py [n] [config] [dtype] [ticks]
Prints ms per tick of the whole batch: one solver; two halves on one stream; two halves on two streams started together; two halves with stream 2 a half tick late."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
K = int(sys.argv[4]) if len(sys.argv) > 4 else 30
obs = 1 if cfg >= 3 else 0
m = W.Model.from_urdf(W.SYNTHETIC_URDF)
P = synth.default_params(observer_order=obs, dtype=dtype)
td = torch.float32 if dtype == "f32" else torch.float64


def make(nn, rank):
    s = W.Solver(m, W.Params.from_dict(P), dtype=dtype, max_batch=nn)
    B = synth.make_batch(cfg, nn, m.total_mass, rank=rank)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
    inp = [dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu")] + [torch.from_numpy(B["mask"]).cuda()]
    extra = []
    if obs:
        ig = s.dynamics(inp[0], inp[1], want=("p",))["p"]
        extra = [dev(B["tau_prev"]), dev(B["f_prev"]), ig, torch.zeros_like(ig)]
    run, out = s.prepare_step(*inp, *extra, want_mats=True)
    return s, run, out


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best / K * 1e3


s0, run0, _ = make(n, 0)
sa, runa, _ = make(n // 2, 1)
sb, runb, _ = make(n // 2, 2)
for r in (run0, runa, runb):
    for _ in range(3): r()
torch.cuda.synchronize()
print("n", n, "config", cfg, dtype, "plan whole", s0.plan_tick(n), "half", sa.plan_tick(n // 2))
print("one solver                      %.4f ms per tick" % timed(lambda: [run0() for _ in range(K)]))
print("two halves, one stream          %.4f" % timed(lambda: [(runa(), runb()) for _ in range(K)]))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two(offset_cycles):
    def f():
        with torch.cuda.stream(s2):
            if offset_cycles: torch.cuda._sleep(offset_cycles)
        for _ in range(K):
            with torch.cuda.stream(s1): runa()
            with torch.cuda.stream(s2): runb()
    return f


def two_sync():
    # every tick forks and joins: stream 2 starts the tick a phase late (sleep), both join before the next tick
    ev0, ev1, ev2 = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
    def f():
        for _ in range(K):
            ev0.record(torch.cuda.current_stream())
            s1.wait_event(ev0); s2.wait_event(ev0)
            with torch.cuda.stream(s1): runa(); ev1.record(s1)
            with torch.cuda.stream(s2):
                if OFF: torch.cuda._sleep(OFF)
                runb(); ev2.record(s2)
            torch.cuda.current_stream().wait_event(ev1); torch.cuda.current_stream().wait_event(ev2)
    return f


print("two halves, two streams         %.4f" % timed(two(0)))
for us in (40, 80, 120, 160):
    print("  stream 2 %3d us late (free-running) %.4f" % (us, timed(two(int(us * 2100)))))
for us in (0, 10, 20, 40, 80):
    OFF = int(us * 2100)
    print("  fork / join every tick, stream 2 %3d us late %.4f" % (us, timed(two_sync())))

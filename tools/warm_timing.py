import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
model = W.Model.from_urdf(W.SYNTHETIC_URDF)
for cfg, obs, dtype, n in ((4,1,"f32",32768),(4,1,"f32",16384),(3,1,"f64",32768),(2,0,"f64",32768),(4,0,"f32",32768)):
    td = torch.float64 if dtype == "f64" else torch.float32
    P = synth.default_params(observer_order=obs, dtype=dtype)
    B = synth.make_batch(cfg, n, model.total_mass, rank=1)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
    for warm in (False, True):
        solver = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=0, max_batch=n)
        inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
        mask = torch.from_numpy(B["mask"]).cuda()
        integ = rr = None
        if obs:
            integ = solver.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone(); rr = torch.zeros_like(integ)
        tick, out = solver.prepare_step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"], inp["f_prev"], integ, rr, want_mats=True, warm=warm)
        for _ in range(10): tick()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(100): tick()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("   wall per tick %.1f us (host enqueue %.1f us)" % ((t2 - t0) / 100 * 1e6, (t1 - t0) / 100 * 1e6))
        solver.enable_timing(1)
        for _ in range(20): tick()
        torch.cuda.synchronize()
        tm = solver.collect_timing()
        print(cfg, dtype, obs, n, "warm" if warm else "cold", {k: round(v*1e3/max(1,tm[k[:-3]+"_launches"]),1) for k, v in tm.items() if k.endswith("_ms") and v > 0}, "iters", float(out["iters"].double().mean()), solver.plan_tick(n, warm=warm))

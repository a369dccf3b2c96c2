#!/usr/bin/env python3
"""Back-to-back (un-synchronised) warm ticks with the states drifting, every tick timed by its dispatch events (read one by one through the timing ring is
not possible: collect returns sums) -- so: K ticks per collect, K = 1 but WITHOUT a device synchronisation before the next enqueue is impossible either;
instead time blocks of 10 ticks by wall clock and report the slowest blocks."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
model = W.Model.from_urdf(W.SYNTHETIC_URDF)
for cfg, obs, dtype, n in ((4, 1, "f32", 8192), (4, 1, "f32", 32768), (4, 0, "f32", 32768), (3, 1, "f64", 32768)):
    td = torch.float64 if dtype == "f64" else torch.float32
    P = synth.default_params(observer_order=obs, dtype=dtype)
    B = synth.make_batch(cfg, n, model.total_mass, rank=1)
    B["w_des"][:, 0:2] += np.random.default_rng(1).uniform(-40, 40, (n, 2))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
    for warm in (False, True):
        solver = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=0, max_batch=n)
        inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
        mask = torch.from_numpy(B["mask"]).cuda()
        integ = rr = None
        if obs:
            integ = solver.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone(); rr = torch.zeros_like(integ)
        dq = 1e-3 * torch.randn((12, n), dtype=td, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
        dw = 0.2 * torch.randn((6, n), dtype=td, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
        tick, out = solver.prepare_step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"], inp["f_prev"], integ, rr, want_mats=True, warm=warm)
        for _ in range(10): tick()
        torch.cuda.synchronize()
        blocks = []
        i = 0
        for b in range(40):
            t0 = time.perf_counter()
            for _ in range(10):
                inp["q"][7:] += dq if i % 2 == 0 else -dq
                inp["w_des"] += dw if i % 2 == 0 else -dw
                tick(); i += 1
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / 10 * 1e6)
            if obs and b % 10 == 9:
                pass
        blocks = np.array(blocks)
        rmax = float(rr.abs().max()) if obs else 0.0
        print(cfg, dtype, obs, n, "warm" if warm else "cold", "us/tick per 10-tick block: median %.1f  max %.1f  first five %s  worst at block %d | iters mean %.2f max %d status!=0 %d  max|rhat| %.3g" % (
            np.median(blocks), blocks.max(), np.round(blocks[:5], 1).tolist(), int(blocks.argmax()), float(out["iters"].double().mean()), int(out["iters"].max()), int((out["status"] != 0).sum()), rmax))

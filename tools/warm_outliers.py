#!/usr/bin/env python3
"""Per-launch durations of the warm fused tick (dispatch events of EVERY tick, collected one by one): looks for slow outliers."""
import sys
import numpy as np
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
model = W.Model.from_urdf(W.SYNTHETIC_URDF)
for cfg, obs, dtype, n in ((4, 1, "f32", 8192), (4, 1, "f32", 4096), (3, 1, "f64", 8192), (4, 1, "f32", 32768)):
    td = torch.float64 if dtype == "f64" else torch.float32
    P = synth.default_params(observer_order=obs, dtype=dtype)
    B = synth.make_batch(cfg, n, model.total_mass, rank=1)
    B["w_des"][:, 0:2] += np.random.default_rng(1).uniform(-40, 40, (n, 2))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
    for warm in (False, True):
        solver = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=0, max_batch=n)
        inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
        mask = torch.from_numpy(B["mask"]).cuda()
        integ = solver.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone(); rr = torch.zeros_like(integ)
        dq = 1e-3 * torch.randn((12, n), dtype=td, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
        tick, out = solver.prepare_step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"], inp["f_prev"], integ, rr, want_mats=True, warm=warm)
        for _ in range(10): tick()
        torch.cuda.synchronize()
        solver.enable_timing(1)
        ts, its, sts = [], [], []
        for i in range(300):
            inp["q"][7:] += dq if i % 2 == 0 else -dq
            tick()
            torch.cuda.synchronize()
            tm = solver.collect_timing()
            ts.append(sum(v for k, v in tm.items() if k.endswith("_ms")) * 1e3)
            its.append(float(out["iters"].double().max())); sts.append(int((out["status"] != 0).sum()))
        ts = np.array(ts)
        bad = np.flatnonzero(ts > 3 * np.median(ts))
        print(cfg, dtype, n, "warm" if warm else "cold", "median %.1f us  p99 %.1f  max %.1f  outliers(>3x) %d at %s  max iters seen %.0f  nonzero status %d" % (
            np.median(ts), np.percentile(ts, 99), ts.max(), len(bad), bad[:10].tolist(), max(its), sum(sts)), "iters at outliers", [its[i] for i in bad[:10]])

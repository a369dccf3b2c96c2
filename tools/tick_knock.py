#!/usr/bin/env python3
"""What bounds the one-launch tick at the bench batch?  The same tick with the QP's iteration limit lowered (results are then NOT solutions: status != 0 for
the states that needed more) and with / without the M, h, Jc outputs: if the kernel does not get shorter when the hard QPs are cut off, they are not its bound.
usage: tools/tick_knock.py [config] [n]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import wbc_quadruped_dob_amd as W  # noqa: E402
from wbc_quadruped_dob_amd import synth  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
obs = 0 if cfg == 2 else 1
model = W.Model.from_urdf(W.SYNTHETIC_URDF)
B = synth.make_batch(cfg, n, model.total_mass, rank=0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(torch.float64).cuda()
for mats in (True, False):
    for max_iter in (100, 8, 6, 4, 2, 1):
        P = synth.default_params(observer_order=obs, dtype="f64")
        P["max_iter"] = max_iter
        solver = W.Solver(model, W.Params.from_dict(P, "f64"), dtype="f64", device=0, max_batch=n)
        inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
        mask = torch.from_numpy(B["mask"]).cuda()
        integ = rr = None
        if obs:
            integ = solver.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone()
            rr = torch.zeros_like(integ)
        tick, out = solver.prepare_step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"], inp["f_prev"],
                                        integ, rr, want_mats=mats)
        for _ in range(30):
            tick()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(300):
                tick()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
        st = out["status"].cpu().numpy()
        it = out["iters"].cpu().numpy()
        print("cfg%d n%d mats=%d max_iter=%3d : %.2f us per tick   unsolved %5.1f %%   iters mean %.2f max %d" % (cfg, n, mats, max_iter, best, 100.0 * (st != 0).mean(), it.mean(), it.max()))

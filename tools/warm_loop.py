#!/usr/bin/env python3
"""A closed loop of DEPENDENT ticks through wbc_step_batch_warm (tools/warm_loop.py [N ...]): the same batch ticked K times while the
states drift a little between ticks (joint angles and the commanded wrench), cold start (wbc_step_batch) against warm start from the
previous tick's active set (WARM_LOOP_LANE=1: the warm tick twice, with the per-lane pair switched off -- warm one-wavefront kernel below the tile threshold, reporting cold tiles above -- and forced through the per-lane kernel).  Prints the wall time per tick of the whole loop (the two elementwise drift kernels included, and timed alone beside
it), the tick's kernels by their own dispatch events, and the mean QP iterations."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096, 8192, 32768]
    model = W.Model.from_urdf(W.SYNTHETIC_URDF)
    for cfg, obs, dtype in ((2, 0, "f64"), (3, 1, "f64"), (4, 1, "f32")):
        td = torch.float64 if dtype == "f64" else torch.float32
        for n in sizes:
            P = synth.default_params(observer_order=obs, dtype=dtype)
            B = synth.make_batch(cfg, n, model.total_mass, rank=1)
            B["w_des"][:, 0:2] += np.random.default_rng(1).uniform(-40, 40, (n, 2))
            dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
            res = {}
            variants = [(False, False, {}), (True, True, {})]
            if os.environ.get("WARM_LOOP_LANE") == "1":
                variants = [(False, False, {}), ("warm16", True, {"qp_lane": -1}), ("warmlane", True, {"qp_lane": 1}),
                            ("warm16f", True, {"qp_lane": -1, "qp_tile": -1})]     # (the last: the warm one-wavefront kernel at every size)
            if os.environ.get("WARM_LOOP_LANE") == "2":   # the planner's warm tick against the two-launch warm plans (tile_tick = -1) and the forced per-lane pair
                variants = [(False, False, {}), ("warm", True, {}), ("warm2l", True, {"tile_tick": -1}), ("warmlane", True, {"qp_lane": 1})]
            for tag, warm, opts in variants:
                solver = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=0, max_batch=n, options=opts)
                inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
                mask = torch.from_numpy(B["mask"]).cuda()
                integ = rr = None
                if obs:
                    integ = solver.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone()
                    rr = torch.zeros_like(integ)
                dq = 1e-3 * torch.randn((12, n), dtype=td, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
                dw = 0.2 * torch.randn((6, n), dtype=td, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
                tick, out = solver.prepare_step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"],
                                                inp["f_prev"], integ, rr, want_mats=True, warm=warm)

                dqn, dwn = -dq, -dw            # (no temporaries inside the loop: with a fresh 1.5 MB tensor per iteration the HOST spends 200-700 us
                qj = inp["q"][7:]              #  per iteration in the un-synchronised loop, which has nothing to do with the tick)

                def loop(k):
                    for i in range(k):
                        qj.add_(dq if i % 2 == 0 else dqn)                 # the robots move between ticks
                        inp["w_des"].add_(dw if i % 2 == 0 else dwn)
                        tick()
                loop(20)
                torch.cuda.synchronize()
                K = 200
                el = None
                for _ in range(3):             # best of three blocks: the box's host stalls now and then for 10-80 ms, which is as long as a whole block
                    t0 = time.perf_counter()
                    loop(K)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    el = dt if el is None else min(el, dt)
                # the drift alone (two elementwise kernels per tick)
                t0 = time.perf_counter()
                for i in range(K):
                    qj.add_(dq if i % 2 == 0 else dqn)
                    inp["w_des"].add_(dw if i % 2 == 0 else dwn)
                torch.cuda.synchronize()
                el0 = time.perf_counter() - t0
                solver.enable_timing(7)          # the tick's kernels by their own dispatch events, every 7-th tick
                loop(210)
                torch.cuda.synchronize()
                tm = solver.collect_timing()
                kern = {k[:-3]: round(v * 1e3 / max(1, tm[k[:-3] + "_launches"]), 1) for k, v in tm.items() if k.endswith("_ms") and v > 0}
                solver.enable_timing(0)
                if solver.plan_tick(n, warm=warm)["qp"] == 2:
                    kern["handed_over"] = solver.qp_handover()
                res[tag] = (el / K * 1e6, el0 / K * 1e6, float(out["iters"].double().mean()), float((out["status"] == 0).double().mean()), kern)
            if os.environ.get("WARM_LOOP_LANE") == "2":
                pw = solver_plan = W.plan_tick(n, dtype, obs, warm=True)
                print("cfg%d %s obs%d n=%6d wall us/tick: cold %6.2f  warm (planner: fused=%d qp=%d qp_warm=%d) %6.2f  warm, tile_tick = -1 %6.2f  warm per-lane pair forced %6.2f | iters %.2f / %.2f / %.2f / %.2f" % (
                    cfg, dtype, obs, n, res[False][0], pw["fused"], pw["qp"], pw["qp_warm"], res["warm"][0], res["warm2l"][0], res["warmlane"][0], res[False][2], res["warm"][2], res["warm2l"][2], res["warmlane"][2]), flush=True)
                continue
            if len(variants) == 4:
                print("cfg%d %s obs%d n=%6d wall us/tick: cold %6.2f  warm, per-lane pair off %6.2f  warm per-lane %6.2f  warm one-wavefront kernel forced %6.2f | iters %.2f / %.2f / %.2f / %.2f | kernels cold %s  warm16 %s  warmlane %s  warm16f %s" % (
                    cfg, dtype, obs, n, res[False][0], res["warm16"][0], res["warmlane"][0], res["warm16f"][0], res[False][2], res["warm16"][2], res["warmlane"][2], res["warm16f"][2],
                    res[False][4], res["warm16"][4], res["warmlane"][4], res["warm16f"][4]), flush=True)
                continue
            print("cfg%d %s obs%d n=%6d: wall per tick incl. the drift kernels (drift alone %.1f us): cold %6.2f us (iters %.2f)  warm %6.2f us (iters %.2f)  ok %.4f / %.4f  %s  "
                  "tick kernels by dispatch events: cold %s warm %s" % (
                      cfg, dtype, obs, n, res[False][1], res[False][0], res[False][2], res[True][0], res[True][2], res[False][3], res[True][3],
                      "fused" if W.plan_tick(n, dtype, obs, warm=True)["fused"] else "two-kernel", res[False][4], res[True][4]))


if __name__ == "__main__":
    main()

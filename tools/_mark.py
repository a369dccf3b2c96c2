import sys, numpy as np, torch
sys.path.insert(0, ".")
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
m = W.Model.from_urdf(W.SYNTHETIC_URDF)
for cfg, obs in ((2, 0), (3, 1)):
    n = 65536
    P = synth.default_params(observer_order=obs)
    B = synth.make_batch(cfg, n, m.total_mass, rank=0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(torch.float64).cuda()
    inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
    mask = torch.from_numpy(B["mask"]).cuda()
    res = {}
    for tag, opt in (("lane", {"fused_max": 0, "qp_lane": 1}), ("dense", {"fused_max": 0, "qp_lane": -1})):
        s = W.Solver(m, W.Params.from_dict(P), max_batch=n, options=opt)
        integ = rr = None
        if obs:
            integ = s.dynamics(inp["q"], inp["v"], want=("p",))["p"].clone(); rr = torch.zeros_like(integ)
        out = s.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"], inp["f_prev"], integ, rr)
        torch.cuda.synchronize()
        res[tag] = out["iters"].cpu().numpy()
    il, idn = res["lane"], res["dense"]
    handed = il < 100
    print("cfg", cfg, "handed over %d of %d (%.1f%%)" % (handed.sum(), n, 100 * handed.mean()))
    print("  dense GI iterations, all states:   mean %.2f  max %d  hist %s" % (idn.mean(), idn.max(), np.bincount(idn, minlength=16)[:16]))
    print("  dense GI iterations, handed-over:  mean %.2f  max %d  hist %s" % (idn[handed].mean(), idn[handed].max(), np.bincount(idn[handed], minlength=16)[:16]))
    print("  dense GI iterations, lane-solved:  mean %.2f  max %d  hist %s" % (idn[~handed].mean(), idn[~handed].max(), np.bincount(idn[~handed], minlength=16)[:16]))
    print("  newton iterations of lane-solved: hist", np.bincount(il[~handed] - 100, minlength=7)[:7])
    mk = B["mask"].astype(np.int64) & 15
    pc = np.array([bin(x).count("1") for x in range(16)])[mk]
    for c in range(5):
        sel = pc == c
        if sel.sum(): print("  stance feet %d: %6d states, handed over %.1f%%, dense iters mean %.2f" % (c, sel.sum(), 100 * handed[sel].mean(), idn[sel].mean()))

"""Diagnostic: per-phase cycle stamps of the fused sweep kernel (needs -DWBC_SWEEP_STAMP: WBC_LIB=.../libwbc_hip_sstamp.so)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
m = W.Model.from_urdf(W.SYNTHETIC_URDF)
names = ["issue state loads", "stage table+barrier", "early stores", "base (loads arrive)", "forward sweep", "return sweep",
         "leg outputs", "base block", "step prologue", "drain stores"]
for n in (4096, 262144):
    P = synth.default_params(); s = W.Solver(m, W.Params.from_dict(P), max_batch=n)
    B = synth.make_batch(2, n, m.total_mass)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
    inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu")}
    mask = torch.from_numpy(B["mask"]).cuda()
    for _ in range(3):
        out = s.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, want_mats=True)
    torch.cuda.synchronize()
    pf = out["pf"].cpu().numpy().T  # [N, 12]
    vals = np.concatenate([pf[:, 0:9], pf[:, 9:10]], 1)
    print("N", n, {k: int(v) for k, v in zip(names, vals.mean(0))}, "sum", int(vals.mean(0).sum()))

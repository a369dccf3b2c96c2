import sys, numpy as np, torch
sys.path.insert(0, ".")
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
m = W.Model.from_urdf(W.SYNTHETIC_URDF)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
P = synth.default_params(observer_order=0)
s = W.Solver(m, W.Params.from_dict(P), max_batch=n, options={"fused_max": 0})
B = synth.make_batch(2, n, m.total_mass, rank=0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(torch.float64).cuda()
inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev")}
mask = torch.from_numpy(B["mask"]).cuda()
def run(out, reps=30):
    for _ in range(8): s.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"], inp["f_prev"], None, None, out=out)
    torch.cuda.synchronize(); s.enable_timing(1)
    for _ in range(reps): s.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, inp["tau_prev"], inp["f_prev"], None, None, out=out)
    torch.cuda.synchronize(); t = s.collect_timing(); s.enable_timing(0)
    return t["dyn_ms"] * 1e3 / max(1, t["dyn_launches"])
full = {k: s.empty(r, n) for k, r in (("M", 171), ("h", 18), ("Jc", 216), ("pf", 12))}
nopf = {k: full[k] for k in ("M", "h", "Jc")}
for rep in range(2):
    print("tick sweep with pf %.1f us, without pf %.1f us" % (run(dict(full)), run(dict(nopf))))

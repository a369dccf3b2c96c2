"""fp64 tile tick (forced) against the oracle on a few sizes: status equal, tau / f to 1e-9 (diagnostic; the committed tests are in tests/test_gpu_round6.py)"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
from oracle import oracle_py, urdf_model
from tests.util import relerr, to_dev, to_host
m = W.Model.from_urdf(W.SYNTHETIC_URDF)
orc = oracle_py.Oracle(urdf_model.load_urdf(W.SYNTHETIC_URDF))
for n in (37, 4099, 6144, 12288, 28672, 30001):
    P = synth.default_params()
    s = W.Solver(m, W.Params.from_dict(P), max_batch=n, options={"tile_tick": 1, "fused_max": 0})
    B = synth.make_batch(2, n, m.total_mass, rank=5)
    B["w_des"][: n // 2, 0:2] += np.random.default_rng(9).uniform(-60, 60, (n // 2, 2))
    dv = lambda k: to_dev(B[k], torch, torch.float64)
    out = s.step(dv("q"), dv("v"), dv("w_des"), dv("vdot_des"), dv("normals"), dv("mu"), torch.from_numpy(B["mask"]).cuda(), want_mats=True)
    torch.cuda.synchronize()
    ref = orc.step(P, B["q"], B["v"], B["w_des"], B["vdot_des"], B["normals"], B["mu"], B["mask"], B["tau_prev"], B["f_prev"], None, None, nthreads=8)
    d = orc.dynamics(B["q"], B["v"], nthreads=8)
    st = out["status"].cpu().numpy()
    print(n, s.plan_tick(n), "status equal", np.array_equal(st, ref["status"]), "tau", relerr(to_host(out["tau"]), ref["tau"]), "f", relerr(to_host(out["f"]), ref["f"]),
          "M", relerr(to_host(out["M"]), d["M"]), "Jc", relerr(to_host(out["Jc"]), d["Jc"]), "iters mean", out["iters"].float().mean().item(), "ref", ref["iters"].mean())

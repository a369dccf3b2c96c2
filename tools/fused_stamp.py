"""Diagnostic: timeline of the fused tick's wavefront roles (needs the -DWBC_FUSED_STAMP build:
   make -C wbc_quadruped_dob_amd/csrc -j8 LIBDIR=../lib_fstamp EXTRA=-DWBC_FUSED_STAMP;  WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_fstamp/libwbc_hip.so python tools/fused_stamp.py).
In that build the `pf` output carries 100 MHz timestamps (wall_clock64) per workgroup instead of foot positions."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
m = W.Model.from_urdf(W.SYNTHETIC_URDF)
names = ["QP0 entry", "QP0 tables staged, inputs issued", "QP0 lever arms seen", "QP0 factor done", "QP0 rhat seen", "QP0 iterations done",
         "QP0 tau_partial seen", "rnea: lever arms out", "rnea: done", "mass_jac: done", "obs done / QP3 iterations done", "QP0 stores issued"]
for cfg, obs in ((2, 0), (3, 1)):
    n = 4096
    P = synth.default_params(observer_order=obs); s = W.Solver(m, W.Params.from_dict(P), max_batch=n)
    B = synth.make_batch(cfg, n, m.total_mass)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
    inp = [dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu")]
    mask = torch.from_numpy(B["mask"]).cuda()
    extra = []
    if obs:
        ig = s.dynamics(inp[0], inp[1], want=("p",))["p"]
        extra = [dev(B["tau_prev"]), dev(B["f_prev"]), ig, torch.zeros_like(ig)]
    for _ in range(5):
        out = s.step(*inp, mask, *extra, want_mats=True)
    torch.cuda.synchronize()
    st = out["pf"].cpu().numpy()[:, ::16]            # [12 slots, workgroups]
    t0 = st[0]
    rel = (st - t0[None, :]) * 10.0                   # ns since the workgroup's first stamp
    print("config", cfg, "observer", obs, " kernel-wide span (first entry -> last stamp): %.2f us" % ((st.max() - st[0].min()) * 1e-2))
    print("  workgroup entry spread: %.2f us" % ((t0.max() - t0.min()) * 1e-2))
    for i, nm in enumerate(names):
        print("  %-36s median %+7.2f us   p90 %+7.2f us" % (nm, np.median(rel[i]) * 1e-3, np.percentile(rel[i], 90) * 1e-3))
    # QP wavefront 0 of a workgroup solves states 16 w .. 16 w + 3: time from "b seen" to "iterations done" against its trips
    it = out["iters"].cpu().numpy().reshape(-1, 16)[:, :4].max(1)
    dt = (rel[5] - rel[4]) * 1e-3
    A = np.stack([np.ones_like(dt), it.astype(float)], 1)
    coef = np.linalg.lstsq(A, dt, rcond=None)[0]
    print("  QP wavefront 0: (iterations done - b seen) = %.2f us + %.3f us x trips  (trips: mean %.2f, max %d; slowest workgroup %.2f us after its entry)"
          % (coef[0], coef[1], it.mean(), it.max(), rel[11].max() * 1e-3))
    for k in sorted(set(it.tolist())):
        sel = it == k
        print("     trips %2d: %4d workgroups, median %.2f us" % (k, sel.sum(), np.median(dt[sel])))

# ---- persistent rollout: the LAST tick of a horizon-20 rollout of 1 024 robots (observer on)
rnames = ["tick start (after barrier B)", "QP0 inputs issued", "QP0 lever arms seen", "QP0 factor done", "QP0 rhat seen", "QP0 iterations done",
          "QP0 tau_partial / rhat_joint seen", "integrator: barrier A passed", "integrator: update done", "integrator: M, Jc seen (obs joint rows done)",
          "observer base rows done", "QP0 stores issued"]
import os
os.environ["WBC_ROLLOUT_SPW"] = "16"   # the stamp columns are indexed by 16-state workgroups
n, H = 1024, 20
P = synth.default_params(observer_order=1); s = W.Solver(m, W.Params.from_dict(P), max_batch=n)
B = synth.make_batch(5, n, m.total_mass)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
inp = [dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu")]
mask = torch.from_numpy(B["mask"]).cuda()
ig = s.dynamics(inp[0], inp[1], want=("p",))["p"]; r = torch.zeros_like(ig)
out = s.step(*inp, mask, dev(B["tau_prev"]), dev(B["f_prev"]), ig, r, want_mats=True)
q0, v0 = inp[0].clone(), inp[1].clone()
for _ in range(3):
    inp[0].copy_(q0); inp[1].copy_(v0)
    s.rollout(H, inp[0], inp[1], inp[2], inp[3], inp[4], inp[5], mask, out, ig, r)
torch.cuda.synchronize()
st = out["pf"].cpu().numpy()[:, ::16]
rel = (st - st[0][None, :]) * 10.0
print("rollout, 1024 robots, horizon 20, observer on: last tick")
for i, nm in enumerate(rnames):
    print("  %-46s median %+7.2f us   p90 %+7.2f us" % (nm, np.median(rel[i]) * 1e-3, np.percentile(rel[i], 90) * 1e-3))

"""Diagnostic: role timeline of the LAST tick of a persistent rollout (needs the stamp build with the rollout's own slots:
   make -C wbc_quadruped_dob_amd/csrc -j8 LIBDIR=../lib_rstamp EXTRA="-DWBC_FUSED_STAMP -DWBC_RO_STAMP_ALT"
   WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_rstamp/libwbc_hip.so python tools/rollout_stamp.py [n_robots] [spw]).
In that build the `pf` output carries 100 MHz timestamps (wall_clock64), one column per workgroup."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spw = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.environ["WBC_ROLLOUT_SPW"] = str(spw)
import wbc_quadruped_dob_amd as W  # noqa: E402
from wbc_quadruped_dob_amd import synth  # noqa: E402

m = W.Model.from_urdf(W.SYNTHETIC_URDF)
# (four-wavefront layout of the 4-state workgroups, WBC_RO_MERGE: slots 2 and 9 are not stamped -- there is no integrator wavefront; slot 7 = phase 2 starts on
#  the QP's wavefront; slot 10 = the observer wavefront is through BOTH sets of rows and its stores.  Unstamped slots print n/a.)
names = ["tick start (QP0, after the tick barrier)", "mass_jac: image published", "integrator wavefront at barrier A (8-wavefront layout)", "rnea: done (tau_partial out)",
         "QP0 rhat seen", "QP0 iterations done", "QP0 tau_partial / rhat_joint seen", "integrator phase 2 starts (8 wavefronts: barrier A passed)",
         "integrator: update done (stores issued)", "integrator: M, Jc seen, joint rows done (8-wavefront layout)", "observer wavefront done (8 wavefronts: base rows)",
         "QP0 stores issued"]
H = 20
P = synth.default_params(observer_order=1)
s = W.Solver(m, W.Params.from_dict(P), max_batch=n)
B = synth.make_batch(5, n, m.total_mass)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
inp = [dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu")]
mask = torch.from_numpy(B["mask"]).cuda()
ig = s.dynamics(inp[0], inp[1], want=("p",))["p"]
r = torch.zeros_like(ig)
out = s.step(*inp, mask, dev(B["tau_prev"]), dev(B["f_prev"]), ig, r, want_mats=True)
q0, v0 = inp[0].clone(), inp[1].clone()
for _ in range(3):
    inp[0].copy_(q0); inp[1].copy_(v0)
    s.rollout(H, inp[0], inp[1], inp[2], inp[3], inp[4], inp[5], mask, out, ig, r)
torch.cuda.synchronize()
st = out["pf"].cpu().numpy()[:, ::spw]
rel = (st - st[0][None, :]) * 10.0
print("rollout, %d robots, %d states per workgroup, horizon %d, observer on, warm: last tick, us after the tick's start (median / p90 over %d workgroups)" % (n, spw, H, st.shape[1]))
def line(nm, col):
    if not np.all(np.abs(col) < 1e9):   # a slot this build does not stamp: whatever the buffer held
        print("  %-66s n/a" % nm)
    else:
        print("  %-66s median %+7.2f   p90 %+7.2f" % (nm, np.median(col) * 1e-3, np.percentile(col, 90) * 1e-3))


for i, nm in enumerate(names):
    line(nm, rel[i])

if spw < 16:
    inames = ["integrator entry (M, Jc flag seen)", "image read requested", "leg block inverted", "Schur complement summed over the legs", "Cholesky done",
              "barrier A passed", "tau, f, h arrived: leg right-hand side", "base right-hand side summed", "triangular solves done", "joint rows stored",
              "quaternion advanced", "base rows stored"]
    sti = out["pf"].cpu().numpy()[:, 1::spw]
    reli = (sti - st[0][None, :]) * 10.0
    print("the integrator wavefront's own phases (same tick, same origin):")
    for i, nm in enumerate(inames):
        line(nm, reli[i])

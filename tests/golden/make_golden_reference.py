#!/usr/bin/env python3
"""Generates tests/golden/golden_reference_v1.npz with the INDEPENDENT numpy implementation (oracle/crosscheck_np.py):
the CoM reference generator on 12 states, and 5 states x 4 ticks of planner-in-the-loop rollouts.
No reference-side vectors exist (the planner's source is absent): PARITY UNPINNED.
Run from the repo root:  python tests/golden/make_golden_reference.py"""
import os
import sys
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import urdf_model, crosscheck_np as X  # noqa: E402
from wbc_quadruped_dob_amd import synth  # noqa: E402
import wbc_quadruped_dob_amd as W  # noqa: E402


def main():
    flat = urdf_model.load_urdf(W.SYNTHETIC_URDF)
    npm = X.NPModel(flat)
    G = synth.default_ref_params()
    out = {}
    # ---- reference generator alone
    n = 12
    B = synth.make_batch(4, n, float(flat["mass"].sum()), rank=31)
    plan = synth.make_plan(B, rank=31)
    t = 0.013
    res = [X.reference(npm, G, B["q"][s], B["v"][s], plan[s], t) for s in range(n)]
    out.update(ref_in_q=B["q"], ref_in_v=B["v"], ref_in_plan=plan, ref_in_t=np.array(t),
               ref_out_w_des=np.array([r[0] for r in res]), ref_out_vdot_des=np.array([r[1] for r in res]),
               ref_out_com=np.array([r[2] for r in res]))
    # ---- planner in the loop
    n, horizon = 5, 4
    Bc = synth.make_batch(3, 12, float(flat["mass"].sum()), rank=32)      # candidates
    planc = synth.make_plan(Bc, rank=32)
    P = synth.default_params(observer_order=1)
    tau_c = np.zeros((12, 18))
    tau_c[:, 0:3] = np.random.default_rng(5).uniform(-40, 40, (12, 3))
    qs, vs, taus, coms, igs, rs, keep = [], [], [], [], [], [], []
    for s in range(12):
        ig0 = npm.mass_matrix(Bc["q"][s]) @ Bc["v"][s]
        try:   # the numpy PRIMAL active-set solver can stall on degenerate vertices; such candidates are skipped
            q, v, tt, cc, ig, r = X.rollout_tracking(npm, P, G, horizon, Bc["q"][s], Bc["v"][s], planc[s],
                                                     Bc["normals"][s].reshape(-1, 3), Bc["mu"][s], int(Bc["mask"][s]), tau_c[s],
                                                     ig0, np.zeros(18))
        except RuntimeError:
            continue
        qs.append(q); vs.append(v); taus.append(tt); coms.append(cc); igs.append(ig); rs.append(r); keep.append(s)
        if len(keep) == n:
            break
    assert len(keep) == n, keep
    B = {k: Bc[k][keep] for k in ("q", "v", "normals", "mu", "mask")}
    plan, tau_ext = planc[keep], tau_c[keep]
    integ0 = np.array([npm.mass_matrix(B["q"][s]) @ B["v"][s] for s in range(n)])
    for k in ("q", "v", "normals", "mu", "mask"):
        out["trk_in_" + k] = B[k]
    out.update(trk_in_plan=plan, trk_in_tau_ext=tau_ext, trk_in_integ0=integ0, trk_horizon=np.array(horizon),
               trk_out_q=np.array(qs), trk_out_v=np.array(vs), trk_out_tau_traj=np.array(taus), trk_out_com_traj=np.array(coms),
               trk_out_integ=np.array(igs), trk_out_r=np.array(rs))
    path = os.path.join(ROOT, "tests", "golden", "golden_reference_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

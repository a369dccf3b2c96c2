#!/usr/bin/env python3
"""Generates tests/golden/golden_rollout_v1.npz with the INDEPENDENT numpy implementation (oracle/crosscheck_np.py):
6 states x 4 ticks of {control step, forward dynamics with the planned GRFs by numpy's LU solve, semi-implicit Euler}.
No reference-side vectors exist for this (the reference's simulator is Gazebo): PARITY UNPINNED.
Run from the repo root:  python tests/golden/make_golden_rollout.py"""
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
    n, horizon = 6, 4
    out = {"horizon": np.array(horizon)}
    for name, cfg, obs in (("r_obs0", 2, 0), ("r_obs1", 3, 1)):
        B = synth.make_batch(cfg, n, float(flat["mass"].sum()), rank=21)
        P = synth.default_params(observer_order=obs)
        tau_ext = np.zeros((n, 18))
        tau_ext[:, 0:3] = np.random.default_rng(4).uniform(-40, 40, (n, 3))
        integ0 = np.array([npm.mass_matrix(B["q"][s]) @ B["v"][s] for s in range(n)])
        qs, vs, taus, igs, rs = [], [], [], [], []
        for s in range(n):
            q, v, tt, ig, r = X.rollout(npm, P, horizon, B["q"][s], B["v"][s], B["w_des"][s], B["vdot_des"][s],
                                        B["normals"][s].reshape(-1, 3), B["mu"][s], int(B["mask"][s]), tau_ext[s], integ0[s],
                                        np.zeros(18))
            qs.append(q); vs.append(v); taus.append(tt); igs.append(ig); rs.append(r)
        for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "mask"):
            out[f"{name}_in_{k}"] = B[k]
        out[f"{name}_in_tau_ext"] = tau_ext
        out[f"{name}_in_integ0"] = integ0
        out[f"{name}_observer_order"] = np.array(obs)
        out[f"{name}_out_q"] = np.array(qs)
        out[f"{name}_out_v"] = np.array(vs)
        out[f"{name}_out_tau_traj"] = np.array(taus)
        out[f"{name}_out_integ"] = np.array(igs)
        out[f"{name}_out_r"] = np.array(rs)
        print(name, "done")
    path = os.path.join(ROOT, "tests", "golden", "golden_rollout_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

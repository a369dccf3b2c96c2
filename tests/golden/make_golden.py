#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz with the INDEPENDENT numpy implementation
(oracle/crosscheck_np.py), never with the C++ oracle or the HIP path.

There are no reference-side golden vectors to commit: the reference holds no tests, fixtures or
recorded outputs and its controller source is absent, so PARITY IS UNPINNED
against the reference; these fixtures pin the oracle and the HIP path against a second,
algorithmically different implementation instead.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import urdf_model, crosscheck_np as X  # noqa: E402
from wbc_quadruped_dob_amd import synth  # noqa: E402

URDF = os.path.join(ROOT, "wbc_quadruped_dob_amd", "assets", "synthetic_quadruped.urdf")


def main():
    flat = urdf_model.load_urdf(URDF)
    npm = X.NPModel(flat)
    nv = npm.nv
    out = {}
    for k in ("parent", "Rt", "rt", "axis", "mass", "com", "Ic", "foot_body", "foot_off", "gravity"):
        out["model_" + k] = flat[k]
    out["model_nb"] = np.array(flat["nb"])
    cases = [("cfg2", 2, 0, 24), ("cfg3", 3, 1, 27), ("cfg4o2", 4, 2, 18)]
    for name, cfg, obs, n in cases:
        B = synth.make_batch(cfg, n, float(flat["mass"].sum()), rank=7)
        P = synth.default_params(nv=nv, observer_order=obs)
        # one hand-made flight state (mask 0) and one single-foot state per case
        B["mask"][0] = 0
        B["mask"][1] = 0b0100
        integ0 = np.zeros((n, nv))
        for s in range(n):
            integ0[s] = npm.mass_matrix(B["q"][s]) @ B["v"][s] + 0.01 * np.sin(np.arange(nv) + s)
        r0 = 0.5 * np.cos(np.arange(nv)[None, :] * 0.7 + np.arange(n)[:, None])
        keys = ("M", "h", "Jc", "pf", "f", "tau", "integ", "r")
        acc = {k: [] for k in keys}
        acc["beta"], acc["p"], acc["kkt"] = [], [], []
        for s in range(n):
            o = X.step(npm, P, B["q"][s], B["v"][s], B["w_des"][s], B["vdot_des"][s], B["normals"][s].reshape(-1, 3),
                       B["mu"][s], int(B["mask"][s]), B["tau_prev"][s], B["f_prev"][s], integ0[s], r0[s])
            iu = np.triu_indices(nv)
            acc["M"].append(o["M"][iu])
            acc["h"].append(o["h"])
            acc["Jc"].append(o["Jc"].reshape(-1))
            acc["pf"].append(o["pf"].reshape(-1))
            acc["f"].append(o["f"])
            acc["tau"].append(o["tau"])
            acc["integ"].append(o["integ"])
            acc["r"].append(o["r"])
            acc["beta"].append(npm.beta(B["q"][s], B["v"][s]))
            acc["p"].append(o["M"] @ B["v"][s])
            acc["kkt"].append(max(o["kkt"]) if "kkt" in o else 0.0)
        assert max(acc["kkt"]) < 1e-9, max(acc["kkt"])
        for k, val in B.items():
            out[f"{name}_in_{k}"] = val
        out[f"{name}_in_integ0"] = integ0
        out[f"{name}_in_r0"] = r0
        out[f"{name}_observer_order"] = np.array(obs)
        for k, val in acc.items():
            out[f"{name}_out_{k}"] = np.array(val)
        print(name, "n", n, "max kkt", max(acc["kkt"]))
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""How far apart are the rollouts of 4-state and 16-state workgroups (same arithmetic per state, another distribution)?  Prints per array the largest
absolute difference and where it is; H ticks each.  usage: tools/spw_diff.py [obs] [n] [H ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))


def main():
    import torch
    from wbc_quadruped_dob_amd import synth
    import test_gpu_parity as tp
    import wbc_quadruped_dob_amd as W
    from oracle import oracle_py, urdf_model
    obs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 777
    Hs = [int(x) for x in sys.argv[3:]] or [1, 2, 9]
    model = W.Model.from_urdf(W.SYNTHETIC_URDF)
    orc = oracle_py.Oracle(urdf_model.load_urdf(W.SYNTHETIC_URDF))
    B = synth.make_batch(4 if obs else 3, n, model.total_mass, rank=67)
    tau_ext = np.zeros((n, 18))
    tau_ext[:, 0:3] = B["push"]
    integ = orc.dynamics(B["q"], B["v"], nthreads=8)["p"] if obs else None
    for H in Hs:
        res = {}
        for spw in (4, 16):
            solver, P = tp._solver(model, obs=obs, max_batch=n, options={"rollout_spw": spw})
            res[spw] = tp._gpu_rollout(torch, solver, P, H, B, tau_ext, None if integ is None else integ.copy(), np.zeros((n, 18)) if obs else None)
        for k in res[4]:
            a, b = res[4][k].astype(np.float64), res[16][k].astype(np.float64)
            d = np.abs(a - b)
            i = np.unravel_index(np.argmax(d), d.shape)
            print("H %2d %-9s max|d| %.3e at %s (value %.6g)  differing entries %d of %d" % (H, k, d.max(), i, a[i], int((d > 0).sum()), d.size))


if __name__ == "__main__":
    main()

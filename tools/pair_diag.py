"""fused_pair_kernel against fused_tick_kernel on one batch: the largest difference per output and where it sits (diagnostic for tests/test_gpu_round6.py)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
from tests.test_gpu_parity import _run_step, _solver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
model = W.Model.from_urdf(W.SYNTHETIC_URDF)
B = synth.make_batch(2, n, model.total_mass, rank=31)
res = {}
for tag, opt in (("pair", {"fused_pair": 1}), ("one", {"fused_pair": -1, "fused_max": 65536}), ("one2", {"fused_pair": -1, "fused_max": 65536})):
    solver, P = _solver(model, dtype="f64", obs=0, max_batch=n, options=opt)
    print(tag, solver.plan_tick(n)["fused"])
    res[tag] = _run_step(torch, solver, B, "f64", want_mats=True)
for x, y in (("pair", "one"), ("one", "one2")):
    a, b = res[x], res[y]
    for k in ("M", "h", "Jc", "pf", "tau", "f", "status", "iters"):
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        i = np.unravel_index(np.argmax(d), d.shape)
        print("%s vs %s  %-6s max |diff| %.3e at %s (values %r / %r), entries that differ: %d of %d, states: %d" % (x, y, k, d.max(), i, a[k][i], b[k][i], int((d > 0).sum()), d.size, int((d.reshape(n, -1) > 0).any(axis=1).sum())))
        if (d > 0).any():
            st = np.nonzero((d.reshape(n, -1) > 0).any(axis=1))[0]
            print("    states:", st[:12], "... columns:", np.nonzero((d.reshape(n, -1) > 0).any(axis=0))[0][:20])

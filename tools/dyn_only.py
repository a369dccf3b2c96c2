#!/usr/bin/env python3
"""The dynamics stage exactly as SURVEY.md 8(d) defines it -- q, v -> M, h, Jc as its own kernel (wbc_dynamics_batch) -- run 12
times at one batch size, for the rocprofv3 / PMC passes (tools/pmc_profile.sh).  usage: dyn_only.py N {f64|f32}"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import wbc_quadruped_dob_amd as W   # noqa: E402
from wbc_quadruped_dob_amd import synth   # noqa: E402

n, dtype = int(sys.argv[1]), sys.argv[2]
td = torch.float64 if dtype == "f64" else torch.float32
model = W.Model.from_urdf(W.SYNTHETIC_URDF)
solver = W.Solver(model, W.Params.from_dict(synth.default_params(dtype=dtype), dtype), dtype=dtype, device=0, max_batch=n)
B = synth.make_batch(2, n, model.total_mass)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
q, v = dev(B["q"]), dev(B["v"])
out = solver.dynamics(q, v, want=("M", "h", "Jc"))
for _ in range(12):
    solver.dynamics(q, v, want=("M", "h", "Jc"), out=out)
torch.cuda.synchronize()
print("ok", n, dtype)

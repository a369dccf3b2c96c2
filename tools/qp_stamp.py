"""Diagnostic: per-segment cycle shares of the QP kernel (needs the -DWBC_QP_STAMP build:
   make -C wbc_quadruped_dob_amd/csrc -j8 LIBDIR=../lib_qstamp EXTRA=-DWBC_QP_STAMP; WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_qstamp/libwbc_hip.so python tools/qp_stamp.py).
The stamps come from the one-wavefront-workgroup kernel (two-kernel tick, no tiles)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
m = W.Model.from_urdf(W.SYNTHETIC_URDF)
names = ["setup + x0 + first search", "candidate fetch, v, b, y = Ginv b, z.n", "directions, ratio test, speculative full step + next search", "-", "-", "commit + add (selects)", "drop path", "-"]   # structured body (qp_struct16.hip.hpp); the stamps themselves (s_memtime + lgkmcnt wait) cost ~100 cycles each
for n in (4096, 262144):
    P = synth.default_params(); s = W.Solver(m, W.Params.from_dict(P), max_batch=n, options={"fused_max": 0, "qp_tile": -1})
    B = synth.make_batch(2, n, m.total_mass)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
    inp = {k: dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu")}
    mask = torch.from_numpy(B["mask"]).cuda()
    for _ in range(3):
        out = s.step(inp["q"], inp["v"], inp["w_des"], inp["vdot_des"], inp["normals"], inp["mu"], mask, want_mats=True)
    torch.cuda.synchronize()
    it = out["iters"].cpu().numpy().astype(np.int64).reshape(-1, 4)
    vals = np.stack([it[:, g] & 0xFFFF if h == 0 else (it[:, g] >> 16) & 0x7FFF for g in range(4) for h in range(2)], 1) * 16
    tot = vals.sum(1)
    real = (vals[:, 0] > 0)
    print("N", n, "mean cycles per wave:", {k: int(v) for k, v in zip(names, vals.mean(0))}, "total", int(tot.mean()))

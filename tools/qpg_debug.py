"""Debug aid for qp_general_kernel: first step-2 pass at which the kernel's iterate leaves the oracle's (same max_iter on both)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import wbc_quadruped_dob_amd as W
from oracle import oracle_py as O
from tests.util import random_problem
n, m, meq, N = (int(a) for a in sys.argv[1:5])
rng = np.random.default_rng(100 * n + m + meq)
probs = [random_problem(rng, n, m, meq) for _ in range(N)]
H, g, C, d = (np.stack([p[k] for p in probs]) for k in range(4))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def gpu(mi):
    o = W.qp_dense_batch(dev(H), dev(g), dev(C), dev(d), meq=meq, max_iter=mi, tol=1e-10)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in o.items()}
full = gpu(400)
xr, lr, sr, ir = O.qp_general(H, g, C, d, meq, max_iter=400, tol=1e-10)
bad = np.nonzero((full["status"] != sr) | (np.abs(full["x"] - xr).max(1) > 1e-8))[0]
print("bad", len(bad), "of", N, "first", bad[:10], "oracle iters of bad", ir[bad[:10]], "gpu iters", full["iters"][bad[:10]], "gpu status", full["status"][bad[:10]])
if len(bad):
    b = bad[0]
    for mi in range(0, int(ir[b]) + 1):
        o = gpu(mi)
        xo, lo, so, io = O.qp_general(H[b], g[b], C[b], d[b], meq, max_iter=mi, tol=1e-10)
        print("max_iter", mi, "gpu st/it", o["status"][b], o["iters"][b], "oracle", so, io, "dx", np.abs(o["x"][b] - xo).max(), "dlam", np.abs(o["lam"][b] - lo).max())
        print("   gpu lam", np.round(o["lam"][b], 4)); print("   orc lam", np.round(lo, 4))

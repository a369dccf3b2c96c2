"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def unpack_M(Mp, nv=18):
    """packed upper [.., nv(nv+1)/2] -> full symmetric [.., nv, nv]"""
    Mp = np.asarray(Mp)
    out = np.zeros(Mp.shape[:-1] + (nv, nv), Mp.dtype)
    iu = np.triu_indices(nv)
    out[..., iu[0], iu[1]] = Mp
    out[..., iu[1], iu[0]] = Mp
    return out


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


def elementwise_excess(a, b, rtol=1e-6, atol_frac=1e-9):
    """The element-wise gate of BASELINE.json's "torques within 1e-6 rel": every entry must satisfy
        |a_i - b_i| <= rtol * |b_i| + atol_frac * max|b|
    (relerr() above divides by the LARGEST entry of the whole array, which says nothing about small torques).  Returns the largest
    ratio |a_i - b_i| / bound_i: <= 1 passes."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    bound = rtol * np.abs(b) + atol_frac * float(np.max(np.abs(b)))
    return float(np.max(np.abs(a - b) / np.maximum(bound, np.finfo(np.float64).tiny)))


def to_dev(x, torch, dtype):
    """row-per-state numpy [N, c] -> component-major device tensor [c, N]"""
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(x).T))
    if t.dtype.is_floating_point:
        t = t.to(dtype)
    return t.cuda().contiguous()


def to_host(t):
    """component-major device tensor [c, N] -> row-per-state numpy [N, c]"""
    return t.detach().cpu().numpy().T.copy()


def random_problem(rng, n, m, meq, cond=1e3):
    """A feasible strictly convex QP (H, g, C, d) with m rows, the first meq of them equalities: H = Q diag(1 .. cond) Q^T, rows of C
    random with a third of the inequalities tight or violated at the unconstrained minimum, feasibility guaranteed by construction
    around a random point (d = C x_feas - nonnegative slack; equality rows hold exactly there; needs meq <= n)."""
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    H = (Q * np.geomspace(1.0, cond, n)) @ Q.T
    H = 0.5 * (H + H.T)
    xf = rng.normal(size=n)
    C = rng.normal(size=(m, n))
    slack = np.abs(rng.normal(size=m)) * (rng.random(m) < 0.7)
    slack[:meq] = 0
    d = C @ xf - slack
    g = -H @ (xf + rng.normal(size=n) * 2.0)     # the unconstrained minimum sits away from the feasible point
    return H, g, C, d

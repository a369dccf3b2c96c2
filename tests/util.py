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


def to_dev(x, torch, dtype):
    """row-per-state numpy [N, c] -> component-major device tensor [c, N]"""
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(x).T))
    if t.dtype.is_floating_point:
        t = t.to(dtype)
    return t.cuda().contiguous()


def to_host(t):
    """component-major device tensor [c, N] -> row-per-state numpy [N, c]"""
    return t.detach().cpu().numpy().T.copy()

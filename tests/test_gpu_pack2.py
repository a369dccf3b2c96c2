"""fp32 dynamics sweep with TWO states per lane (dyn_sweep_kernel<float, MODE, BLOCK, 2>: packed v_pk_* arithmetic, a lane
owns states 2k, 2k+1) against the one-state-per-lane kernel and against the oracle, through the C-ABI.

The packed kernel evaluates the same formulas on pairs; sin / cos come from its own straight-line single-precision kernel,
so results agree with the unpacked kernel to fp32 rounding, not bit for bit.  Tolerances are relative to the largest
magnitude of each quantity: 2e-5 between the two fp32 kernels for the dynamics outputs (measured ~1e-6), the stated fp32
gate of 1e-3 against the fp32 oracle for torques and forces."""
import numpy as np
import pytest

from tests.util import relerr, to_dev, to_host
from wbc_quadruped_dob_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch


def _solver(gpu_model, obs, n, **opt):
    import wbc_quadruped_dob_amd as W
    P = synth.default_params(observer_order=obs, dtype="f32")
    return W.Solver(gpu_model, W.Params.from_dict(P, "f32"), dtype="f32", device=0, max_batch=n, options=opt), P


@pytest.mark.parametrize("n", [2, 30, 64, 66, 1000, 4098, 70000])   # ragged against the 32 states of a wavefront; 4-aligned and not
def test_packed_dynamics_vs_unpacked_and_oracle(torch_cuda, gpu_model, oracle, n):
    torch = torch_cuda
    B = synth.make_batch(4, n, gpu_model.total_mass, rank=n)
    q, v = to_dev(B["q"], torch, torch.float32), to_dev(B["v"], torch, torch.float32)
    want = ("M", "h", "Jc", "pf", "p", "beta")
    res = {}
    for pack in (1, -1):
        solver, _ = _solver(gpu_model, 0, n, f32_pack2=pack)
        # buffers poisoned first: every word of every output must be written by the kernel (structural zeros included)
        out = {k: torch.full((r, n), float("nan"), dtype=torch.float32, device="cuda")
               for k, r in (("M", 171), ("h", 18), ("Jc", 216), ("pf", 12), ("p", 18), ("beta", 18))}
        solver.dynamics(q, v, want=want, out=out)
        torch.cuda.synchronize()
        res[pack] = {k: to_host(out[k]) for k in want}
    ref = oracle.dynamics(B["q"], B["v"], nthreads=8)
    for k in want:
        assert np.isfinite(res[1][k]).all(), k
        assert relerr(res[1][k], res[-1][k]) < 2e-5, k
        assert relerr(res[1][k], ref[k]) < (2e-4 if k == "beta" else 5e-5), k


def _step(torch, solver, B, integ, r, want_mats):
    td = torch.float32
    dv = lambda k: to_dev(B[k], torch, td)
    mask = torch.from_numpy(np.ascontiguousarray(B["mask"])).to(torch.int32).cuda()
    ig = None if integ is None else to_dev(integ, torch, td)
    rr = None if r is None else to_dev(r, torch, td)
    out = solver.step(dv("q"), dv("v"), dv("w_des"), dv("vdot_des"), dv("normals"), dv("mu"), mask, dv("tau_prev"), dv("f_prev"),
                      ig, rr, want_mats=want_mats)
    torch.cuda.synchronize()
    res = {k: (to_host(x) if x.dim() == 2 else x.cpu().numpy()) for k, x in out.items()}
    if ig is not None:
        res["integ"], res["r"] = to_host(ig), to_host(rr)
    return res


@pytest.mark.parametrize("obs,n,split", [(0, 9000, -2), (1, 9002, -2), (2, 10000, -2), (1, 9000, 1), (1, 32768, -2)])
def test_packed_tick_vs_unpacked_tick_and_oracle(torch_cuda, gpu_model, oracle, obs, n, split):
    """Two-kernel ticks (fused_max = 0) whose front half is the packed sweep: observer off (SW_NOB), the all-in-one observer
    sweep (SW_OBS, orders 1 and 2) and the observer-free sweep behind the stand-alone observer kernel (obs_split_min = 1);
    the last case is configs[3]'s per-GPU batch."""
    torch = torch_cuda
    B = synth.make_batch(4, n, gpu_model.total_mass, rank=7)
    f32 = lambda a: a.astype(np.float32)
    integ = r = None
    if obs:
        integ = oracle.dynamics(B["q"], B["v"], nthreads=8)["p"] + 0.01
        r = 0.3 * np.sin(np.arange(n * 18).reshape(n, 18))
    got = {}
    for pack in (1, -1):
        solver, P = _solver(gpu_model, obs, n, f32_pack2=pack, fused_max=0, obs_split_min=split)
        got[pack] = _step(torch, solver, B, integ, r, True)
    ig32, r32 = (None, None) if not obs else (f32(integ), f32(r))
    ref = oracle.step(P, f32(B["q"]), f32(B["v"]), f32(B["w_des"]), f32(B["vdot_des"]), f32(B["normals"]), f32(B["mu"]), B["mask"],
                      f32(B["tau_prev"]), f32(B["f_prev"]), ig32, r32, nthreads=8)
    a, b = got[1], got[-1]
    same = (a["status"] == b["status"])
    assert same.mean() > 0.999
    ok = same & (a["status"] == 0) & (ref["status"] == 0)
    assert ok.mean() > 0.99
    for k in ("tau", "f"):
        assert relerr(a[k][ok], b[k][ok]) < 1e-3, k       # the fp32 gate between the two front halves
        assert relerr(a[k][ok], ref[k][ok]) < 1e-3, k     # ... and against the fp32 oracle
    for k in ("M", "h", "Jc", "pf"):
        assert relerr(a[k], b[k]) < 2e-5, k
    if obs:
        assert relerr(a["integ"], b["integ"]) < 2e-5
        assert relerr(a["r"], b["r"]) < 2e-3     # r = K (p - integ): a difference of nearly equal numbers times a gain of 50


def test_odd_batches_keep_the_unpacked_kernel(torch_cuda, gpu_model, oracle):
    """f32_pack2 = 1 on an odd batch: a lane's pair would straddle the end of every component row, so the launcher falls back."""
    torch = torch_cuda
    n = 9001
    B = synth.make_batch(2, n, gpu_model.total_mass, rank=3)
    q, v = to_dev(B["q"], torch, torch.float32), to_dev(B["v"], torch, torch.float32)
    a, _ = _solver(gpu_model, 0, n, f32_pack2=1)
    b, _ = _solver(gpu_model, 0, n, f32_pack2=-1)
    oa, ob = a.dynamics(q, v), b.dynamics(q, v)
    torch.cuda.synchronize()
    for k in ("M", "h", "Jc", "pf"):
        assert torch.equal(oa[k], ob[k]), k

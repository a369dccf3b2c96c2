"""GPU parity: the CoM reference generator (wbc_reference_batch) and planner-in-the-loop rollouts
(wbc_rollout_tracking_batch), SURVEY.md 8f-3 / 8f-4, through the C-ABI against the CPU oracle and the golden fixture.
PARITY UNPINNED against the reference itself (the planner's source is absent)."""
import os

import numpy as np
import pytest

from tests.util import relerr, to_dev, to_host
from wbc_quadruped_dob_amd import synth

pytestmark = pytest.mark.gpu
TIGHT64 = 1e-9


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch


def _solver(gpu_model, dtype="f64", obs=0, max_batch=4096, ref=True, options=None):
    import wbc_quadruped_dob_amd as W
    P = synth.default_params(observer_order=obs, dtype=dtype)
    s = W.Solver(gpu_model, W.Params.from_dict(P, dtype), dtype=dtype, device=0, max_batch=max_batch, options=options or {})
    G = synth.default_ref_params()
    if ref:
        s.set_ref_params(G)
    return s, P, G


@pytest.mark.parametrize("n", [1, 15, 17, 1000, 4096])
def test_reference_vs_oracle_ragged_sizes(torch_cuda, gpu_model, oracle, n):
    torch = torch_cuda
    solver, P, G = _solver(gpu_model, max_batch=n)
    B = synth.make_batch(4, n, gpu_model.total_mass, rank=41)
    plan = synth.make_plan(B, rank=41)
    ref = oracle.reference(G, B["q"], B["v"], plan, 0.017)
    dv = lambda a: to_dev(a, torch, torch.float64)
    got = solver.reference(dv(B["q"]), dv(B["v"]), dv(plan), 0.017, want_com=True)
    torch.cuda.synchronize()
    for k in ("w_des", "vdot_des", "com"):
        assert relerr(to_host(got[k]), ref[k]) < 1e-12, k


def test_reference_vs_golden(torch_cuda, gpu_model):
    torch = torch_cuda
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_reference_v1.npz")))
    n = g["ref_in_q"].shape[0]
    solver, P, G = _solver(gpu_model, max_batch=n)
    dv = lambda a: to_dev(a, torch, torch.float64)
    got = solver.reference(dv(g["ref_in_q"]), dv(g["ref_in_v"]), dv(g["ref_in_plan"]), float(g["ref_in_t"]), want_com=True)
    torch.cuda.synchronize()
    assert relerr(to_host(got["w_des"]), g["ref_out_w_des"]) < 1e-12
    assert relerr(to_host(got["vdot_des"]), g["ref_out_vdot_des"]) < 1e-12
    assert relerr(to_host(got["com"]), g["ref_out_com"]) < 1e-13


def test_reference_fp32(torch_cuda, gpu_model, oracle):
    torch = torch_cuda
    n = 2048
    solver, P, G = _solver(gpu_model, dtype="f32", max_batch=n)
    B = synth.make_batch(4, n, gpu_model.total_mass, rank=42)
    plan = synth.make_plan(B, rank=42)
    c32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    ref = oracle.reference(G, c32(B["q"]), c32(B["v"]), c32(plan), 0.0)
    dv = lambda a: to_dev(a, torch, torch.float32)
    got = solver.reference(dv(B["q"]), dv(B["v"]), dv(plan), 0.0, want_com=True)
    torch.cuda.synchronize()
    for k in ("w_des", "vdot_des", "com"):
        assert relerr(to_host(got[k]), ref[k]) < 2e-5, k      # fp32: a few ulps through the PD gains (<= 200)


def _gpu_tracking(torch, solver, H, B, plan, tau_ext, integ, r, want_com=True):
    td = torch.float64
    n = B["q"].shape[0]
    dv = lambda a: to_dev(a, torch, td)
    q, v = dv(B["q"]), dv(B["v"])
    mask = torch.from_numpy(np.ascontiguousarray(B["mask"])).to(torch.int32).cuda()
    out = dict(tau=torch.zeros((12, n), dtype=td, device="cuda"), f=torch.zeros((12, n), dtype=td, device="cuda"),
               status=torch.zeros(n, dtype=torch.int32, device="cuda"), iters=torch.zeros(n, dtype=torch.int32, device="cuda"),
               M=solver.empty(171, n), h=solver.empty(18, n), Jc=solver.empty(216, n), pf=solver.empty(12, n))
    ig = None if integ is None else dv(integ)
    rr = None if r is None else dv(r)
    traj = torch.zeros((H, 12, n), dtype=td, device="cuda")
    com = torch.zeros((H, 6, n), dtype=td, device="cuda") if want_com else None
    solver.rollout_tracking(H, q, v, dv(plan), dv(B["normals"]), dv(B["mu"]), mask, out, solver.empty(6, n), solver.empty(18, n),
                            ig, rr, None if tau_ext is None else dv(tau_ext), traj, com)
    torch.cuda.synchronize()
    res = dict(q=to_host(q), v=to_host(v), status=out["status"].cpu().numpy(), f=to_host(out["f"]),
               tau_traj=traj.cpu().numpy().transpose(2, 0, 1).copy())
    if com is not None:
        res["com_traj"] = com.cpu().numpy().transpose(2, 0, 1).copy()
    if ig is not None:
        res["integ"], res["r"] = to_host(ig), to_host(rr)
    return res


@pytest.mark.parametrize("cfg,obs,n,H,spw", [(3, 1, 1000, 20, 0), (4, 2, 257, 9, 0), (2, 0, 64, 20, 0), (3, 1, 301, 7, 16), (2, 0, 1100, 5, 0), (4, 2, 37, 3, 16)])
def test_tracking_rollout_vs_oracle(torch_cuda, gpu_model, oracle, cfg, obs, n, H, spw):
    """(spw = 16, or more than 1 024 rollouts: the 16-state workgroups of the persistent kernel -- planner on QP wavefront 0, references through LDS)"""
    torch = torch_cuda
    solver, P, G = _solver(gpu_model, obs=obs, max_batch=n, options={"rollout_spw": spw} if spw else None)
    B = synth.make_batch(cfg, n, gpu_model.total_mass, rank=43)
    plan = synth.make_plan(B, rank=43)
    tau_ext = np.zeros((n, 18)); tau_ext[:, 0:3] = B["push"]
    integ = oracle.dynamics(B["q"], B["v"], nthreads=8)["p"] if obs else None
    r = np.zeros((n, 18)) if obs else None
    q, v = B["q"].copy(), B["v"].copy()
    ig_ref = None if integ is None else integ.copy()
    r_ref = None if r is None else r.copy()
    ref = oracle.rollout_tracking(P, G, H, q, v, plan, B["normals"], B["mu"], B["mask"], tau_ext=tau_ext, integ=ig_ref,
                                  r=r_ref, want_traj=True, want_com=True, nthreads=8)
    got = _gpu_tracking(torch, solver, H, B, plan, tau_ext, integ, r)
    ok = (ref["status"] == 0) & (got["status"] == 0)
    assert ok.mean() > 0.99          # random far-from-plan states with PD gains may saturate a force box in rare rows
    assert relerr(got["q"][ok], q[ok]) < 1e-8 and relerr(got["v"][ok], v[ok]) < 1e-8
    assert relerr(got["tau_traj"][ok], ref["tau_traj"][ok]) < 1e-7
    assert relerr(got["tau_traj"][ok][:, 0], ref["tau_traj"][ok][:, 0]) < TIGHT64
    assert relerr(got["com_traj"][ok], ref["com_traj"][ok]) < 1e-9
    if obs:
        assert relerr(got["r"][ok], r_ref[ok]) < 1e-6


def test_persistent_tracking_equals_per_tick_launches(torch_cuda, gpu_model, oracle):
    """Planner-in-the-loop rollouts of small batches run as ONE launch (rollout_kernel<., TRACK>: the integrator wavefront
    runs the reference generator at the head of every tick); wbc_solver_options.rollout_persistent = 0 = {reference, fused tick, integrate}
    launches per tick.  Same device functions -> equal to rounding."""
    torch = torch_cuda
    n, H = 777, 12
    B = synth.make_batch(3, n, gpu_model.total_mass, rank=45)
    plan = synth.make_plan(B, rank=45)
    tau_ext = np.zeros((n, 18)); tau_ext[:, 0:3] = B["push"]
    integ = oracle.dynamics(B["q"], B["v"], nthreads=8)["p"]
    res = {}
    for tag, opt in (("persistent", {}), ("per_tick", {"rollout_persistent": 0})):
        solver, P, G = _solver(gpu_model, obs=1, max_batch=n, options=opt)
        res[tag] = _gpu_tracking(torch, solver, H, B, plan, tau_ext, integ.copy(), np.zeros((n, 18)))
    a, b = res["persistent"], res["per_tick"]
    assert np.array_equal(a["status"], b["status"])
    for k in ("q", "v", "tau_traj", "com_traj", "integ", "r"):   # (two implementations of the same maths: see test_persistent_rollout_equals_per_tick_launches)
        assert relerr(a[k], b[k]) < TIGHT64, k


def test_tracking_rollout_vs_golden(torch_cuda, gpu_model):
    torch = torch_cuda
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_reference_v1.npz")))
    n, H = g["trk_in_q"].shape[0], int(g["trk_horizon"])
    solver, P, G = _solver(gpu_model, obs=1, max_batch=n)
    B = {k: g["trk_in_" + k] for k in ("q", "v", "normals", "mu", "mask")}
    got = _gpu_tracking(torch, solver, H, B, g["trk_in_plan"], g["trk_in_tau_ext"], g["trk_in_integ0"].copy(), np.zeros((n, 18)))
    assert relerr(got["q"], g["trk_out_q"]) < 1e-10
    assert relerr(got["v"], g["trk_out_v"]) < 1e-9
    assert relerr(got["tau_traj"], g["trk_out_tau_traj"]) < 1e-8
    assert relerr(got["com_traj"], g["trk_out_com_traj"]) < 1e-10


def test_case_study_push_rejection_on_gpu(torch_cuda, gpu_model, oracle):
    """The reference's headline behaviour (README.md:11) end to end on the GPU path: planner -> observer -> GRF QP ->
    torque map -> forward dynamics -> integrator for 600 ticks under a constant 36 N push, 512 robots with different
    push directions.  Observer off: ~1 cm steady CoM offset; observer on: rejected, residual = push."""
    torch = torch_cuda
    n, H = 512, 600
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    push = np.zeros((n, 18)); push[:, 0] = 36 * np.cos(ang); push[:, 1] = 36 * np.sin(ang)
    final = {}
    for obs in (0, 1):
        solver, P, G = _solver(gpu_model, obs=obs, max_batch=n)
        q = np.zeros((n, 19)); q[:, 2] = 0.40; q[:, 6] = 1.0; q[:, 7:] = G["q_nom"]
        v = np.zeros((n, 18))
        ident = np.zeros((n, 12)); ident[:, 11] = 1.0
        com0 = oracle.reference(G, q, v, ident)["com"]
        plan = ident.copy(); plan[:, 0:3] = com0[:, 0:3]; plan[:, 3:6] = com0[:, 0:3] + np.array([0.05, 0.0, 0.02]); plan[:, 6] = 0.4
        B = dict(q=q, v=v, normals=np.tile([0, 0, 1.0], (n, 4)), mu=np.full((n, 4), 0.6), mask=np.full(n, 15, np.int32))
        integ = oracle.dynamics(q, v)["p"] if obs else None
        r = np.zeros((n, 18)) if obs else None
        got = _gpu_tracking(torch, solver, H, B, plan, push, integ, r)
        assert np.all(got["status"] == 0)
        err = np.linalg.norm(got["com_traj"][:, -1, 0:3] - plan[:, 3:6], axis=1)
        final[obs] = err
        if obs:
            assert np.abs(got["r"][:, 0:2] - push[:, 0:2]).max() < 1.0      # the residual estimates the horizontal push
            assert np.abs(got["r"][:, 2]).max() < 1.5
    assert final[0].min() > 8e-3 and final[1].max() < 1.5e-3


def test_compute_reference_single_robot(torch_cuda, gpu_model, oracle):
    solver, P, G = _solver(gpu_model, max_batch=1)
    B = synth.make_batch(4, 3, gpu_model.total_mass, rank=44)
    plan = synth.make_plan(B, rank=44)
    ref = oracle.reference(G, B["q"], B["v"], plan, 0.05)
    for s in range(3):
        w, vd, com = solver.compute_reference(B["q"][s], B["v"][s], plan[s], 0.05)
        assert relerr(w, ref["w_des"][s]) < 1e-12 and relerr(vd, ref["vdot_des"][s]) < 1e-12 and relerr(com, ref["com"][s]) < 1e-13


def test_reference_errors(torch_cuda, gpu_model):
    import wbc_quadruped_dob_amd as W
    torch = torch_cuda
    solver, P, G = _solver(gpu_model, max_batch=16, ref=False)
    z = lambda r: torch.zeros((r, 16), dtype=torch.float64, device="cuda")
    with pytest.raises(W.WbcError):      # gains never set
        solver.reference(z(19), z(18), z(12))
    solver.set_ref_params(G)
    solver.reference(z(19) + 1.0, z(18), z(12) + 1.0)
    with pytest.raises(W.WbcError):      # capacity
        solver.reference(torch.zeros((19, 17), dtype=torch.float64, device="cuda"), torch.zeros((18, 17), dtype=torch.float64, device="cuda"),
                         torch.zeros((12, 17), dtype=torch.float64, device="cuda"))

"""GPU: end-to-end scenarios shaped like what the reference ships (/root/reference/README.md:14-17): a Gazebo-style robot
description pushed through the tick, and case study #6 of the reference's figure (/root/reference/play_video_figure.png:
blocks of height 0.015 / 0.02 / 0.04 m with friction 0.4 / 0.8 / 0.6 -- the only terrain numbers the reference
publishes) as a closed-loop rollout: planner -> observer -> GRF QP -> torque map -> forward dynamics on the GPU, with a
subset replayed by the CPU oracle.  Robot models are SYNTHETIC (the DogBot URDF is absent)."""
import itertools
import os

import numpy as np
import pytest

from tests.util import relerr, to_dev, to_host
from wbc_quadruped_dob_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GAZEBO_URDF = os.path.join(ROOT, "tests", "golden", "gazebo_like_quadruped.urdf")
FEET = ["fl_foot", "fr_foot", "rl_foot", "rr_foot"]
CASE6_BLOCKS = [(0.015, 0.4), (0.02, 0.8), (0.04, 0.6)]   # (height m, friction): blue, red, green blocks of case study #6
GROUND = (0.0, 0.6)
# a left/right and front/back symmetric stance of the synthetic quadruped (its right knees and rear rolls have mirrored
# axes): all four feet 0.445 m below the trunk origin
STANCE = np.array([0.05, 0.75, -1.5, -0.05, 0.75, 1.5, -0.05, 0.75, -1.5, 0.05, 0.75, 1.5])


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch


@pytest.mark.parametrize("obs,dtype", [(0, "f64"), (2, "f64"), (1, "f32")])
def test_gazebo_like_description_through_the_tick(torch_cuda, oracle, obs, dtype):
    """The Gazebo-style description (non-axis-aligned joint axes, continuous knees, lumped guards and sensors, legs
    interleaved and back to front, feet named explicitly) parsed by the C++ reader and ticked on the GPU, against the
    oracle built from the independent Python parser's model."""
    import wbc_quadruped_dob_amd as W
    from oracle import oracle_py, urdf_model
    torch = torch_cuda
    model = W.Model.from_urdf(GAZEBO_URDF, foot_links=FEET)
    orc = oracle_py.Oracle(urdf_model.load_urdf(GAZEBO_URDF, foot_links=FEET))
    n = 3000
    P = synth.default_params(observer_order=obs, dtype=dtype)
    solver = W.Solver(model, W.Params.from_dict(P, dtype), dtype=dtype, device=0, max_batch=n, options={})
    B = synth.make_batch(4, n, model.total_mass, rank=71)
    nd = np.float64 if dtype == "f64" else np.float32
    c = lambda a: np.ascontiguousarray(a, nd)
    integ = orc.dynamics(B["q"], B["v"], nthreads=8)["p"] if obs else None
    r = np.zeros((n, 18)) if obs else None
    ig_ref = None if integ is None else c(integ).copy()   # (the oracle advances its observer state in place)
    r_ref = None if r is None else c(r).copy()
    ref = orc.step(P, c(B["q"]), c(B["v"]), c(B["w_des"]), c(B["vdot_des"]), c(B["normals"]), c(B["mu"]), B["mask"], c(B["tau_prev"]),
                   c(B["f_prev"]), ig_ref, r_ref, nthreads=8)
    td = torch.float64 if dtype == "f64" else torch.float32
    dv = lambda k: to_dev(B[k], torch, td)
    ig = None if integ is None else to_dev(integ, torch, td)
    rr = None if r is None else to_dev(r, torch, td)
    out = solver.step(dv("q"), dv("v"), dv("w_des"), dv("vdot_des"), dv("normals"), dv("mu"), torch.from_numpy(B["mask"]).cuda(),
                      dv("tau_prev"), dv("f_prev"), ig, rr, want_mats=True)
    torch.cuda.synchronize()
    tol = 1e-9 if dtype == "f64" else 1e-3
    ok = (out["status"].cpu().numpy() == ref["status"]) & (ref["status"] == 0)
    assert ok.mean() > (0.999 if dtype == "f64" else 0.99)
    d = orc.dynamics(B["q"], B["v"], nthreads=8)
    for k in ("M", "h", "Jc", "pf"):
        assert relerr(to_host(out[k]), d[k]) < (1e-10 if dtype == "f64" else 1e-4), k
    assert relerr(to_host(out["tau"])[ok], ref["tau"][ok]) < tol and relerr(to_host(out["f"])[ok], ref["f"][ok]) < tol
    if obs:
        assert relerr(to_host(rr), r_ref) < (1e-9 if dtype == "f64" else 1e-3)


def _stand_on_blocks(oracle, q_nom, heights):
    """Joint angles that put foot k `heights[:, k]` metres above its nominal height, trunk unchanged (Newton on the own-leg
    block of the oracle's contact Jacobian; the legs are independent 3-joint chains)."""
    n = heights.shape[0]
    q = np.zeros((n, 19)); q[:, 2] = 0.40; q[:, 6] = 1.0; q[:, 7:] = q_nom
    v = np.zeros((n, 18))
    d0 = oracle.dynamics(q, v, nthreads=8)
    target = d0["pf"].reshape(n, 4, 3).copy()
    target[:, :, 2] += heights
    leg_of_foot = None
    for _ in range(8):
        d = oracle.dynamics(q, v, nthreads=8)
        pf = d["pf"].reshape(n, 4, 3)
        J = d["Jc"].reshape(n, 4, 3, 18)[:, :, :, 6:]                      # [n, foot, 3, 12 joints]
        if leg_of_foot is None:
            leg_of_foot = [np.flatnonzero(np.abs(J[0, k]).sum(0) > 1e-9) for k in range(4)]
        for k in range(4):
            cols = leg_of_foot[k]
            dq = np.linalg.solve(J[:, k][:, :, cols], (target[:, k] - pf[:, k])[..., None])[..., 0]
            q[:, 7 + cols] += dq
    pf = oracle.dynamics(q, v, nthreads=8)["pf"].reshape(n, 4, 3)
    assert np.abs(pf - target).max() < 1e-10
    return q


def test_case_study_6_blocks_of_different_height_and_friction(torch_cuda, gpu_model, oracle):
    """Case study #6: every robot stands with three feet on the blue / red / green blocks (h = 0.015 / 0.02 / 0.04 m,
    mu = 0.4 / 0.8 / 0.6) and one on the ground, in all 24 assignments of blocks to feet, shifts its CoM 4 cm and is pushed
    sideways with 30 N from 16 directions (384 robots, 500 ticks, observer on).  The GRF optimisation must keep every foot
    inside ITS friction pyramid, load the slippery foot less tangentially than the grippy one, reject the push, and the
    first 48 robots must agree with the CPU oracle's replay of the same closed loop."""
    import wbc_quadruped_dob_amd as W
    torch = torch_cuda
    assign = list(itertools.permutations(CASE6_BLOCKS + [GROUND]))           # 24 block-to-foot assignments
    ndir = 16
    n, H = len(assign) * ndir, 500
    hts = np.array([[hm[0] for hm in a] for a in assign]).repeat(ndir, axis=0)
    mus = np.array([[hm[1] for hm in a] for a in assign]).repeat(ndir, axis=0)
    ang = np.tile(np.linspace(0, 2 * np.pi, ndir, endpoint=False), len(assign))
    P = synth.default_params(observer_order=1)
    G = synth.default_ref_params()
    G["q_nom"] = STANCE.copy()
    solver = W.Solver(gpu_model, W.Params.from_dict(P), device=0, max_batch=n, options={})
    solver.set_ref_params(G)
    q = _stand_on_blocks(oracle, G["q_nom"], hts)
    v = np.zeros((n, 18))
    push = np.zeros((n, 18)); push[:, 0] = 30 * np.cos(ang); push[:, 1] = 30 * np.sin(ang)
    ident = np.zeros((n, 12)); ident[:, 11] = 1.0
    com0 = oracle.reference(G, q, v, ident)["com"]
    plan = ident.copy(); plan[:, 0:3] = com0[:, 0:3]; plan[:, 3:6] = com0[:, 0:3] + np.array([0.04, 0.0, 0.0]); plan[:, 6] = 0.3
    B = dict(q=q, v=v, normals=np.tile([0, 0, 1.0], (n, 4)), mu=mus, mask=np.full(n, 15, np.int32))
    integ = oracle.dynamics(q, v, nthreads=8)["p"]
    from tests.test_gpu_reference import _gpu_tracking
    got = _gpu_tracking(torch, solver, H, B, plan, push, integ.copy(), np.zeros((n, 18)))
    assert np.all(got["status"] == 0)
    f = got["f"].reshape(n, 4, 3)
    fz, ft = f[:, :, 2], np.linalg.norm(f[:, :, 0:2], axis=2)
    assert fz.min() > 1.0                                                     # every foot keeps pushing on its block
    assert np.all(np.abs(f[:, :, 0]) <= mus * fz + 1e-6) and np.all(np.abs(f[:, :, 1]) <= mus * fz + 1e-6)   # its own pyramid
    ratio = ft / fz
    slip = np.array([r[m == 0.4][0] for r, m in zip(ratio, mus)]); grip = np.array([r[m == 0.8][0] for r, m in zip(ratio, mus)])
    assert slip.max() <= np.sqrt(2) * 0.4 + 1e-6
    assert np.all(np.isfinite(grip))
    err = np.linalg.norm(got["com_traj"][:, -1, 0:3] - plan[:, 3:6], axis=1)
    assert err.max() < 2e-3                                                   # push rejected, goal reached on the uneven stance
    assert np.abs(got["r"][:, 0:2] - push[:, 0:2]).max() < 1.0                # the momentum observer found the push
    # a tilted or harder push must make the slippery foot saturate: the scenario does exercise the friction rows
    hard = push.copy(); hard[:, 0:2] *= 4.0
    got_h = _gpu_tracking(torch, solver, 60, B, plan, hard, integ.copy(), np.zeros((n, 18)))
    fh = got_h["f"].reshape(n, 4, 3)
    sat = np.isclose(np.maximum(np.abs(fh[:, :, 0]), np.abs(fh[:, :, 1])), mus * fh[:, :, 2], rtol=0, atol=1e-6) & (fh[:, :, 2] > 1.0)
    assert sat[mus == 0.4].mean() > 0.2
    # CPU replay of the same closed loop for the first 48 robots
    m = 48
    qo, vo = q[:m].copy(), v[:m].copy()
    ig_o, r_o = integ[:m].copy(), np.zeros((m, 18))
    ref = oracle.rollout_tracking(P, G, 120, qo, vo, plan[:m], B["normals"][:m], mus[:m], B["mask"][:m], tau_ext=push[:m], integ=ig_o, r=r_o,
                                  want_traj=True, want_com=True, nthreads=8)
    Bs = {k: x[:m] for k, x in B.items()}
    s2 = W.Solver(gpu_model, W.Params.from_dict(P), device=0, max_batch=m, options={})
    s2.set_ref_params(G)
    g2 = _gpu_tracking(torch, s2, 120, Bs, plan[:m], push[:m], integ[:m].copy(), np.zeros((m, 18)))
    assert np.array_equal(g2["status"], ref["status"])
    assert relerr(g2["q"], qo) < 1e-8 and relerr(g2["v"], vo) < 1e-7 and relerr(g2["com_traj"], ref["com_traj"]) < 1e-8

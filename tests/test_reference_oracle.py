"""CPU: the oracle's CoM reference generator and planner-in-the-loop rollout against the numpy
fixture, analytic properties of the plan, and the closed-loop case study the reference advertises
(/root/reference/README.md:11 "allows the robot to reject external disturbances")."""
import os

import numpy as np
import pytest

from oracle import crosscheck_np as X
from tests.util import relerr
from wbc_quadruped_dob_amd import synth

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def gref():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_reference_v1.npz")))


def test_reference_vs_golden(oracle, gref):
    G = synth.default_ref_params()
    o = oracle.reference(G, gref["ref_in_q"], gref["ref_in_v"], gref["ref_in_plan"], float(gref["ref_in_t"]))
    assert relerr(o["w_des"], gref["ref_out_w_des"]) < 1e-12
    assert relerr(o["vdot_des"], gref["ref_out_vdot_des"]) < 1e-12
    assert relerr(o["com"], gref["ref_out_com"]) < 1e-13


def test_tracking_rollout_vs_golden(oracle, gref):
    g = lambda k: gref["trk_in_" + k]
    G = synth.default_ref_params()
    P = synth.default_params(observer_order=1)
    H = int(gref["trk_horizon"])
    q, v = g("q").copy(), g("v").copy()
    integ, r = g("integ0").copy(), np.zeros_like(g("integ0"))
    o = oracle.rollout_tracking(P, G, H, q, v, g("plan"), g("normals"), g("mu"), g("mask"), tau_ext=g("tau_ext"), integ=integ,
                                r=r, want_traj=True, want_com=True)
    assert np.all(o["status"] == 0)
    assert relerr(q, gref["trk_out_q"]) < 1e-11
    assert relerr(v, gref["trk_out_v"]) < 1e-10
    assert relerr(o["tau_traj"], gref["trk_out_tau_traj"]) < 1e-9
    assert relerr(o["com_traj"], gref["trk_out_com_traj"]) < 1e-11
    assert relerr(r, gref["trk_out_r"]) < 1e-7


def test_com_state_matches_mass_matrix_and_momentum(oracle, flat_model):
    """CoM from the reference generator = what M encodes: M[lin, ang] = -m [c - p_b]x, and cd = (M v)[0:3] / m."""
    from tests.util import unpack_M
    B = synth.make_batch(3, 16, float(flat_model["mass"].sum()), rank=5)
    plan = synth.make_plan(B)
    com = oracle.reference(synth.default_ref_params(), B["q"], B["v"], plan)["com"]
    d = oracle.dynamics(B["q"], B["v"])
    M = unpack_M(d["M"])
    m = float(flat_model["mass"].sum())
    for s in range(16):
        K = -M[s, 0:3, 3:6] / m
        c = np.array([K[2, 1], K[0, 2], K[1, 0]])
        np.testing.assert_allclose(com[s, 0:3] - B["q"][s, 0:3], c, atol=1e-13)
    np.testing.assert_allclose(com[:, 3:6], d["p"][:, 0:3] / m, atol=1e-13)


def test_plan_time_law_and_regulation_limits(oracle, flat_model):
    G = synth.default_ref_params()
    n = 6
    q = np.zeros((n, 19)); q[:, 2] = 0.4; q[:, 6] = 1.0; q[:, 7:] = G["q_nom"]
    v = np.zeros((n, 18))
    ident = np.zeros((n, 12)); ident[:, 11] = 1.0
    com = oracle.reference(G, q, v, ident)["com"]
    m = float(flat_model["mass"].sum())
    # robot at rest ON its goal, upright, nominal posture, plan finished: no acceleration asked, wrench = weight
    plan = ident.copy(); plan[:, 0:3] = com[:, 0:3] - 0.1; plan[:, 3:6] = com[:, 0:3]; plan[:, 6] = 0.5; plan[:, 7] = 0.5
    o = oracle.reference(G, q, v, plan)
    assert np.abs(o["vdot_des"]).max() < 1e-10
    np.testing.assert_allclose(o["w_des"][:, 0:3], np.tile([0, 0, 9.81 * m], (n, 1)), atol=1e-9)
    np.testing.assert_allclose(o["w_des"][:, 3:6], np.cross(com[:, 0:3] - q[:, 0:3], o["w_des"][:, 0:3]), atol=1e-9)
    # T <= 0 is "already there"; t beyond the end is clamped
    p2 = plan.copy(); p2[:, 6] = 0.0; p2[:, 7] = 0.0
    assert relerr(oracle.reference(G, q, v, p2)["vdot_des"], o["vdot_des"]) < 1e-12 or np.abs(o["vdot_des"]).max() < 1e-9
    assert np.abs(oracle.reference(G, q, v, plan, t=3.0)["vdot_des"]).max() < 1e-10
    # mid-plan feed-forward: at u = 1/2 the quintic has s = 1/2, sd = 15/8 / T, sdd = 0
    p3 = plan.copy(); p3[:, 7] = 0.25
    o3 = oracle.reference(G, q, v, p3)
    d = p3[:, 3:6] - p3[:, 0:3]
    expect = G["kp_com"] * (p3[:, 0:3] + 0.5 * d - com[:, 0:3]) + G["kd_com"] * (15.0 / 8.0 / 0.5 * d)
    np.testing.assert_allclose(o3["vdot_des"][:, 0:3], expect, atol=1e-10)
    # attitude error: desired = 0.2 rad about z from upright -> e_R = 2 sin(0.1) z
    p4 = plan.copy(); p4[:, 8:12] = [0, 0, np.sin(0.1), np.cos(0.1)]
    o4 = oracle.reference(G, q, v, p4)
    np.testing.assert_allclose(o4["vdot_des"][:, 3:6], np.tile([0, 0, G["kp_rot"][2] * 2 * np.sin(0.1)], (n, 1)), atol=1e-10)
    # q and -q are the same attitude
    p5 = p4.copy(); p5[:, 8:12] *= -1
    assert relerr(oracle.reference(G, q, v, p5)["vdot_des"], o4["vdot_des"]) < 1e-13


def test_oracle_matches_numpy_on_random_states(oracle, flat_model):
    npm = X.NPModel(flat_model)
    G = synth.default_ref_params()
    B = synth.make_batch(4, 10, float(flat_model["mass"].sum()), rank=9)
    plan = synth.make_plan(B, rank=9)
    o = oracle.reference(G, B["q"], B["v"], plan, 0.02)
    for s in range(10):
        w, vd, com = X.reference(npm, G, B["q"][s], B["v"][s], plan[s], 0.02)
        assert relerr(o["w_des"][s], w) < 1e-12 and relerr(o["vdot_des"][s], vd) < 1e-12 and relerr(o["com"][s], com) < 1e-13


def _standing(flat_model, oracle, n):
    G = synth.default_ref_params()
    q = np.zeros((n, 19)); q[:, 2] = 0.40; q[:, 6] = 1.0; q[:, 7:] = G["q_nom"]
    v = np.zeros((n, 18))
    ident = np.zeros((n, 12)); ident[:, 11] = 1.0
    com0 = oracle.reference(G, q, v, ident)["com"]
    plan = ident.copy()
    plan[:, 0:3] = com0[:, 0:3]
    plan[:, 3:6] = com0[:, 0:3] + np.array([0.05, 0.0, 0.02])
    plan[:, 6] = 0.4
    return G, q, v, plan


def cref_of(plan, H, dt):
    t = np.arange(H) * dt
    u = np.clip((plan[7] + t) / plan[6], 0, 1)
    s = 10 * u**3 - 15 * u**4 + 6 * u**5
    return plan[0:3] + s[:, None] * (plan[3:6] - plan[0:3])


@pytest.mark.parametrize("order", [1, 2])
def test_case_study_push_rejection_with_the_momentum_observer(oracle, flat_model, order):
    """The reference's headline (README.md:11): planner + momentum observer + GRF optimisation reject an external push.
    Standing robot follows a 5 cm CoM transfer while a constant 36 N push acts on the trunk: without the observer the
    CoM settles ~1 cm off the plan, with it the estimate converges to the push and the error drops by > 10x."""
    n, H = 2, 600
    G, q0, v0, plan = _standing(flat_model, oracle, n)
    normals = np.tile([0, 0, 1.0], (n, 4)); mu = np.full((n, 4), 0.6); mask = np.full(n, 15, np.int32)
    push = np.zeros((n, 18)); push[:, 0] = 30.0; push[:, 1] = -20.0
    errs = {}
    for obs in (0, order):
        P = synth.default_params(observer_order=obs)
        q, v = q0.copy(), v0.copy()
        integ = oracle.dynamics(q, v)["p"].copy() if obs else None
        r = np.zeros((n, 18)) if obs else None
        o = oracle.rollout_tracking(P, G, H, q, v, plan, normals, mu, mask, tau_ext=push, integ=integ, r=r, want_com=True)
        assert np.all(o["status"] == 0)
        err = np.linalg.norm(o["com_traj"][0][:, 0:3] - cref_of(plan[0], H, P["dt"]), axis=1)
        errs[obs] = err
        if obs:
            np.testing.assert_allclose(r[0, 0:3], push[0, 0:3], atol=0.5)     # the residual IS the push
    assert errs[0][-1] > 8e-3                      # observer off: steady offset of about a centimetre
    assert errs[order][-1] < errs[0][-1] / 10      # observer on: rejected
    assert errs[order].max() < 3e-3


def test_case_study_tilted_terrain_keeps_forces_in_the_cones(oracle, flat_model):
    """Irregular-terrain case (README.md:15-16): 15-degree tilted contact normals with mu = 0.4; the tracked plan must
    be achieved with every GRF inside its (tilted) friction pyramid at every tick."""
    n, H = 3, 200
    G, q, v, plan = _standing(flat_model, oracle, n)
    rng = np.random.default_rng(3)
    tilt, az = np.deg2rad(15), rng.uniform(0, 2 * np.pi, (n, 4))
    normals = np.stack([np.sin(tilt) * np.cos(az), np.sin(tilt) * np.sin(az), np.full((n, 4), np.cos(tilt))], 2).reshape(n, 12)
    mu = np.full((n, 4), 0.4); mask = np.full(n, 15, np.int32)
    P = synth.default_params(observer_order=1)
    integ = oracle.dynamics(q, v)["p"].copy(); r = np.zeros((n, 18))
    for _ in range(H // 20):
        o = oracle.rollout_tracking(P, G, 20, q, v, plan, normals, mu, mask, integ=integ, r=r,
                                    tau_prev=None, f_prev=None)
        plan[:, 7] += 20 * P["dt"]
        assert np.all(o["status"] == 0)
        f = o["f_prev"].reshape(n, 4, 3); nn = normals.reshape(n, 4, 3)
        fn = (f * nn).sum(2)
        ft = np.linalg.norm(f - fn[..., None] * nn, axis=2)
        assert np.all(fn >= -1e-9) and np.all(ft <= np.sqrt(2) * mu * fn + 1e-7)
    c = oracle.reference(G, q, v, plan)["com"][:, 0:3]
    assert np.abs(c - cref_of(plan[0], 1, P["dt"])[0]).max() < 5e-3

"""Synthetic per-tick inputs for the BASELINE.json configs.

There is no recorded state from the reference (manual Gazebo runs only,
/root/reference/README.md:56-62), so every batch is seeded synthetic data.  The only
reference-sourced numbers are the friction coefficients 0.4/0.6/0.8 of case study #6
(/root/reference/play_video_figure.png).  Arrays are row-per-state [N, ncomp] float64;
callers transpose to the device layout [ncomp, N].
"""
import numpy as np

SEED = 0xD06B07
NOMINAL_LEG = (0.05, 0.75, -1.5)  # roll, pitch, knee of the synthetic quadruped's stance
TROT_MASKS = (0b1111, 0b1001, 0b0110, 0b1110, 0b0111, 0b1011, 0b1101, 0b0011, 0b1100)


def _rand_unit_quat_near_upright(rng, n, max_angle):
    ax = rng.normal(size=(n, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = rng.uniform(0, max_angle, size=(n, 1))
    return np.concatenate([ax * np.sin(ang / 2), np.cos(ang / 2)], axis=1)


def make_batch(config, n, total_mass, rank=0, nj=12, nf=4):
    """config in {2,3,4}: returns dict of row-per-state float64 arrays + int32 mask."""
    rng = np.random.default_rng(SEED + rank + 1000 * config)
    q = np.zeros((n, 7 + nj))
    q[:, 2] = rng.uniform(0.30, 0.45, n)
    q[:, 3:7] = _rand_unit_quat_near_upright(rng, n, 0.3)
    nominal = np.tile(np.array(NOMINAL_LEG), nj // 3)
    q[:, 7:] = nominal + rng.uniform(-0.3, 0.3, (n, nj))
    v = np.concatenate([rng.uniform(-0.5, 0.5, (n, 3)), rng.uniform(-0.5, 0.5, (n, 3)), rng.uniform(-2, 2, (n, nj))], 1)
    w_des = np.zeros((n, 6))
    w_des[:, 2] = total_mass * 9.81
    w_des[:, 0:3] += rng.uniform(-20, 20, (n, 3))
    w_des[:, 3:6] += rng.uniform(-5, 5, (n, 3))
    vdot_des = np.concatenate([rng.uniform(-1, 1, (n, 6)), rng.uniform(-5, 5, (n, nj))], 1)
    normals = np.tile(np.array([0.0, 0.0, 1.0]), (n, nf))
    mu = np.full((n, nf), 0.6)
    mask = np.full(n, (1 << nf) - 1, dtype=np.int32)
    tau_prev = np.zeros((n, nj))
    f_prev = np.zeros((n, 3 * nf))
    push = np.zeros((n, 3))
    if config >= 3:
        mask = np.array([TROT_MASKS[i % len(TROT_MASKS)] for i in range(n)], dtype=np.int32)
        push = rng.uniform(-50, 50, (n, 3))
        tau_prev = rng.uniform(-20, 20, (n, nj))
        f_prev = np.zeros((n, 3 * nf))
        for k in range(nf):
            on = ((mask >> k) & 1).astype(np.float64)
            f_prev[:, 3 * k + 2] = on * rng.uniform(20, 80, n)
            f_prev[:, 3 * k:3 * k + 2] = on[:, None] * rng.uniform(-8, 8, (n, 2))
    if config >= 4:
        tilt = rng.uniform(0, np.deg2rad(15), (n, nf))
        az = rng.uniform(0, 2 * np.pi, (n, nf))
        normals = np.stack([np.sin(tilt) * np.cos(az), np.sin(tilt) * np.sin(az), np.cos(tilt)], axis=2).reshape(n, 3 * nf)
        mu = rng.choice([0.4, 0.6, 0.8], size=(n, nf))
    return dict(q=q, v=v, w_des=w_des, vdot_des=vdot_des, normals=normals, mu=mu, mask=mask, tau_prev=tau_prev,
                f_prev=f_prev, push=push)


def default_params(nv=18, observer_order=0, dtype="f64"):
    return dict(S=np.ones(6), alpha=1e-3, fn_min=0.0, fn_max=400.0, mu_scale=1.0, dt=1e-3,
                observer_order=observer_order, max_iter=100, qp_tol=1e-9 if dtype == "f64" else 1e-3,
                K1=np.full(nv, 50.0), K2=np.full(nv, 200.0))


def default_ref_params(nj=12):
    """Gains of the CoM reference generator (wbc_ref_params); nominal posture = the synthetic quadruped's stance."""
    return dict(kp_com=np.array([100.0, 100.0, 150.0]), kd_com=np.array([20.0, 20.0, 25.0]),
                kp_rot=np.array([200.0, 200.0, 100.0]), kd_rot=np.array([25.0, 25.0, 15.0]), kp_joint=200.0, kd_joint=28.0,
                inertia_nom=np.array([0.8, 1.85, 2.05]), q_nom=np.tile(np.array(NOMINAL_LEG), nj // 3))


def make_plan(B, rank=0, duration=0.5, reach=0.08):
    """Synthetic CoM plans for a batch from make_batch(): start near the base position, goal within `reach` metres,
    random elapsed time, desired attitude within 0.1 rad of upright.  Row-per-state [n, 12] float64.
    Edge rows: state 3 (if present) has T = 0 (goal reached: pure regulation), state 4 is past its end time."""
    n = B["q"].shape[0]
    rng = np.random.default_rng(SEED + 77 + rank)
    plan = np.zeros((n, 12))
    plan[:, 0:3] = B["q"][:, 0:3] + rng.uniform(-0.02, 0.02, (n, 3))
    plan[:, 3:6] = plan[:, 0:3] + rng.uniform(-reach, reach, (n, 3))
    plan[:, 6] = duration
    plan[:, 7] = rng.uniform(0.0, 0.6 * duration, n)
    plan[:, 8:12] = _rand_unit_quat_near_upright(rng, n, 0.1)
    if n > 3:
        plan[3, 6] = 0.0
    if n > 4:
        plan[4, 7] = 1.5 * duration
    return plan

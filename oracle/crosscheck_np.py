"""TEST INFRASTRUCTURE -- independent numpy/scipy implementation of the WBC hot path.

PARITY UNPINNED w.r.t. the reference (its sources are absent).  This module is the
*second opinion* that pins the C++ oracle (oracle/wbc_oracle.hpp): every quantity is computed by a
DIFFERENT algorithm from the oracle's, so an error in either shows up as a disagreement:

  quantity            oracle (C++)                         here (numpy)
  ------------------  -----------------------------------  ------------------------------------------
  kinematics          3x3 rotation recursion               4x4 homogeneous-transform products
  M(q)                CRBA, body coordinates               sum_i m J_v^T J_v + J_w^T I_w J_w (world frame)
  h(q,v)              spatial RNEA, body coordinates        Kane/virtual-power projection of world-frame
                                                           Newton-Euler body accelerations
  Jc                  geometric columns                    same geometric formula on the 4x4 FK, plus a
                                                           finite-difference check of the FK map
  beta = C^T v - g    one momentum sweep                   (d/dt M(q(t))) v - h by Richardson-extrapolated
                                                           central differences of M along the flow
  QP                  Goldfarb-Idnani DUAL active set      PRIMAL active set (Nocedal & Wright alg. 16.3)
                                                           with dense KKT solves + KKT residual check
                                                           (+ scipy SLSQP as a loose third opinion)

It also generates the committed fixtures (tests/golden/make_golden.py).
Conventions are those of wbc_oracle.hpp's header (mixed velocity representation etc.).
"""
import numpy as np

NV = None  # set per model


def skew(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])


def quat_to_R(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def axis_angle_R(a, th):
    K = skew(a)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def quat_mul(a, b):  # (x,y,z,w)
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def integrate_q(q, v, eps):
    """q(t+eps) for constant mixed velocity v: world-frame omega => left-multiplied increment."""
    qn = q.copy()
    qn[0:3] += eps * v[0:3]
    w = v[3:6] * eps
    th = np.linalg.norm(w)
    if th > 0:
        dq = np.concatenate([np.sin(th / 2) * w / th, [np.cos(th / 2)]])
    else:
        dq = np.array([0, 0, 0, 1.0])
    qn[3:7] = quat_mul(dq, q[3:7] / np.linalg.norm(q[3:7]))
    qn[7:] += eps * v[6:]
    return qn


class NPModel:
    def __init__(self, flat):
        self.nb = flat["nb"]
        self.nv = 6 + self.nb - 1
        self.parent = flat["parent"]
        self.Rt = flat["Rt"].reshape(-1, 3, 3)
        self.rt = flat["rt"]
        self.axis = flat["axis"]
        self.mass = flat["mass"]
        self.com = flat["com"]
        Ic = flat["Ic"]
        self.Ic = np.array([[[a[0], a[1], a[2]], [a[1], a[3], a[4]], [a[2], a[4], a[5]]] for a in Ic])
        self.foot_body = flat["foot_body"]
        self.foot_off = flat["foot_off"]
        self.nf = len(self.foot_body)
        self.grav = flat["gravity"]

    # ---- kinematics by homogeneous transforms
    def fk(self, q):
        T = [None] * self.nb
        T0 = np.eye(4)
        T0[:3, :3] = quat_to_R(q[3:7])
        T0[:3, 3] = q[0:3]
        T[0] = T0
        for i in range(1, self.nb):
            Tj = np.eye(4)
            Tj[:3, :3] = self.Rt[i] @ axis_angle_R(self.axis[i], q[7 + i - 1])
            Tj[:3, 3] = self.rt[i]
            T[i] = T[self.parent[i]] @ Tj
        return T

    def foot_pos(self, q, T=None):
        T = T or self.fk(q)
        return np.array([(T[b] @ np.append(o, 1.0))[:3] for b, o in zip(self.foot_body, self.foot_off)])

    def ancestors(self, i):
        out = []
        while i > 0:
            out.append(i)
            i = self.parent[i]
        return out

    def point_jacobian(self, T, body, pw):
        """3 x nv Jacobian of the world velocity of point pw (world coords) fixed to `body`."""
        J = np.zeros((3, self.nv))
        J[:, 0:3] = np.eye(3)
        J[:, 3:6] = -skew(pw - T[0][:3, 3])
        for k in self.ancestors(body):
            z = T[k][:3, :3] @ self.axis[k]
            J[:, 6 + k - 1] = np.cross(z, pw - T[k][:3, 3])
        return J

    def rot_jacobian(self, T, body):
        J = np.zeros((3, self.nv))
        J[:, 3:6] = np.eye(3)
        for k in self.ancestors(body):
            J[:, 6 + k - 1] = T[k][:3, :3] @ self.axis[k]
        return J

    def contact_jacobians(self, q):
        T = self.fk(q)
        pf = self.foot_pos(q, T)
        return np.array([self.point_jacobian(T, b, p) for b, p in zip(self.foot_body, pf)]), pf

    def contact_jacobians_fd(self, q, eps=1e-6):
        """finite-difference Jacobian of the foot positions along each generalized velocity."""
        J = np.zeros((self.nf, 3, self.nv))
        for c in range(self.nv):
            e = np.zeros(self.nv)
            e[c] = 1.0
            fp = self.foot_pos(integrate_q(q, e, eps))
            fm = self.foot_pos(integrate_q(q, e, -eps))
            J[:, :, c] = (fp - fm) / (2 * eps)
        return J

    # ---- mass matrix from body Jacobians
    def mass_matrix(self, q):
        T = self.fk(q)
        M = np.zeros((self.nv, self.nv))
        for i in range(self.nb):
            R = T[i][:3, :3]
            c = (T[i] @ np.append(self.com[i], 1.0))[:3]
            Jv = self.point_jacobian(T, i, c)
            Jw = self.rot_jacobian(T, i)
            Iw = R @ self.Ic[i] @ R.T
            M += self.mass[i] * Jv.T @ Jv + Jw.T @ Iw @ Jw
        return M

    # ---- inverse dynamics by virtual power of world-frame Newton-Euler body wrenches
    def inverse_dynamics(self, q, v, vdot=None, gravity=True):
        """M vdot + C v + g, mixed representation."""
        nv = self.nv
        vdot = np.zeros(nv) if vdot is None else vdot
        T = self.fk(q)
        om = [None] * self.nb
        al = [None] * self.nb
        pdd = [None] * self.nb  # acceleration of body origin
        pd = [None] * self.nb
        om[0] = v[3:6].copy()
        al[0] = vdot[3:6].copy()
        pd[0] = v[0:3].copy()
        pdd[0] = vdot[0:3].copy()
        for i in range(1, self.nb):
            p = self.parent[i]
            z = T[i][:3, :3] @ self.axis[i]
            d = T[i][:3, 3] - T[p][:3, 3]
            qd, qdd = v[6 + i - 1], vdot[6 + i - 1]
            om[i] = om[p] + z * qd
            al[i] = al[p] + np.cross(om[p], z) * qd + z * qdd
            pd[i] = pd[p] + np.cross(om[p], d)
            pdd[i] = pdd[p] + np.cross(al[p], d) + np.cross(om[p], np.cross(om[p], d))
        g = self.grav if gravity else np.zeros(3)
        out = np.zeros(nv)
        for i in range(self.nb):
            R = T[i][:3, :3]
            c = (T[i] @ np.append(self.com[i], 1.0))[:3]
            rc = c - T[i][:3, 3]
            ac = pdd[i] + np.cross(al[i], rc) + np.cross(om[i], np.cross(om[i], rc))
            Iw = R @ self.Ic[i] @ R.T
            F = self.mass[i] * (ac - g)
            N = Iw @ al[i] + np.cross(om[i], Iw @ om[i])
            out += self.point_jacobian(T, i, c).T @ F + self.rot_jacobian(T, i).T @ N
        return out

    def bias(self, q, v):
        return self.inverse_dynamics(q, v)

    def gravity_vec(self, q):
        return self.inverse_dynamics(q, np.zeros(self.nv))

    def Mdot_v(self, q, v, eps=2e-4):
        """(d/dt M(q(t))) v along qdot <-> v, Richardson-extrapolated central differences."""
        def D(h):
            return (self.mass_matrix(integrate_q(q, v, h)) - self.mass_matrix(integrate_q(q, v, -h))) @ v / (2 * h)
        return (4 * D(eps / 2) - D(eps)) / 3

    def beta(self, q, v):
        """C^T v - g  ==  Mdot v - h   (since Mdot = C + C^T and h = C v + g)."""
        return self.Mdot_v(q, v) - self.bias(q, v)


# ------------------------------------------------------------------ QP pieces
def contact_frame(n):
    n = n / np.linalg.norm(n)
    ref = np.array([1.0, 0, 0]) if abs(n[0]) < 0.9 else np.array([0, 1.0, 0])
    t1 = ref - n * ref.dot(n)
    t1 /= np.linalg.norm(t1)
    return t1, np.cross(n, t1), n


def qp_assemble(P, mask, pb, pf, normals, mu, b):
    st = [f for f in range(len(pf)) if mask & (1 << f)]
    ns = len(st)
    A = np.zeros((6, 3 * ns))
    for s, f in enumerate(st):
        A[0:3, 3 * s:3 * s + 3] = np.eye(3)
        A[3:6, 3 * s:3 * s + 3] = skew(pf[f] - pb)
    S = np.diag(P["S"])
    H = A.T @ S @ A + P["alpha"] * np.eye(3 * ns)
    g = -A.T @ S @ b
    C = np.zeros((6 * ns, 3 * ns))
    d = np.zeros(6 * ns)
    for s, f in enumerate(st):
        t1, t2, n = contact_frame(normals[f])
        mt = mu[f] * P["mu_scale"]
        rows = [mt * n - t1, mt * n + t1, mt * n - t2, mt * n + t2, n, -n]
        rhs = [0, 0, 0, 0, P["fn_min"], -P["fn_max"]]
        for c in range(6):
            C[6 * s + c, 3 * s:3 * s + 3] = rows[c]
            d[6 * s + c] = rhs[c]
    return H, g, C, d, st


def qp_primal_active_set(H, g, C, d, x0, max_iter=200, tol=1e-11):
    """Nocedal & Wright alg. 16.3 (primal active set), dense KKT solves.  C x >= d.  x0 feasible."""
    n = len(g)
    x = x0.copy()
    assert np.all(C @ x - d >= -1e-9), "x0 infeasible"
    W = [i for i in range(len(d)) if abs(C[i] @ x - d[i]) < 1e-12]
    # keep W linearly independent
    Wl = []
    for i in W:
        if np.linalg.matrix_rank(C[Wl + [i]]) == len(Wl) + 1:
            Wl.append(i)
    W = Wl
    for it in range(max_iter):
        k = len(W)
        K = np.zeros((n + k, n + k))
        K[:n, :n] = H
        if k:
            K[:n, n:] = -C[W].T
            K[n:, :n] = C[W]
        rhs = np.concatenate([-(H @ x + g), np.zeros(k)])
        sol = np.linalg.solve(K, rhs)
        p, lam = sol[:n], sol[n:]
        if np.linalg.norm(p) < tol * (1 + np.linalg.norm(x)):
            if k == 0 or lam.min() >= -1e-12:
                lam_full = np.zeros(len(d))
                for i, l in zip(W, lam):
                    lam_full[i] = l
                return x, lam_full, it
            W.pop(int(np.argmin(lam)))
            continue
        alpha, blk = 1.0, -1
        for i in range(len(d)):
            if i in W:
                continue
            cp = C[i] @ p
            if cp < -1e-14:
                a = (d[i] - C[i] @ x) / cp
                if a < alpha:
                    alpha, blk = a, i
        x = x + max(alpha, 0.0) * p
        if blk >= 0:
            if np.linalg.matrix_rank(C[W + [blk]]) == len(W) + 1:
                W.append(blk)
    raise RuntimeError("primal active set did not converge")


def kkt_residuals(H, g, C, d, x, lam):
    stat = np.linalg.norm(H @ x + g - C.T @ lam, np.inf)
    feas = max(0.0, float(np.max(d - C @ x))) if len(d) else 0.0
    dual = max(0.0, float(np.max(-lam))) if len(d) else 0.0
    comp = float(np.max(np.abs(lam * (C @ x - d)))) if len(d) else 0.0
    return stat, feas, dual, comp


def feasible_start(P, st, normals):
    x = np.zeros(3 * len(st))
    c = min(max(P["fn_min"] + 1.0, 1.0), P["fn_max"])
    for s, f in enumerate(st):
        n = normals[f] / np.linalg.norm(normals[f])
        x[3 * s:3 * s + 3] = c * n
    return x


# ------------------------------------------------------------------ whole step
def default_params(nv=18):
    return dict(S=np.array([1.0, 1.0, 1.0, 1.0, 1.0, 1.0]), alpha=1e-3, fn_min=0.0, fn_max=400.0, mu_scale=1.0,
                dt=1e-3, observer_order=0, max_iter=100, qp_tol=1e-9, K1=np.full(nv, 50.0), K2=np.full(nv, 200.0))


def step(model, P, q, v, w_des, vdot_des, normals, mu, mask, tau_prev, f_prev, integ, r):
    """One control tick for one state by the independent algorithms.  Returns dict."""
    nv = model.nv
    M = model.mass_matrix(q)
    h = model.bias(q, v)
    Jc, pf = model.contact_jacobians(q)
    Jd = Jc.reshape(-1, nv)
    out = dict(M=M, h=h, Jc=Jc, pf=pf)
    rhat = np.zeros(nv)
    if P["observer_order"] > 0:
        beta = model.beta(q, v)
        p = M @ v
        u = np.concatenate([np.zeros(6), tau_prev]) + Jd.T @ f_prev
        integ = integ + P["dt"] * (u + beta + r)
        e = p - integ
        if P["observer_order"] == 1:
            r = P["K1"] * e
        else:
            r = r + P["dt"] * P["K2"] * (P["K1"] * e - r)
        rhat = r
        out.update(beta=beta, p=p)
    out.update(integ=integ, r=r)
    b = w_des - rhat[:6]
    H, g, C, d, st = qp_assemble(P, mask, q[0:3], pf, normals, mu, b)
    f = np.zeros(3 * model.nf)
    if st:
        x, lam, it = qp_primal_active_set(H, g, C, d, feasible_start(P, st, normals))
        out["kkt"] = kkt_residuals(H, g, C, d, x, lam)
        for s, ft in enumerate(st):
            f[3 * ft:3 * ft + 3] = x[3 * s:3 * s + 3]
        out["nactive"] = int(np.sum(lam > 1e-12))
    tau = (M @ vdot_des + h - Jd.T @ f - rhat)[6:]
    out.update(f=f, tau=tau)
    return out


# ------------------------------------------------------------------ rollout (a10): independent restatement
def rollout(model, P, horizon, q, v, w_des, vdot_des, normals, mu, mask, tau_ext=None, integ=None, r=None):
    """horizon ticks of {step by the independent algorithms, vdot = solve(M, S^T tau + Jc^T f + tau_ext - h) with
    numpy's LU, semi-implicit Euler with integrate_q}.  Returns final (q, v, tau list, integ, r)."""
    nv = model.nv
    q, v = q.copy(), v.copy()
    tau_prev = np.zeros(nv - 6)
    f_prev = np.zeros(3 * model.nf)
    integ = np.zeros(nv) if integ is None else integ.copy()
    r = np.zeros(nv) if r is None else r.copy()
    taus = []
    for _ in range(horizon):
        o = step(model, P, q, v, w_des, vdot_des, normals, mu, mask, tau_prev, f_prev, integ, r)
        integ, r = o["integ"], o["r"]
        rhs = np.concatenate([np.zeros(6), o["tau"]]) + o["Jc"].reshape(-1, nv).T @ o["f"] - o["h"]
        if tau_ext is not None:
            rhs = rhs + tau_ext
        vdot = np.linalg.solve(o["M"], rhs)
        v = v + P["dt"] * vdot
        q = integrate_q(q, v, P["dt"])
        tau_prev, f_prev = o["tau"], o["f"]
        taus.append(o["tau"])
    return q, v, np.array(taus), integ, r


# ------------------------------------------------------------------ CoM reference generator (a11): independent restatement
def com_state(model, q, v):
    """CoM position and velocity by the mass-weighted point Jacobians (oracle: first moments + linear momentum)."""
    T = model.fk(q)
    mtot = model.mass.sum()
    c = np.zeros(3)
    Jc = np.zeros((3, model.nv))
    for i in range(model.nb):
        ci = (T[i] @ np.append(model.com[i], 1.0))[:3]
        c += model.mass[i] * ci
        Jc += model.mass[i] * model.point_jacobian(T, i, ci)
    return c / mtot, Jc @ v / mtot, mtot, T


def default_ref_params(nj=12):
    return dict(kp_com=np.array([100.0, 100.0, 150.0]), kd_com=np.array([20.0, 20.0, 25.0]),
                kp_rot=np.array([200.0, 200.0, 100.0]), kd_rot=np.array([25.0, 25.0, 15.0]), kp_joint=200.0, kd_joint=28.0,
                inertia_nom=np.array([0.8, 1.85, 2.05]), q_nom=np.zeros(nj))


def reference(model, G, q, v, plan, t):
    """Returns (w_des[6], vdot_des[nv], com[6]); plan = [c0 3, c1 3, T, t0, quat_des 4]."""
    c, cd, mtot, T = com_state(model, q, v)
    Tp = plan[6]
    u = min(max((plan[7] + t) / Tp, 0.0), 1.0) if Tp > 0 else 1.0
    iT = 1.0 / Tp if Tp > 0 else 0.0
    s = np.polyval([6, -15, 10, 0, 0, 0], u)
    sd = np.polyval(np.polyder([6, -15, 10, 0, 0, 0]), u) * iT
    sdd = np.polyval(np.polyder([6, -15, 10, 0, 0, 0], 2), u) * iT * iT
    d = plan[3:6] - plan[0:3]
    a_cmd = sdd * d + G["kp_com"] * (plan[0:3] + s * d - c) + G["kd_com"] * (sd * d - cd)
    # attitude error from rotation matrices: R_e = R_des R^T, e_R = 2 * vector part of its quaternion (w >= 0)
    R = quat_to_R(q[3:7])
    Re = quat_to_R(plan[8:12]) @ R.T
    w4 = max(1.0 + np.trace(Re), 0.0)
    ew = 0.5 * np.sqrt(w4)
    if ew > 1e-6:
        evec = np.array([Re[2, 1] - Re[1, 2], Re[0, 2] - Re[2, 0], Re[1, 0] - Re[0, 1]]) / (4 * ew)
    else:  # rotation by pi: fall back to the quaternion product
        qc = q[3:7] / np.linalg.norm(q[3:7]) * np.array([-1, -1, -1, 1.0])
        qe = quat_mul(plan[8:12] / np.linalg.norm(plan[8:12]), qc)
        evec = qe[:3] * (1 if qe[3] >= 0 else -1)
    al_cmd = G["kp_rot"] * 2 * evec - G["kd_rot"] * v[3:6]
    nj = model.nv - 6
    vdot_des = np.concatenate([a_cmd, al_cmd, G["kp_joint"] * (G["q_nom"][:nj] - q[7:]) - G["kd_joint"] * v[6:]])
    F = mtot * (a_cmd - model.grav)
    Mo = np.cross(c - q[0:3], F) + R @ (G["inertia_nom"] * (R.T @ al_cmd))
    return np.concatenate([F, Mo]), vdot_des, np.concatenate([c, cd])


def rollout_tracking(model, P, G, horizon, q, v, plan, normals, mu, mask, tau_ext=None, integ=None, r=None):
    nv = model.nv
    q, v = q.copy(), v.copy()
    tau_prev = np.zeros(nv - 6)
    f_prev = np.zeros(3 * model.nf)
    integ = np.zeros(nv) if integ is None else integ.copy()
    r = np.zeros(nv) if r is None else r.copy()
    taus, coms = [], []
    for k in range(horizon):
        w_des, vdot_des, com = reference(model, G, q, v, plan, k * P["dt"])
        o = step(model, P, q, v, w_des, vdot_des, normals, mu, mask, tau_prev, f_prev, integ, r)
        integ, r = o["integ"], o["r"]
        rhs = np.concatenate([np.zeros(6), o["tau"]]) + o["Jc"].reshape(-1, nv).T @ o["f"] - o["h"]
        if tau_ext is not None:
            rhs = rhs + tau_ext
        vdot = np.linalg.solve(o["M"], rhs)
        v = v + P["dt"] * vdot
        q = integrate_q(q, v, P["dt"])
        tau_prev, f_prev = o["tau"], o["f"]
        taus.append(o["tau"])
        coms.append(com)
    return q, v, np.array(taus), np.array(coms), integ, r

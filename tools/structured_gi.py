#!/usr/bin/env python3
"""Design prototype (numpy, CPU, this container only): the GRF QP's Goldfarb-Idnani iteration with ALL linear algebra done
on the 6-dimensional wrench space instead of on 12 x 12 factors.

    min 1/2 alpha |f|^2 + 1/2 |B f - beta|^2,  B = S^(1/2) [I ; [d_k]x] per stance foot,   s.t. per-foot pyramid + box

H = alpha I + B^T B, constraints are local to a foot (3 variables).  With N_k the active normals of foot k (<= 3, linearly
independent), P_k the projector onto their null space and G_A = alpha I + sum_k B_k P_k B_k^T (6 x 6):

    primal step direction for candidate normal n+ on foot kp:   v = P_kp n+,  b = B_kp v,  y = G_A^-1 b,
        z_k = (delta_{k,kp} v - P_k B_k^T y) / alpha,    z . n+ = (|v|^2 - b . y) / alpha      ( = |d2|^2 of the dense method)
    dual step direction (rate of decrease of the active multipliers):  r_k = N_k^+ (delta_{k,kp} n+ - B_k^T y)
    adding n+:   G_A^-1 += y y^T / (alpha z . n+)     (Sherman-Morrison: the update needs nothing that is not already there)

Pivoting rules, tolerances and status codes are those of oracle/wbc_oracle.hpp:qp_solve_gi, so the iterates are the SAME
(same constraint added / dropped in every iteration) up to rounding.  Run: python tools/structured_gi.py [n] -- compares
solutions, iteration counts and status with the C++ oracle on synthetic batches and prints the accuracy of the
Sherman-Morrison-updated inverse against a refactorisation in every iteration.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def skew(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])


def contact_frame(n):
    nn = n / np.linalg.norm(n)
    ref = np.array([1.0, 0, 0]) if abs(nn[0]) < 0.9 else np.array([0, 1.0, 0])
    t = ref - nn * ref.dot(nn)
    t1 = t / np.linalg.norm(t)
    return t1, np.cross(nn, t1), nn


def proj_and_pinv(normals):
    """normals: list of <= 3 independent 3-vectors.  Returns P (projector onto their null space) and N^+ (rows)."""
    q = len(normals)
    if q == 0:
        return np.eye(3), np.zeros((0, 3))
    if q == 1:
        n1 = normals[0]
        i = 1.0 / n1.dot(n1)
        return np.eye(3) - np.outer(n1, n1) * i, (n1 * i)[None, :]
    if q == 2:
        n1, n2 = normals
        w = np.cross(n1, n2)
        i = 1.0 / w.dot(w)
        return np.outer(w, w) * i, np.stack([np.cross(n2, w) * i, np.cross(w, n1) * i])
    n1, n2, n3 = normals
    c1, c2, c3 = np.cross(n2, n3), np.cross(n3, n1), np.cross(n1, n2)
    i = 1.0 / n1.dot(c1)
    return np.zeros((3, 3)), np.stack([c1 * i, c2 * i, c3 * i])


class StructuredGI:
    def __init__(self, S, alpha, mask, d, normals, mu, fmin, fmax, b, tol=1e-9, max_iter=100, sm_update=True):
        self.alpha, self.tol, self.max_iter, self.sm = alpha, tol, max_iter, sm_update
        self.on = [(mask >> k) & 1 for k in range(4)]
        sS = np.sqrt(S)
        self.B = [np.vstack([np.eye(3), skew(d[k])]) * sS[:, None] * self.on[k] for k in range(4)]   # 6 x 3 each
        self.beta = sS * b
        self._S = np.asarray(S, float)
        self.C, self.rhs = [], []      # constraint k*6+c: normal (3), rhs; C f_k >= rhs
        for k in range(4):
            t1, t2, nn = contact_frame(normals[k])
            mt = mu[k]
            self.C += [nn * mt - t1, nn * mt + t1, nn * mt - t2, nn * mt + t2, nn, -nn]
            self.rhs += [0, 0, 0, 0, fmin, -fmax]
        self.sm_err = 0.0

    def G(self, P):
        return self.alpha * np.eye(6) + sum(self.B[k] @ P[k] @ self.B[k].T for k in range(4))

    def warm_setup(self, ids):
        """Block set-up from a GIVEN active set (what csrc/qp_struct16.hip.hpp does for dependent ticks): per-foot projectors and
        pseudo-inverses in closed form, G_A factorised directly, the minimiser ON the set and its multipliers
            f_k = f_k^p - P_k B_k^T y,   f_k^p = N_k^+T rhs_k,   G_A y = sum_k B_k f_k^p - beta,   u_k = alpha N_k^+ (f_k + B_k^T y).
        Returns None when the set is no S-pair of the dual method (more than three rows on a foot, dependent rows), and the list of rows
        with a negative multiplier when there are any (the caller retries once without them)."""
        a, B = self.alpha, self.B
        act = [[] for _ in range(4)]
        for c in sorted(ids):
            if self.on[c // 6]:
                act[c // 6].append(c)
        P, Np = [], []
        for k in range(4):
            nrm = [self.C[c] for c in act[k]]
            if len(nrm) > 3:
                return None
            if len(nrm) == 2 and np.cross(nrm[0], nrm[1]).dot(np.cross(nrm[0], nrm[1])) <= 1e-12:
                return None
            if len(nrm) == 3 and nrm[0].dot(np.cross(nrm[1], nrm[2])) ** 2 <= 1e-12:
                return None
            Pk, Nk = proj_and_pinv(nrm)
            P.append(Pk); Np.append(Nk)
        Ginv = np.linalg.inv(self.G(P))
        fp = [Np[k].T @ np.array([self.rhs[c] for c in act[k]]) if act[k] else np.zeros(3) for k in range(4)]
        y = Ginv @ (sum(B[k] @ fp[k] for k in range(4)) - self.beta)
        f = [fp[k] - P[k] @ (B[k].T @ y) for k in range(4)]
        u = {}
        for k in range(4):
            if act[k]:
                for c, val in zip(act[k], a * (Np[k] @ (f[k] + B[k].T @ y))):
                    u[c] = val
        neg = [c for c, val in u.items() if not (val >= 0)]
        if neg:
            return neg
        return act, P, Np, Ginv, np.concatenate(f), u

    def solve(self, warm=None, move_to=None):
        """move_to (a target wrench b): the speculative start of the fused observer-on tick (csrc/qp_struct16.hip.hpp, SPEC) -- the iteration runs on
        the wrench this object was built with (b~), then the minimiser on the active set it has reached is moved to b:
            d = S^(1/2) (b - b~),  dy = -G_A^-1 d,  df_k = -P_k B_k^T dy,  du_k = alpha N_k^+ (df_k + B_k^T dy),
        and the iteration goes on from there; a negative multiplier after the move starts the solve over from the empty set with b.
        self.moved = True when the moved point was an S-pair."""
        a, B = self.alpha, self.B
        act = [[] for _ in range(4)]                # per foot: constraint ids in slot order
        order = []                                  # global add order (tie-breaking of the ratio test)
        u = {}
        P = [np.eye(3) for _ in range(4)]
        Np = [np.zeros((0, 3)) for _ in range(4)]
        Ginv = np.linalg.inv(self.G(P))
        x = np.concatenate([B[k].T @ (Ginv @ self.beta) for k in range(4)])   # unconstrained minimum B^T G^-1 S^(1/2) b
        ws = self.warm_setup(warm) if warm is not None else None
        if isinstance(ws, list):      # rows with negative multipliers (constraints the state has just left): once more without them, else cold
            ws = self.warm_setup([c for c in warm if c not in ws])
            ws = None if isinstance(ws, list) else ws
        self.warm_used = ws is not None
        if ws is not None:
            act, P, Np, Ginv, x, u = ws
            order = [c for k in range(4) for c in act[k]]
        eps = np.finfo(float).eps
        Rnorm, it, status = 1.0, 0, 0
        slack = lambda c, xx: self.C[c].dot(xx[3 * (c // 6):3 * (c // 6) + 3]) - self.rhs[c]

        def refresh(k):
            P[k], Np[k] = proj_and_pinv([self.C[c] for c in act[k]])

        def drop(l, Ginv):
            k = l // 6
            Pold = P[k]
            act[k].remove(l); order.remove(l); del u[l]
            refresh(k)
            if self.sm:   # rank-one UPDATE of G (P grows by w w^T): G^-1 -= (G^-1 bb)(G^-1 bb)^T / (1 + bb . G^-1 bb)
                D = P[k] - Pold
                ev, evec = np.linalg.eigh(D)
                w = evec[:, -1] * np.sqrt(max(ev[-1], 0.0))
                bb = B[k] @ w
                gb = Ginv @ bb
                Ginv = Ginv - np.outer(gb, gb) / (1.0 + bb.dot(gb))
                self.sm_err = max(self.sm_err, np.abs(Ginv - np.linalg.inv(self.G(P))).max() / np.abs(Ginv).max())
                return Ginv
            return np.linalg.inv(self.G(P))

        while True:
            ip, smin = -1, -self.tol
            for c in range(24):
                if not self.on[c // 6] or c in u:
                    continue
                s = slack(c, x)
                if s < smin:
                    smin, ip = s, c
            if ip < 0:
                if move_to is None:
                    break
                beta_new = np.sqrt(self._S) * np.asarray(move_to, float)
                d = beta_new - self.beta
                self.beta, move_to = beta_new, None
                dy = -Ginv @ d
                df = [-P[k] @ (B[k].T @ dy) for k in range(4)]
                for k in range(4):
                    if act[k]:
                        for c, val in zip(act[k], a * (Np[k] @ (df[k] + B[k].T @ dy))):
                            u[c] += val
                x = x + np.concatenate(df)
                self.moved = all(val >= 0 for val in u.values())
                if not self.moved:      # no S-pair for b: from the empty set, with b
                    act = [[] for _ in range(4)]; order = []; u = {}
                    P = [np.eye(3) for _ in range(4)]; Np = [np.zeros((0, 3)) for _ in range(4)]
                    Ginv = np.linalg.inv(self.G(P))
                    x = np.concatenate([B[k].T @ (Ginv @ self.beta) for k in range(4)])
                    Rnorm = 1.0
                continue
            sip, up, kp = smin, 0.0, ip // 6
            nplus = self.C[ip]
            while True:
                it += 1
                if it > self.max_iter:
                    return x, it, 1, u
                v = P[kp] @ nplus
                bb = B[kp] @ v
                y = Ginv @ bb
                zn = (v.dot(v) - bb.dot(y)) / a
                z = np.concatenate([((v if k == kp else 0) - P[k] @ (B[k].T @ y)) / a for k in range(4)])
                r = {}
                for k in range(4):
                    if act[k]:
                        rk = Np[k] @ ((nplus if k == kp else 0) - B[k].T @ y)
                        for c, val in zip(act[k], rk):
                            r[c] = val
                t1, l = np.inf, -1
                for c in order:
                    if r[c] > 0 and u[c] / r[c] < t1:
                        t1, l = u[c] / r[c], c
                t2 = -sip / zn if (zn > (eps * Rnorm) ** 2) else np.inf
                if t1 == np.inf and t2 == np.inf:
                    return x, it, 2, u
                if t2 == np.inf:
                    for c in order:
                        u[c] -= t1 * r[c]
                    up += t1
                    Ginv = drop(l, Ginv)
                    continue
                full = not (t1 < t2)
                t = t2 if full else t1
                x = x + t * z
                for c in order:
                    u[c] -= t * r[c]
                up += t
                if not full:
                    Ginv = drop(l, Ginv)
                    sip = slack(ip, x)
                    continue
                act[kp].append(ip); order.append(ip); u[ip] = up
                refresh(kp)
                if self.sm:
                    Ginv = Ginv + np.outer(y, y) / (a * zn)
                    self.sm_err = max(self.sm_err, np.abs(Ginv - np.linalg.inv(self.G(P))).max() / np.abs(Ginv).max())
                else:
                    Ginv = np.linalg.inv(self.G(P))
                Rnorm = max(Rnorm, np.sqrt(zn))
                break
        return x, it, status, u


def main():
    import wbc_quadruped_dob_amd  # noqa: F401  (path)
    from wbc_quadruped_dob_amd import synth
    from oracle import oracle_py, urdf_model
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    urdf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wbc_quadruped_dob_amd", "assets", "synthetic_quadruped.urdf")
    flat = urdf_model.load_urdf(urdf)
    orc = oracle_py.Oracle(flat)
    for cfg in (2, 3, 4):
        P = synth.default_params(observer_order=0)
        Bt = synth.make_batch(cfg, n, float(flat["mass"].sum()), rank=5)
        if cfg == 2:   # harder QPs: strong lateral demands (many active friction faces)
            Bt["w_des"][: n // 2, 0:2] += np.random.default_rng(1).uniform(-150, 150, (n // 2, 2))
        dyn = orc.dynamics(Bt["q"], Bt["v"])
        ref = orc.step(P, Bt["q"], Bt["v"], Bt["w_des"], Bt["vdot_des"], Bt["normals"], Bt["mu"], Bt["mask"])
        worst, it_diff, st_diff, sm_err, itmax = 0.0, 0, 0, 0.0, 0
        for i in range(n):
            d = (dyn["pf"][i].reshape(4, 3) - Bt["q"][i, :3])
            s = StructuredGI(np.asarray(P["S"], float), P["alpha"], int(Bt["mask"][i]), d, Bt["normals"][i].reshape(4, 3), Bt["mu"][i] * P["mu_scale"],
                             P["fn_min"], P["fn_max"], Bt["w_des"][i], tol=P["qp_tol"], max_iter=P["max_iter"])
            x, it, st, _ = s.solve()
            on = np.repeat([(int(Bt["mask"][i]) >> k) & 1 for k in range(4)], 3)
            err = np.abs(x * on - ref["f"][i]).max() / max(1.0, np.abs(ref["f"][i]).max())
            worst = max(worst, err)
            it_diff += int(it != ref["iters"][i])
            st_diff += int(st != ref["status"][i])
            sm_err = max(sm_err, s.sm_err)
            itmax = max(itmax, it)
        print("cfg %d: %d QPs, max rel force error vs oracle %.2e, iteration-count mismatches %d, status mismatches %d, max iters %d, "
              "Sherman-Morrison inverse vs refactorisation %.1e" % (cfg, n, worst, it_diff, st_diff, itmax, sm_err))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Simulation (numpy, CPU) for VERDICT r4 item 4(a): a different FINISHER for the stragglers of the cold fused tick.

The cold dual active-set solve pays one trip per active constraint; the tick lasts as long as the slowest of a workgroup's 16 QPs.  Proposal: after K
trips, jump -- guess the final active set from what the iterate shows (its active rows with positive multipliers plus the rows it violates, at most three
per foot), build the minimiser ON that set in one block set-up (cost: 1.7 trips, qp_struct16.hip.hpp's warm set-up), drop rows whose multipliers come out
negative and repeat once, and let the dual iteration go on from that S-pair; if no S-pair comes out, go on from where the iteration was.

Cost model (trip-equivalents): a dual trip = 1, a block set-up = 1.7 (measured ratio).  Reported: per QP and per 16-state workgroup (= what a
workgroup of the fused tick waits for), for the bench's configs[1] batch and the same batch with +-40 N lateral commands (tools/warm_loop.py).

usage: tools/sim_finisher.py [n_states] [K ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.structured_gi import StructuredGI  # noqa: E402

SETUP = 1.7


class Sim(StructuredGI):
    def run_partial(self, K):
        """the cold iteration, stopped at the first S-pair reached after >= K trips: (done, x, trips, u)"""
        self.max_iter_save = self.max_iter
        x, it, st, u = self._solve_until(K)
        return x, it, st, u

    def _solve_until(self, K):
        # the body of StructuredGI.solve (cold), returning at S-pair boundaries once `it >= K`
        a, B = self.alpha, self.B
        act = [[] for _ in range(4)]
        order = []
        u = {}
        from tools.structured_gi import proj_and_pinv
        P = [np.eye(3) for _ in range(4)]
        Np = [np.zeros((0, 3)) for _ in range(4)]
        Ginv = np.linalg.inv(self.G(P))
        x = np.concatenate([B[k].T @ (Ginv @ self.beta) for k in range(4)])
        eps = np.finfo(float).eps
        Rnorm, it = 1.0, 0
        slack = lambda c, xx: self.C[c].dot(xx[3 * (c // 6):3 * (c // 6) + 3]) - self.rhs[c]

        def refresh(k):
            P[k], Np[k] = proj_and_pinv([self.C[c] for c in act[k]])

        def drop(l):
            k = l // 6
            act[k].remove(l); order.remove(l); del u[l]
            refresh(k)
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
                return x, it, 0, dict(u)
            if it >= K:
                return x, it, -1, dict(u)      # -1: stopped at an S-pair, not finished
            sip, up, kp = smin, 0.0, ip // 6
            nplus = self.C[ip]
            while True:
                it += 1
                if it > self.max_iter:
                    return x, it, 1, dict(u)
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
                    return x, it, 2, dict(u)
                if t2 == np.inf:
                    for c in order:
                        u[c] -= t1 * r[c]
                    up += t1
                    Ginv = drop(l)
                    continue
                full = not (t1 < t2)
                t = t2 if full else t1
                x = x + t * z
                for c in order:
                    u[c] -= t * r[c]
                up += t
                if not full:
                    Ginv = drop(l)
                    sip = slack(ip, x)
                    continue
                act[kp].append(ip); order.append(ip); u[ip] = up
                refresh(kp)
                Ginv = np.linalg.inv(self.G(P))
                Rnorm = max(Rnorm, np.sqrt(zn))
                break

    def guess(self, x, u):
        """active rows with positive multipliers + the violated rows, most violated first, at most three rows per foot and never both bounds of a normal force"""
        keep = {c for c, val in u.items() if val > 0}
        viol = []
        for c in range(24):
            if not self.on[c // 6] or c in keep:
                continue
            s = self.C[c].dot(x[3 * (c // 6):3 * (c // 6) + 3]) - self.rhs[c]
            if s < -self.tol:
                viol.append((s, c))
        for s, c in sorted(viol):
            k = c // 6
            rows = [d for d in keep if d // 6 == k]
            if len(rows) >= 3:
                continue
            if c % 6 >= 4 and any(d % 6 >= 4 for d in rows):
                continue
            keep.add(c)
        return sorted(keep)


def trips_with_finisher(s, K):
    """(trip-equivalents, used): the cold iteration for K trips, the jump, the rest of the iteration"""
    x, it, st, u = s.run_partial(K)
    if st != -1:
        return float(it), "finished_before"
    A = s.guess(x, u)
    if set(A) == set(u.keys()):
        return None, "no_new_rows"
    cost = float(it)
    ws = s.warm_setup(A)
    cost += SETUP
    if isinstance(ws, list):
        A = [c for c in A if c not in ws]
        ws = s.warm_setup(A)
        cost += SETUP
    if ws is None or isinstance(ws, list):
        return cost, "rejected"       # the caller adds the rest of the COLD path
    x2, it2, st2, _ = s.solve(warm=A)   # (repeats the accepted set-up: not counted twice)
    assert s.warm_used
    return cost + it2, "jumped"


def main():
    from wbc_quadruped_dob_amd import synth
    from oracle import oracle_py, urdf_model
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    Ks = [int(a) for a in sys.argv[2:]] or [2, 3, 4, 5]
    urdf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wbc_quadruped_dob_amd", "assets", "synthetic_quadruped.urdf")
    flat = urdf_model.load_urdf(urdf)
    orc = oracle_py.Oracle(flat)
    P = synth.default_params(observer_order=0)
    for tag, hard in (("configs[1] batch", False), ("configs[1] batch + 40 N lateral commands", True)):
        Bt = synth.make_batch(2, n, float(flat["mass"].sum()), rank=0)
        if hard:
            Bt["w_des"][:, 0:2] += np.random.default_rng(1).uniform(-40, 40, (n, 2))
        dyn = orc.dynamics(Bt["q"], Bt["v"])
        mk = lambda i: Sim(np.asarray(P["S"], float), P["alpha"], int(Bt["mask"][i]), dyn["pf"][i].reshape(4, 3) - Bt["q"][i, :3], Bt["normals"][i].reshape(4, 3),
                           Bt["mu"][i] * P["mu_scale"], P["fn_min"], P["fn_max"], Bt["w_des"][i], tol=P["qp_tol"], max_iter=P["max_iter"])
        cold = np.zeros(n)
        for i in range(n):
            _, it, st, _ = mk(i).solve()
            cold[i] = it
        wg = lambda v: v.reshape(-1, 16).max(axis=1)
        print("== %s, %d QPs: cold trips mean %.2f, max %d; per 16-state workgroup: mean of max %.2f, p90 %.1f, max %.1f" % (
            tag, n, cold.mean(), cold.max(), wg(cold).mean(), np.percentile(wg(cold), 90), wg(cold).max()))
        for K in Ks:
            tot = cold.copy()
            kinds = {}
            for i in range(n):
                if cold[i] <= K:
                    continue
                c, kind = trips_with_finisher(mk(i), K)
                kinds[kind] = kinds.get(kind, 0) + 1
                if kind == "jumped":
                    tot[i] = c
                elif kind == "rejected":
                    tot[i] = cold[i] + (c - K if c else 0) - 0 * K     # set-ups wasted on top of the cold path
                    tot[i] = cold[i] + (c - min(K, cold[i])) if c else cold[i]
            hardq = cold > K
            print("   K = %d: %4d QPs go past K (%s); their trip-equivalents %.2f -> %.2f (max %.1f -> %.1f); per workgroup: mean of max %.2f -> %.2f, "
                  "p90 %.1f -> %.1f, max %.1f -> %.1f" % (
                      K, int(hardq.sum()), ", ".join("%s %d" % kv for kv in sorted(kinds.items())), cold[hardq].mean() if hardq.any() else 0, tot[hardq].mean() if hardq.any() else 0,
                      cold.max(), tot.max(), wg(cold).mean(), wg(tot).mean(), np.percentile(wg(cold), 90), np.percentile(wg(tot), 90), wg(cold).max(), wg(tot).max()))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Simulation (numpy, CPU): a block PRE-START of the cold dual iteration.  At the unconstrained minimum x0 the violated rows are visible; guess one row per
foot (variant 1), up to two / three per foot with different directions (2 / 3), build the minimiser on that set in one block set-up (1.7 trip-equivalents,
the warm set-up of qp_struct16.hip.hpp), drop rows with negative multipliers, and let the dual iteration go on -- only for states whose guess has at least
`min_rows` rows.  Result (profiles/r05_sim_prestart.log): the per-workgroup mean of the maximum does not improve (6.28 -> 6.24 ... 6.74): the rows that
cost the hard states their trips become violated only as others are enforced.  Not built.   usage: tools/sim_prestart.py [n_states]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.sim_finisher import Sim, SETUP

def guess0(s, variant):
    # violations at the unconstrained minimum x0
    x, it, st, u = s.run_partial(0)
    if st != -1:
        return None
    viol = {}
    for c in range(24):
        if not s.on[c // 6]:
            continue
        sl = s.C[c].dot(x[3 * (c // 6):3 * (c // 6) + 3]) - s.rhs[c]
        if sl < -s.tol:
            viol[c] = sl
    rows = []
    for k in range(4):
        vk = sorted((sl, c) for c, sl in viol.items() if c // 6 == k)
        if not vk:
            continue
        if variant == 1:
            rows.append(vk[0][1])
        else:
            # up to two friction rows with different tangent direction + most violated first; row ids within foot: 0,1 = A rows (mu n - t1/t2), 2 = fn_min, 3 = fn_max, 4,5 = B rows (mu n + t1/t2)
            taken_dir = set(); cnt = 0
            for sl, c in vk:
                j = c % 6
                d = {0: 't1', 4: 't1', 1: 't2', 5: 't2', 2: 'n', 3: 'n'}[j]
                if d in taken_dir:
                    continue
                taken_dir.add(d); rows.append(c); cnt += 1
                if cnt >= (2 if variant == 2 else 3):
                    break
    return sorted(rows)

def trips_prestart(mk, i, variant, min_rows, cold_i):
    s = mk(i)
    A = guess0(s, variant)
    if A is None or len(A) < min_rows:
        return cold_i, "cold"
    s = mk(i)
    cost = 0.0
    ws = s.warm_setup(A); cost += SETUP
    if isinstance(ws, list):
        A = [c for c in A if c not in ws]
        ws = s.warm_setup(A); cost += SETUP
    if ws is None or isinstance(ws, list):
        return cold_i + cost, "rejected"
    x2, it2, st2, _ = s.solve(warm=A)
    return cost + it2, "pre"

def main():
    from wbc_quadruped_dob_amd import synth
    from oracle import oracle_py, urdf_model
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    urdf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wbc_quadruped_dob_amd", "assets", "synthetic_quadruped.urdf")
    flat = urdf_model.load_urdf(urdf)
    orc = oracle_py.Oracle(flat)
    P = synth.default_params(observer_order=0)
    for tag, hard in (("configs[1] batch", False), ("+40 N", True)):
        Bt = synth.make_batch(2, n, float(flat["mass"].sum()), rank=0)
        if hard:
            Bt["w_des"][:, 0:2] += np.random.default_rng(1).uniform(-40, 40, (n, 2))
        dyn = orc.dynamics(Bt["q"], Bt["v"])
        mk = lambda i: Sim(np.asarray(P["S"], float), P["alpha"], int(Bt["mask"][i]), dyn["pf"][i].reshape(4, 3) - Bt["q"][i, :3], Bt["normals"][i].reshape(4, 3),
                           Bt["mu"][i] * P["mu_scale"], P["fn_min"], P["fn_max"], Bt["w_des"][i], tol=P["qp_tol"], max_iter=P["max_iter"])
        cold = np.zeros(n)
        for i in range(n):
            _, it, st, _ = mk(i).solve(); cold[i] = it
        wg = lambda v: v.reshape(-1, 16).max(axis=1)
        print("== %s: cold mean %.2f max %d; per WG mean of max %.2f p90 %.1f max %.1f; hist %s" % (tag, cold.mean(), cold.max(), wg(cold).mean(), np.percentile(wg(cold), 90), wg(cold).max(), np.bincount(cold.astype(int)).tolist()))
        for variant in (1, 2, 3):
            for min_rows in (2, 3, 4):
                tot = cold.copy(); kinds = {}
                for i in range(n):
                    c, kind = trips_prestart(mk, i, variant, min_rows, cold[i])
                    kinds[kind] = kinds.get(kind, 0) + 1
                    tot[i] = c
                print("  variant %d min_rows %d: %s  mean %.2f -> %.2f  max %.1f -> %.1f  per WG mean of max %.2f -> %.2f  p90 %.1f -> %.1f  max %.1f -> %.1f" % (
                    variant, min_rows, kinds, cold.mean(), tot.mean(), cold.max(), tot.max(), wg(cold).mean(), wg(tot).mean(), np.percentile(wg(cold), 90), np.percentile(wg(tot), 90), wg(cold).max(), wg(tot).max()))

if __name__ == "__main__":
    main()

"""Soak run on the GPU box: many random batches through every small-batch dispatch variant, each compared with the two-kernel /
per-tick-launch path of the same library (same device functions, other synchronisation) and, every few cases, with the CPU
oracle.  The fused tick and the persistent rollouts hand data between wavefronts through LDS counters; a synchronisation
mistake there would show up as a rare, timing-dependent mismatch, which the fixed-seed unit tests could miss.
usage: python tools/soak.py [cases] [seed] [f64|f32]     (prints one summary line; exit status 1 on any mismatch;
fp32: single ticks only, rounding-level tolerances, a state whose status flips between the variants is counted, not failed)"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from tests.util import relerr
from tests.test_gpu_parity import _solver, _run_step, _gpu_rollout
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
from oracle import oracle_py, urdf_model

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
DT = sys.argv[3] if len(sys.argv) > 3 else "f64"
TOL = 1e-10 if DT == "f64" else 2e-2   # (fp64: rounding of two pivot sequences to one solution -- since the observer-on fused tick starts its QP on b~ and moves the solution to b, up to 1.1e-11 was seen)
flips = 0
rng = np.random.default_rng(seed)
gm = W.Model.from_urdf(W.SYNTHETIC_URDF)
orc = oracle_py.Oracle(urdf_model.load_urdf(W.SYNTHETIC_URDF))


def solver_with(options, **kw):
    """options: wbc_solver_options overrides of this variant"""
    return _solver(gm, options=options, **kw)


bad, worst, worst_big, t0 = [], 0.0, 0.0, time.time()
for c in range(cases):
    obs = int(rng.integers(0, 3))
    cfg = int(rng.choice([2, 3, 4]))
    n = int(rng.choice([1, 7, 16, 17, 100, 1000, 2048, 4096, 5000, 8192, 11000, 12288][: (11 if obs == 0 else 12)]))   # (the one-launch tick: up to 11 264 states, observer on 12 288)
    if rng.random() < 0.3:
        n = int(rng.integers(1, 4097))
    only = os.environ.get("SOAK_ONLY")   # replay ONE case of a seed (the others only consume their random draws): SOAK_ONLY=1607 python tools/soak.py 3000 101 f64
    if only is not None and c != int(only):
        if c % 7 == 3:
            rng.integers(12288, 60000)
            if DT == "f32":
                rng.integers(12288, 140000)
            rng.choice([0, 32, 44, 64, 128] if DT == "f64" else [0, 64, 96, 128])
        elif not (c % 3 != 2 or DT == "f32"):
            rng.integers(2, 12); rng.choice([0, 0, 4, 16])
        continue
    big = c % 7 == 3   # a larger batch: the tiled QP kernel (dealt by predicted work; the predictor hands G^-1 / x0 to the solver or finishes the state) and the per-lane kernel against the one-wave kernel
    if big:
        n = int(rng.integers(12288, 60000))
    B = synth.make_batch(cfg, n, gm.total_mass, rank=1000 + c)
    integ0 = orc.dynamics(B["q"], B["v"], nthreads=8)["p"] if obs else None
    z = lambda: (None if integ0 is None else integ0.copy(), None if integ0 is None else np.zeros((n, 18)))
    if big:
        res = {}
        tiles = [0, 32, 44, 64, 128] if DT == "f64" else [0, 64, 96, 128]
        if DT == "f32":
            n = int(rng.integers(12288, 140000))     # (fp32: past 65 536 states the leaner fp32 body runs the tiles)
            B = synth.make_batch(cfg, n, gm.total_mass, rank=1000 + c)
            integ0 = orc.dynamics(B["q"], B["v"], nthreads=8)["p"] if obs else None
        zz = lambda: tuple(None if t is None else t.astype(np.float32 if DT == "f32" else np.float64) for t in z())
        variants = [("tiled", {"qp_tile": int(rng.choice(tiles)), "qp_lane": -1}), ("plain", {"qp_tile": -1, "qp_lane": -1}), ("lane", {"qp_lane": 1})]
        # round 6: the tile tick (one launch: roles + staged QP tile per workgroup) where it exists -- fp32 with the observer on (even batches), fp64 with it on or off; matrix outputs
        ttick = bool(c % 2) and ((DT == "f32" and obs > 0 and n % 2 == 0) or DT == "f64")
        if ttick:
            variants.append(("ttick", {"tile_tick": 1, "fused_max": 0}))
        for tag, opt in variants:
            s, P = solver_with(opt, obs=obs, max_batch=n, dtype=DT)
            if tag == "ttick" and s.plan_tick(n, want_mats=True)["fused"] != 2:
                bad.append((c, "tile tick not planned", n, obs, cfg))
            a1 = _run_step(torch, s, B, DT, *zz(), want_mats=bool(c % 2))
            a2 = _run_step(torch, s, B, DT, a1.get("integ"), a1.get("r"), want_mats=bool(c % 2))
            res[tag] = (a1, a2)
        for vtag, i in [(t_, i_) for t_ in (("tiled", "ttick") if ttick else ("tiled",)) for i_ in (0, 1)]:     # same solver body, started from the predictor's numbers: status equal, iteration counts equal up to near-ties, values to rounding
            a, b = res[vtag][i], res["plain"][i]
            same = a["status"] == b["status"]
            if not same.all():
                if DT == "f64" or same.mean() < 0.995:
                    bad.append((c, vtag + " status", n, obs, cfg))
                flips += int((~same).sum())
            # (the standing batch is symmetric: equally violated rows on different feet are exact ties in one kernel and rounding-level
            #  near-ties in the other -- 0.1-0.15 % of its states take another, equally valid pivot sequence to the same solution; fp32 batches
            #  past 65 536 states run the tiles on the fp32-arithmetic body: 2-3 %)
            if np.mean(a["iters"] != b["iters"]) > (5e-3 if DT == "f64" else 5e-2):
                bad.append((c, vtag + " iters", n, obs, cfg, float(np.mean(a["iters"] != b["iters"]))))
            for k in b:
                if k in ("status", "iters"):
                    continue
                x, y = (a[k][same], b[k][same]) if a[k].shape[0] == same.shape[0] else (a[k], b[k])
                e = relerr(x, y)
                worst_big = max(worst_big, e)
                if not e < (1e-10 if DT == "f64" else TOL):
                    bad.append((c, vtag + " " + k, n, obs, cfg, e))
        if DT == "f32":
            continue
        for i in (0, 1):   # per-lane semismooth Newton + hand-over list: another algorithm, same unique solution
            a, b = res["lane"][i], res["plain"][i]
            if not np.array_equal(a["status"], b["status"]):
                bad.append((c, "lane status", n, obs, cfg))
            for k in b:
                if k in ("status", "iters"):
                    continue
                e = relerr(a[k], b[k])
                worst = max(worst, e)
                if not e < 1e-9:
                    bad.append((c, "lane " + k, n, obs, cfg, e))
        # wbc_step_batch_warm at this size: tick 1 cold (reports the sets), tick 2 from them -- the planner's choice, the warm per-lane pair and the warm
        # one-wavefront kernel forced -- against the cold second tick of the one-wave kernel
        from tests.util import to_dev, to_host
        for wtag, wopt in (("warm default", {}), ("warm per-lane", {"qp_lane": 1}), ("warm one-wave", {"qp_lane": -1, "qp_tile": -1})):
            s, P = solver_with(wopt, obs=obs, max_batch=n, dtype=DT)
            dv = lambda k_: to_dev(B[k_], torch, torch.float64)
            ins_ = [dv(k_) for k_ in ("q", "v", "w_des", "vdot_des", "normals", "mu")]
            mk_ = torch.from_numpy(np.ascontiguousarray(B["mask"])).to(torch.int32).cuda()
            zw = tuple(None if t is None else to_dev(t, torch, torch.float64) for t in z())
            tp_, fp_ = dv("tau_prev"), dv("f_prev")
            o1 = s.step(*ins_, mk_, tp_, fp_, zw[0], zw[1], want_mats=bool(c % 2), warm=True)
            o2 = s.step(*ins_, mk_, tp_, fp_, zw[0], zw[1], want_mats=bool(c % 2), active_in=o1["active"].clone(), out={k_: v_ for k_, v_ in o1.items() if k_ != "active"})
            torch.cuda.synchronize()
            b2 = res["plain"][1]
            if not np.array_equal(o2["status"].cpu().numpy(), b2["status"]):
                bad.append((c, wtag + " status", n, obs, cfg))
            for k_ in ("tau", "f"):
                e = relerr(to_host(o2[k_]), b2[k_])
                worst = max(worst, e)
                if not e < 1e-9:
                    bad.append((c, wtag + " " + k_, n, obs, cfg, e))
            del s
        if c % 2:
            ig, r = z()
            ref = orc.step(P, B["q"], B["v"], B["w_des"], B["vdot_des"], B["normals"], B["mu"], B["mask"], B["tau_prev"], B["f_prev"], ig, r, nthreads=8)
            e = max(relerr(res["tiled"][0]["tau"], ref["tau"]), relerr(res["tiled"][0]["f"], ref["f"]))
            worst = max(worst, e)
            if not e < 1e-9 or not np.array_equal(res["tiled"][0]["status"], ref["status"]):
                bad.append((c, "oracle (tiled)", n, obs, cfg, e))
    elif c % 3 != 2 or DT == "f32":   # ---- one tick: fused vs two-kernel, two consecutive ticks (observer state carried)
        res = {}
        for tag, env in (("fused", {}), ("two", {"fused_max": 0})):
            s, P = solver_with(env, obs=obs, max_batch=n, dtype=DT)
            zz = tuple(None if t is None else t.astype(np.float32 if DT == "f32" else np.float64) for t in z())
            a1 = _run_step(torch, s, B, DT, *zz, want_mats=bool(c % 2))
            a2 = _run_step(torch, s, B, DT, a1.get("integ"), a1.get("r"), want_mats=bool(c % 2))
            res[tag] = (a1, a2)
        if c % 2 == 0:   # wbc_step_batch_warm: tick 1 cold (reports its active sets), tick 2 from them -- against the cold tick 2 of the fused variant;
        #   default plan (fused at these sizes) and the warm per-lane pair forced
            for wtag, wopt in (("", {}), (" (per-lane)", {"qp_lane": 1, "fused_max": 0})):
                s, P = solver_with(wopt, obs=obs, max_batch=n, dtype=DT)
                td_ = torch.float64 if DT == "f64" else torch.float32
                from tests.util import to_dev, to_host
                dv = lambda k_: to_dev(B[k_], torch, td_)
                ins_ = [dv(k_) for k_ in ("q", "v", "w_des", "vdot_des", "normals", "mu")]
                mk_ = torch.from_numpy(np.ascontiguousarray(B["mask"])).to(torch.int32).cuda()
                zz = tuple(None if t is None else to_dev(t.astype(np.float32 if DT == "f32" else np.float64), torch, td_) for t in z())
                tp_, fp_ = dv("tau_prev"), dv("f_prev")
                o1 = s.step(*ins_, mk_, tp_, fp_, zz[0], zz[1], want_mats=bool(c % 2), warm=True)
                o2 = s.step(*ins_, mk_, tp_, fp_, zz[0], zz[1], want_mats=bool(c % 2), active_in=o1["active"].clone(), out={k_: v_ for k_, v_ in o1.items() if k_ != "active"})
                torch.cuda.synchronize()
                b2 = res["fused"][1]
                st_w = o2["status"].cpu().numpy()
                same = st_w == b2["status"]
                if not same.all():
                    if DT == "f64" or same.mean() < 0.995:
                        bad.append((c, "warm tick status" + wtag, n, obs, cfg))
                    flips += int((~same).sum())
                for k_ in ("tau", "f"):
                    e = relerr(to_host(o2[k_])[same], b2[k_][same]) if same.any() else 0.0
                    worst = max(worst, e)
                    if not e < (1e-9 if DT == "f64" else TOL):
                        bad.append((c, "warm tick " + k_ + wtag, n, obs, cfg, e))
        if obs == 0 and c % 2 and n >= 64:   # round 6: the one-launch tick as 32-state workgroups (fused_pair_kernel; ragged batches: a tail workgroup) -- same bodies: bit for bit
            pr = {}
            for ptag, popt in (("pair", {"fused_pair": 1}), ("one", {"fused_pair": -1, "fused_max": 65536})):
                s, P = solver_with(popt, obs=0, max_batch=n, dtype=DT)
                pr[ptag] = _run_step(torch, s, B, DT, want_mats=True)
            for k in pr["pair"]:
                if not np.array_equal(pr["pair"][k], pr["one"][k], equal_nan=True):
                    bad.append((c, "fused pair " + k, n, obs, cfg))
        if c % 5 == 0 and DT == "f64":   # the same launch again, several times: results must be bit-identical run to run
            s, P = solver_with({}, obs=obs, max_batch=n)
            first = _run_step(torch, s, B, "f64", *z(), want_mats=True)
            for _ in range(6):
                again = _run_step(torch, s, B, "f64", *z(), want_mats=True)
                for k in first:
                    if not np.array_equal(first[k], again[k], equal_nan=True):
                        bad.append((c, "nondeterministic " + k, n, obs, cfg))
        for i in (0, 1):
            a, b = res["fused"][i], res["two"][i]
            same = a["status"] == b["status"]
            if not same.all():
                if DT == "f64" or same.mean() < 0.995:
                    bad.append((c, "status", n, obs, cfg))
                flips += int((~same).sum())
            for k in a:
                if k in ("status", "iters"):
                    continue
                x, y = (a[k][same], b[k][same]) if a[k].shape[0] == same.shape[0] else (a[k], b[k])
                e = relerr(x, y) if x.size else 0.0
                worst = max(worst, e)
                if not e < TOL:
                    bad.append((c, k, n, obs, cfg, e))
        if c % 9 == 0 and DT == "f64":
            ig, r = z()
            ref = orc.step(P, B["q"], B["v"], B["w_des"], B["vdot_des"], B["normals"], B["mu"], B["mask"], B["tau_prev"], B["f_prev"], ig, r, nthreads=8)
            e = max(relerr(res["fused"][0]["tau"], ref["tau"]), relerr(res["fused"][0]["f"], ref["f"]))
            if not e < 1e-9:
                bad.append((c, "oracle", n, obs, cfg, e))
    else:            # ---- rollout: persistent vs per-tick launches
        n = min(n, 2048)
        B = synth.make_batch(cfg, n, gm.total_mass, rank=1000 + c)
        integ0 = orc.dynamics(B["q"], B["v"], nthreads=8)["p"] if obs else None
        H = int(rng.integers(2, 12))
        tau_ext = np.zeros((n, 18)); tau_ext[:, 0:3] = B["push"] if cfg > 2 else 5.0
        res = {}
        spw = int(rng.choice([0, 0, 4, 16]))   # (round 5: the persistent kernel's 4- and 16-state workgroups, whatever the batch size)
        if c % 4 == 1:   # ---- the planner in the loop: persistent against per-tick launches
            from tests import test_gpu_reference as tr
            plan = synth.make_plan(B, rank=1000 + c)
            rt = {}
            for tag, env in (("persistent", {"rollout_spw": spw} if spw else {}), ("per_tick", {"rollout_persistent": 0})):
                s, P, G = tr._solver(gm, obs=obs, max_batch=n, options=env)
                rt[tag] = tr._gpu_tracking(torch, s, H, B, plan, tau_ext, None if integ0 is None else integ0.copy(), np.zeros((n, 18)) if obs else None)
            a, b = rt["persistent"], rt["per_tick"]
            ok = (a["status"] == 0) & (b["status"] == 0)   # (far-from-plan random states may saturate a force box: such rows amplify rounding)
            if not np.array_equal(a["status"], b["status"]):
                bad.append((c, "tracking status", n, obs, cfg, spw))
            for k in a:
                if k == "status" or not ok.any():
                    continue
                e = relerr(a[k][ok], b[k][ok])
                worst = max(worst, e)
                if not e < 1e-8:
                    bad.append((c, "tracking " + k, n, obs, cfg, H, spw, e))
            continue
        for tag, env in (("persistent", {"rollout_spw": spw} if spw else {}), ("per_tick", {"rollout_persistent": 0}), ("cold", {"rollout_warm": 0})):   # (the first two start every tick after the first from the previous tick's active set)
            s, P = solver_with(env, obs=obs, max_batch=n)
            res[tag] = _gpu_rollout(torch, s, P, H, B, tau_ext, None if integ0 is None else integ0.copy(), np.zeros((n, 18)) if obs else None)
        if only is not None:   # the replayed case in detail: every output of the three variants against each other
            print("case %d: n %d obs %d cfg %d H %d spw %d" % (c, n, obs, cfg, H, spw))
            for x_, y_ in (("persistent", "per_tick"), ("persistent", "cold"), ("cold", "per_tick")):
                for k in res[x_]:
                    d_ = np.abs(res[x_][k].astype(np.float64) - res[y_][k].astype(np.float64)).reshape(n, -1)
                    if d_.max() > 0:
                        print("  %-10s vs %-9s %-8s max |diff| %.3e (rel %.2e), states %s" % (x_, y_, k, d_.max(), relerr(res[x_][k], res[y_][k]) if k not in ("status", "iters") else 0.0, np.nonzero(d_.max(axis=1) > 1e-3 * d_.max())[0][:16]))
        a, b = res["persistent"], res["cold"]
        if not np.array_equal(a["status"], b["status"]):
            bad.append((c, "warm rollout status", n, obs, cfg))
        for k in a:
            if k in ("status", "iters"):   # (iters: a warm-started tick and a cold one reach the same solution in different numbers of iterations)
                continue
            e = relerr(a[k], b[k])
            worst = max(worst, e)
            if not e < 1e-9:
                bad.append((c, "warm rollout " + k, n, obs, cfg, H, e))
        a, b = res["persistent"], res["per_tick"]
        if not np.array_equal(a["status"], b["status"]):
            bad.append((c, "rollout status", n, obs, cfg))
        for k in a:
            if k == "status":
                continue
            if k == "iters":   # (the last tick's: last-bit differences of the state flip a near-tie between two violated rows now and then -- a share of the states, not a bound)
                if np.sum(a[k] != b[k]) > max(1, 1e-2 * a[k].size):   # (one state of a small batch is not a share)
                    bad.append((c, "rollout iters differ in %.3f of the states" % np.mean(a[k] != b[k]), n, obs, cfg, H))
                continue
            e = relerr(a[k], b[k])
            worst = max(worst, e)
            if not e < 1e-8:   # (the persistent kernel's roles normalise with rsqrt_fast, the per-tick kernels with 1 / sqrt: 1-2 ulp per tick, compounding through
                               #  the dynamics over up to 11 ticks -- up to 1e-9 seen; the warm-against-cold comparison above runs the same kernels: 1e-9)
                bad.append((c, "rollout " + k, n, obs, cfg, H, e))
print("soak: %d cases (%s), seed %d, %.0f s, worst relative difference between dispatch variants %.2e (tiled vs one-wave QP kernel: %.2e), status flips %d, mismatches: %d %s"
      % (cases, DT, seed, time.time() - t0, worst, worst_big, flips, len(bad), bad[:10]))
sys.exit(1 if bad else 0)

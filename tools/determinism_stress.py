"""Run-to-run bit identity of every tick family: the same inputs through the same solver `reps` times, every output of every launch compared with the first launch's bits.
The kernels hand data between wavefronts through LDS flags; an ordering mistake shows up as a rare, timing-dependent difference (round 6: the rollout workgroups' pf store,
found by tools/soak.py).   usage: python tools/determinism_stress.py [reps]      exit status 1 on any difference"""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from tests.test_gpu_parity import _solver, _run_step, _gpu_rollout
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
from oracle import oracle_py, urdf_model

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
gm = W.Model.from_urdf(W.SYNTHETIC_URDF)
orc = oracle_py.Oracle(urdf_model.load_urdf(W.SYNTHETIC_URDF))
bad = []
# (dtype, observer order, config of the synthetic batch, N, options, label)
ticks = [("f64", 0, 2, 4096, {}, "one-launch tick"), ("f64", 1, 3, 4096, {}, "one-launch tick, observer"), ("f64", 2, 3, 3000, {}, "one-launch tick, observer order 2"),
         ("f64", 0, 2, 6000, {}, "pair tick (ragged)"), ("f64", 0, 2, 8192, {}, "pair tick"), ("f64", 1, 3, 12000, {}, "tile tick fp64 observer"),
         ("f64", 0, 2, 20000, {}, "tile tick fp64"), ("f32", 1, 4, 12290, {}, "tile tick fp32"), ("f32", 1, 4, 40000, {"tile_tick": -1}, "two-launch fp32 (sweep_obs / staged tiles)"),
         ("f64", 1, 3, 30000, {"tile_tick": -1}, "two-launch fp64 (observer + sweep, tiles)"), ("f64", 0, 2, 120000, {}, "per-lane pair + list"), ("f32", 0, 2, 4000, {}, "one-launch tick fp32"), ("f32", 0, 2, 9001, {}, "pair tick fp32 (ragged)")]
for dtype, obs, cfg, n, opt, label in ticks:
    nd = np.float64 if dtype == "f64" else np.float32
    B = synth.make_batch(cfg, n, gm.total_mass, rank=5)
    s, P = _solver(gm, dtype=dtype, obs=obs, max_batch=n, options=opt)
    integ0 = orc.dynamics(B["q"], B["v"], nthreads=8)["p"].astype(nd) if obs else None
    z = lambda: (None if integ0 is None else integ0.copy(), None if integ0 is None else np.zeros((n, 18), nd))
    first = _run_step(torch, s, B, dtype, *z(), want_mats=True)
    diff = 0
    for _ in range(reps if n <= 20000 else max(20, reps // 5)):
        again = _run_step(torch, s, B, dtype, *z(), want_mats=True)
        for k in first:
            if not np.array_equal(first[k], again[k], equal_nan=True):
                diff += 1
                bad.append((label, k))
    print("%-46s n %-7d plan %s: %s" % (label, n, s.plan_tick(n), "identical" if not diff else "%d DIFFERENCES" % diff), flush=True)
for obs, cfg, n, H, opt, label in ((1, 3, 1024, 5, {}, "rollouts, 4-state workgroups"), (2, 2, 16, 5, {}, "16 rollouts"), (1, 3, 2048, 4, {}, "rollouts, 16-state workgroups"),
                                   (0, 2, 512, 6, {}, "rollouts, observer off"), (1, 3, 700, 5, {"rollout_warm": 0}, "rollouts, cold")):
    B = synth.make_batch(cfg, n, gm.total_mass, rank=6)
    tau_ext = np.zeros((n, 18)); tau_ext[:, 0:3] = B["push"] if cfg > 2 else 5.0
    integ0 = orc.dynamics(B["q"], B["v"], nthreads=8)["p"] if obs else None
    s, P = _solver(gm, obs=obs, max_batch=n, options=opt)
    z = lambda: (None if integ0 is None else integ0.copy(), np.zeros((n, 18)) if obs else None)
    first = _gpu_rollout(torch, s, P, H, B, tau_ext, *z())
    diff = 0
    for _ in range(reps):
        again = _gpu_rollout(torch, s, P, H, B, tau_ext, *z())
        for k in first:
            if not np.array_equal(first[k], again[k], equal_nan=True):
                diff += 1
                bad.append((label, k))
    print("%-46s n %-7d H %d: %s" % (label, n, H, "identical" if not diff else "%d DIFFERENCES" % diff), flush=True)
print("determinism stress: %d launches per family, differences: %d %s" % (reps, len(bad), sorted(set(bad))[:10]))
sys.exit(1 if bad else 0)

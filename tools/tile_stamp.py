"""Diagnostic: timeline of qp_tile_kernel's workgroups (needs the -DWBC_TILE_STAMP build:
   make -C wbc_quadruped_dob_amd/csrc -j8 LIBDIR=../lib_tstamp EXTRA=-DWBC_TILE_STAMP;
   WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_tstamp/libwbc_hip.so python tools/tile_stamp.py [n] [f32|f64] [config]).
In that build the `iters` output of a tile carries, per wavefront, shader-clock stamps relative to the workgroup's entry:
predictor barrier reached, sort done, first group done, group loop left, groups taken (+ wavefront 0: predictor decision
done), the 100 MHz wall clock at entry and exit, and the number of states the tile had to solve."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 4
obs = 1 if cfg >= 3 else 0
m = W.Model.from_urdf(W.SYNTHETIC_URDF)
P = synth.default_params(observer_order=obs, dtype=dtype)
s = W.Solver(m, W.Params.from_dict(P), dtype=dtype, max_batch=n)
B = synth.make_batch(cfg, n, m.total_mass)
td = torch.float32 if dtype == "f32" else torch.float64
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
inp = [dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu")]
mask = torch.from_numpy(B["mask"]).cuda()
extra = []
if obs:
    ig = s.dynamics(inp[0], inp[1], want=("p",))["p"]
    extra = [dev(B["tau_prev"]), dev(B["f_prev"]), ig, torch.zeros_like(ig)]
plan = s.plan_tick(n)
warm = int(sys.argv[6]) if len(sys.argv) > 6 else 5      # ticks before the stamped one (the observer state moves from tick to tick: bench.py's steady state is reached after ~1 000)
for _ in range(warm):
    out = s.step(*inp, mask, *extra, want_mats=True)
torch.cuda.synchronize()
it = out["iters"].cpu().numpy().astype(np.int64)
tile = int(sys.argv[4]) if len(sys.argv) > 4 else 64
nw = int(sys.argv[5]) if len(sys.argv) > 5 else 4
st = it[: (n // tile) * tile].reshape(-1, tile)[:, :8 * nw].reshape(-1, nw, 8)      # [tiles, wavefront, slot]
wall0, wall1 = st[:, 0, 5], st[:, :, 6].max(1)
span_us = ((wall1 - wall0) & 0x7FFFFFFF) * 0.01
cyc_end = st[:, :, 3].max(1)
ghz = np.median(cyc_end / np.maximum(span_us, 1e-3)) * 1e-3
us = lambda c: c / (ghz * 1e3)
print("ticks before the stamped one:", warm, " plan:", plan, " tiles %d, shader clock ~%.2f GHz (cycle stamps / wall-clock span)" % (st.shape[0], ghz))
print("kernel-wide: first entry -> last exit %.2f us; entry spread %.2f us" % (((wall1.max() - wall0.min()) & 0x7FFFFFFF) * 0.01, ((wall0.max() - wall0.min()) & 0x7FFFFFFF) * 0.01))
pp = ((st[:, 0, 4] >> 8) << 4)
ng = st[:, :, 4] & 0xFF
rows = [("predictor decision done (wavefront 0)", pp), ("predictor + finish done (wavefront 0 at barrier)", st[:, 0, 0]), ("sort done", st[:, 0, 1]),
        ("first group done: earliest wavefront", np.where(ng > 0, st[:, :, 2], 1 << 40).min(1)), ("first group done: latest wavefront", np.where(ng > 0, st[:, :, 2], 0).max(1)),
        ("group loop left: earliest wavefront", st[:, :, 3].min(1)), ("group loop left: latest wavefront = tile done", cyc_end)]
for nm, c in rows:
    c = c.astype(float)
    print("  %-52s median %+6.2f us   p10 %+6.2f   p90 %+6.2f   max %+6.2f" % (nm, us(np.median(c)), us(np.percentile(c, 10)), us(np.percentile(c, 90)), us(c.max())))
print("  states to solve per tile: mean %.1f (min %d, max %d); groups per wavefront: mean %.2f, max %d; tiles whose busiest wavefront took 1 / 2 / 3+ groups: %d / %d / %d"
      % (st[:, 0, 7].mean(), st[:, 0, 7].min(), st[:, 0, 7].max(), ng.mean(), ng.max(), (ng.max(1) == 1).sum(), (ng.max(1) == 2).sum(), (ng.max(1) >= 3).sum()))
for k in range(0, 4):
    sel = ng.max(1) == k
    if sel.any():
        print("     busiest wavefront took %d groups: %4d tiles, tile done median %.2f us (sort done %.2f)" % (k, sel.sum(), us(np.median(cyc_end[sel])), us(np.median(st[sel, 0, 1]))))

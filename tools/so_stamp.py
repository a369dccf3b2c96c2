"""Diagnostic: where the workgroups of sweep_obs_kernel land and how long each role runs there (needs the -DWBC_SO_STAMP build:
   make -C wbc_quadruped_dob_amd/csrc -j8 LIBDIR=../lib_sostamp EXTRA=-DWBC_SO_STAMP;
   WBC_LIB=$PWD/wbc_quadruped_dob_amd/lib_sostamp/libwbc_hip.so python tools/so_stamp.py [n] [f32|f64]).
In that build `pf` carries per workgroup: role (0 sweep, 1 observer), 100 MHz wall clock at entry / exit, HW_ID, XCC_ID."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import wbc_quadruped_dob_amd as W
from wbc_quadruped_dob_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
m = W.Model.from_urdf(W.SYNTHETIC_URDF)
P = synth.default_params(observer_order=1, dtype=dtype)
s = W.Solver(m, W.Params.from_dict(P), dtype=dtype, max_batch=n)
B = synth.make_batch(4, n, m.total_mass)
td = torch.float32 if dtype == "f32" else torch.float64
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(td).cuda()
inp = [dev(B[k]) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu")]
mask = torch.from_numpy(B["mask"]).cuda()
ig = s.dynamics(inp[0], inp[1], want=("p",))["p"]
extra = [dev(B["tau_prev"]), dev(B["f_prev"]), ig, torch.zeros_like(ig)]
print("plan:", s.plan_tick(n))
for _ in range(20):
    out = s.step(*inp, mask, *extra, want_mats=True)
torch.cuda.synchronize()
raw = out["pf"].contiguous().view(torch.int32).cpu().numpy().reshape(-1).astype(np.int64) & 0xFFFFFFFF
nwg = 2 * ((n // (2 if dtype == "f32" else 1) + 15) // 16)
st = raw[: nwg * 8].reshape(nwg, 8)
role, t0, t1, hw, xcc = st[:, 0], st[:, 1], st[:, 2], st[:, 3], st[:, 4] & 0xF
wave, simd, cu, sh, se = hw & 0xF, (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
slot = ((xcc * 8 + se) * 2 + sh) * 16 + cu            # a CU
simd_id = slot * 4 + simd
base = t0.min()
dur = (t1 - t0) * 0.01
print("workgroups %d; kernel-wide first entry -> last exit %.2f us; entry spread %.2f us" % (nwg, (t1.max() - base) * 0.01, (t0.max() - base) * 0.01))
for r, nm in ((0, "sweep role"), (1, "observer role")):
    d = dur[role == r]
    print("  %-14s duration median %.2f us  p10 %.2f  p90 %.2f  max %.2f;  exit (since first entry) median %.2f  max %.2f" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.max(), np.median((t1 - base)[role == r]) * 0.01, (t1 - base)[role == r].max() * 0.01))
print("distinct CUs used: %d, distinct SIMDs: %d" % (len(set(slot.tolist())), len(set(simd_id.tolist()))))
# who shares a SIMD
from collections import defaultdict
by = defaultdict(list)
for i in range(nwg):
    by[int(simd_id[i])].append(i)
kinds = defaultdict(list)
for sidx, wl in by.items():
    key = "".join(sorted("SO"[int(role[i])] for i in wl))
    kinds[key].append(wl)
for key, lst in sorted(kinds.items()):
    d_s = [dur[i] for wl in lst for i in wl if role[i] == 0]
    d_o = [dur[i] for wl in lst for i in wl if role[i] == 1]
    ends = [max((t1[i] - base) * 0.01 for i in wl) for wl in lst]
    print("  SIMDs holding %-4s: %4d   sweep dur median %s  observer dur median %s   last exit median %.2f max %.2f" % (key, len(lst), "%.2f" % np.median(d_s) if d_s else "-", "%.2f" % np.median(d_o) if d_o else "-", np.median(ends), max(ends)))
# dispatch order: which workgroup indices share a SIMD / CU
ex = sorted(by.items())[:6]
print("  examples (SIMD id: workgroup indices):", [(k, v) for k, v in ex])
bycu = defaultdict(list)
for i in range(nwg):
    bycu[int(slot[i])].append(i)
print("  examples (CU: workgroup indices):", [(k, sorted(v)) for k, v in sorted(bycu.items())[:3]])
print("  xcc of workgroups 0..15:", xcc[:16].tolist(), " cu of workgroups 0, 8, 16, ...:", cu[0:8 * 20:8].tolist(), " simd:", simd[0:8 * 20:8].tolist())

"""TEST INFRASTRUCTURE -- ctypes wrapper around oracle/_build/libwbc_oracle.so (the CPU oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
Batch arrays here are row-per-state numpy arrays: q[N,19], v[N,18], ...
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libwbc_oracle.so")


def build(force=False):
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB)
            for f in ("wbc_oracle.hpp", "wbc_oracle_capi.cpp", "op_count.cpp", "qp_general.hpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.wbco_model_create.restype = C.c_void_p
        _lib.wbco_maxv.restype = C.c_int
    return _lib


def _maxv():
    return lib().wbco_maxv()


def make_params_struct(P):
    MAXV = _maxv()

    class Params(C.Structure):
        _fields_ = [("S", C.c_double * 6), ("alpha", C.c_double), ("fn_min", C.c_double), ("fn_max", C.c_double),
                    ("mu_scale", C.c_double), ("dt", C.c_double), ("observer_order", C.c_int), ("max_iter", C.c_int),
                    ("qp_tol", C.c_double), ("K1", C.c_double * MAXV), ("K2", C.c_double * MAXV)]

    s = Params()
    for i in range(6):
        s.S[i] = float(P["S"][i])
    s.alpha, s.fn_min, s.fn_max = float(P["alpha"]), float(P["fn_min"]), float(P["fn_max"])
    s.mu_scale, s.dt = float(P["mu_scale"]), float(P["dt"])
    s.observer_order, s.max_iter, s.qp_tol = int(P["observer_order"]), int(P["max_iter"]), float(P["qp_tol"])
    for i in range(len(P["K1"])):
        s.K1[i] = float(P["K1"][i])
        s.K2[i] = float(P["K2"][i])
    return s


def make_ref_params_struct(G):
    MAXV = _maxv()

    class RefParams(C.Structure):
        _fields_ = [("kp_com", C.c_double * 3), ("kd_com", C.c_double * 3), ("kp_rot", C.c_double * 3),
                    ("kd_rot", C.c_double * 3), ("kp_joint", C.c_double), ("kd_joint", C.c_double),
                    ("inertia_nom", C.c_double * 3), ("q_nom", C.c_double * MAXV)]

    s = RefParams()
    for i in range(3):
        s.kp_com[i], s.kd_com[i] = float(G["kp_com"][i]), float(G["kd_com"][i])
        s.kp_rot[i], s.kd_rot[i] = float(G["kp_rot"][i]), float(G["kd_rot"][i])
        s.inertia_nom[i] = float(G["inertia_nom"][i])
    s.kp_joint, s.kd_joint = float(G["kp_joint"]), float(G["kd_joint"])
    for i, x in enumerate(G["q_nom"]):
        s.q_nom[i] = float(x)
    return s


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, flat):
        self.flat = flat
        self.nb = int(flat["nb"])
        self.nv = 6 + self.nb - 1
        self.nq = self.nv + 1
        self.nj = self.nb - 1
        self.nf = len(flat["foot_body"])
        self.nm = self.nv * (self.nv + 1) // 2
        f = lambda k: np.ascontiguousarray(flat[k], dtype=np.float64)
        self._keep = [np.ascontiguousarray(flat["parent"], dtype=np.int32), f("Rt"), f("rt"), f("axis"), f("mass"),
                      f("com"), f("Ic"), np.ascontiguousarray(flat["foot_body"], dtype=np.int32), f("foot_off"),
                      f("gravity")]
        k = self._keep
        self.h = lib().wbco_model_create(self.nb, _p(k[0]), _p(k[1]), _p(k[2]), _p(k[3]), _p(k[4]), _p(k[5]), _p(k[6]),
                                         self.nf, _p(k[7]), _p(k[8]), _p(k[9]))
        assert self.h, "oracle model_create failed"
        self.h = C.c_void_p(self.h)

    def __del__(self):
        try:
            lib().wbco_model_destroy(self.h)
        except Exception:
            pass

    @staticmethod
    def _suf(dtype):
        return "f64" if np.dtype(dtype) == np.float64 else "f32"

    def dynamics(self, q, v, nthreads=1):
        dt = q.dtype
        N = q.shape[0]
        q = np.ascontiguousarray(q)
        v = np.ascontiguousarray(v, dtype=dt)
        o = dict(M=np.empty((N, self.nm), dt), h=np.empty((N, self.nv), dt), Jc=np.empty((N, 3 * self.nf * self.nv), dt),
                 pf=np.empty((N, 3 * self.nf), dt), p=np.empty((N, self.nv), dt), beta=np.empty((N, self.nv), dt))
        getattr(lib(), "wbco_dynamics_" + self._suf(dt))(self.h, N, _p(q), _p(v), _p(o["M"]), _p(o["h"]), _p(o["Jc"]),
                                                         _p(o["pf"]), _p(o["p"]), _p(o["beta"]), int(nthreads))
        return o

    def rnea(self, q, v, vdot=None, gravity=True):
        dt = q.dtype
        N = q.shape[0]
        out = np.empty((N, self.nv), dt)
        q = np.ascontiguousarray(q)
        v = np.ascontiguousarray(v, dtype=dt)
        vd = None if vdot is None else np.ascontiguousarray(vdot, dtype=dt)
        getattr(lib(), "wbco_rnea_" + self._suf(dt))(self.h, N, _p(q), _p(v), _p(vd), int(gravity), _p(out))
        return out

    def step(self, P, q, v, w_des, vdot_des, normals, mu, mask, tau_prev=None, f_prev=None, integ=None, r=None,
             nthreads=1, aset=None):
        """integ, r are updated IN PLACE when the observer is on.  aset (optional, uint32 [N]): warm start of the QP from that active
        set (encoding: include/wbc_hip.h, wbc_step_batch_warm); the returned dict always carries the active set at the solution."""
        dt = q.dtype
        N = q.shape[0]
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dt)
        q, v, w_des, vdot_des, normals, mu = map(c, (q, v, w_des, vdot_des, normals, mu))
        tau_prev, f_prev = c(tau_prev), c(f_prev)
        mask = np.ascontiguousarray(mask, dtype=np.int32)
        if P["observer_order"] > 0:
            assert integ is not None and r is not None and integ.dtype == dt and r.dtype == dt
            assert integ.flags.c_contiguous and r.flags.c_contiguous
        tau = np.empty((N, self.nj), dt)
        f = np.empty((N, 3 * self.nf), dt)
        status = np.empty(N, np.int32)
        iters = np.empty(N, np.int32)
        ps = make_params_struct(P)
        aset_in = None if aset is None else np.ascontiguousarray(aset, dtype=np.uint32)
        aset_out = np.zeros(N, np.uint32)
        getattr(lib(), "wbco_step_" + self._suf(dt))(self.h, C.byref(ps), N, _p(q), _p(v), _p(w_des), _p(vdot_des),
                                                     _p(normals), _p(mu), _p(mask), _p(tau_prev), _p(f_prev), _p(integ),
                                                     _p(r), _p(tau), _p(f), _p(status), _p(iters), int(nthreads), _p(aset_in), _p(aset_out))
        return dict(tau=tau, f=f, status=status, iters=iters, aset=aset_out)


def _rollout(self, P, horizon, q, v, w_des, vdot_des, normals, mu, mask, tau_ext=None, tau_prev=None, f_prev=None,
             integ=None, r=None, want_traj=False, nthreads=1, warm=False):
    """q, v (and tau_prev, f_prev, integ, r when given) are updated IN PLACE.  Returns dict(status, iters_sum[, tau_traj]).
    warm: ticks after the first start their QP from the previous tick's active set."""
    dt = q.dtype
    N = q.shape[0]
    assert q.flags.c_contiguous and v.flags.c_contiguous and v.dtype == dt
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dt)
    w_des, vdot_des, normals, mu, tau_ext = map(c, (w_des, vdot_des, normals, mu, tau_ext))
    mask = np.ascontiguousarray(mask, dtype=np.int32)
    if tau_prev is None:
        tau_prev = np.zeros((N, self.nj), dt)
    if f_prev is None:
        f_prev = np.zeros((N, 3 * self.nf), dt)
    for a in (tau_prev, f_prev, integ, r):
        assert a is None or (a.dtype == dt and a.flags.c_contiguous)
    traj = np.zeros((N, horizon, self.nj), dt) if want_traj else None
    status = np.zeros(N, np.int32)
    iters_sum = np.zeros(N, np.int32)
    ps = make_params_struct(P)
    getattr(lib(), "wbco_rollout_" + self._suf(dt))(self.h, C.byref(ps), N, int(horizon), _p(q), _p(v), _p(w_des),
                                                    _p(vdot_des), _p(normals), _p(mu), _p(mask), _p(tau_ext), _p(tau_prev),
                                                    _p(f_prev), _p(integ), _p(r), _p(traj), _p(status), int(nthreads), int(bool(warm)), _p(iters_sum))
    out = dict(status=status, tau_prev=tau_prev, f_prev=f_prev, iters_sum=iters_sum)
    if want_traj:
        out["tau_traj"] = traj
    return out


Oracle.rollout = _rollout

def _reference(self, G, q, v, plan, t=0.0):
    """CoM reference generator (a11): plan [N,12] -> dict(w_des [N,6], vdot_des [N,nv], com [N,6])."""
    dt = q.dtype
    N = q.shape[0]
    c = lambda a: np.ascontiguousarray(a, dtype=dt)
    w = np.empty((N, 6), dt)
    vd = np.empty((N, self.nv), dt)
    com = np.empty((N, 6), dt)
    gs = make_ref_params_struct(G)
    ct = C.c_double if dt == np.float64 else C.c_float
    getattr(lib(), "wbco_reference_" + self._suf(dt))(self.h, C.byref(gs), N, _p(c(q)), _p(c(v)), _p(c(plan)), ct(t), _p(w),
                                                      _p(vd), _p(com))
    return dict(w_des=w, vdot_des=vd, com=com)


def _rollout_tracking(self, P, G, horizon, q, v, plan, normals, mu, mask, tau_ext=None, tau_prev=None, f_prev=None,
                      integ=None, r=None, want_traj=False, want_com=False, nthreads=1, warm=False):
    """Planner-in-the-loop rollout; q, v (and tau_prev, f_prev, integ, r when given) are updated IN PLACE."""
    dt = q.dtype
    N = q.shape[0]
    assert q.flags.c_contiguous and v.flags.c_contiguous and v.dtype == dt
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dt)
    plan, normals, mu, tau_ext = map(c, (plan, normals, mu, tau_ext))
    mask = np.ascontiguousarray(mask, dtype=np.int32)
    if tau_prev is None:
        tau_prev = np.zeros((N, self.nj), dt)
    if f_prev is None:
        f_prev = np.zeros((N, 3 * self.nf), dt)
    for a in (tau_prev, f_prev, integ, r):
        assert a is None or (a.dtype == dt and a.flags.c_contiguous)
    traj = np.zeros((N, horizon, self.nj), dt) if want_traj else None
    com = np.zeros((N, horizon, 6), dt) if want_com else None
    status = np.zeros(N, np.int32)
    getattr(lib(), "wbco_rollout_tracking_" + self._suf(dt))(
        self.h, C.byref(make_params_struct(P)), C.byref(make_ref_params_struct(G)), N, int(horizon), _p(q), _p(v), _p(plan),
        _p(normals), _p(mu), _p(mask), _p(tau_ext), _p(tau_prev), _p(f_prev), _p(integ), _p(r), _p(traj), _p(com), _p(status),
        int(nthreads), int(bool(warm)))
    out = dict(status=status, tau_prev=tau_prev, f_prev=f_prev)
    if want_traj:
        out["tau_traj"] = traj
    if want_com:
        out["com_traj"] = com
    return out


Oracle.reference = _reference
Oracle.rollout_tracking = _rollout_tracking


def _qp_time(self, P, q, v, w_des, normals, mu, mask):
    """Per-QP wall time (ns) of {assembly + solve} on one thread, float64; returns (ns[N], iters[N])."""
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    N = q.shape[0]
    ns = np.zeros(N)
    it = np.zeros(N, np.int32)
    ps = make_params_struct(P)
    lib().wbco_qp_time_f64(self.h, C.byref(ps), N, _p(c(q)), _p(c(v)), _p(c(w_des)), _p(c(normals)), _p(c(mu)),
                           _p(np.ascontiguousarray(mask, dtype=np.int32)), _p(ns), _p(it))
    return ns, it


Oracle.qp_time = _qp_time

OP_STAGES = ("dynamics", "observer", "qp_assemble", "qp_solve", "torque_map")
OP_KINDS = ("add", "mul", "div", "sqrt", "trig", "cmp")


def _op_count(self, P, q, v, w_des, vdot_des, normals, mu, mask, tau_prev=None, f_prev=None, integ=None, r=None):
    """Instrumented operation count of ONE oracle step for ONE state (1-D float64 inputs; oracle/op_count.cpp).
    Returns dict(counts={stage: {kind: n}}, flops=adds+muls+divs+sqrts+trig, iters, tau, f)."""
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    k = self._keep
    counts = np.zeros((len(OP_STAGES), len(OP_KINDS)), np.int64)
    tau = np.zeros(self.nj)
    f = np.zeros(3 * self.nf)
    ps = make_params_struct(P)
    if integ is not None:
        integ, r = integ.copy(), r.copy()
    fn = lib().wbco_op_count
    fn.restype = C.c_int
    it = fn(self.nb, _p(k[0]), _p(k[1]), _p(k[2]), _p(k[3]), _p(k[4]), _p(k[5]), _p(k[6]), self.nf, _p(k[7]), _p(k[8]),
            _p(k[9]), C.byref(ps), _p(c(q)), _p(c(v)), _p(c(w_des)), _p(c(vdot_des)), _p(c(normals)), _p(c(mu)), int(mask),
            _p(c(tau_prev)), _p(c(f_prev)), _p(integ), _p(r), _p(counts), _p(tau), _p(f))
    assert it >= 0
    d = {s: {kd: int(counts[i, j]) for j, kd in enumerate(OP_KINDS)} for i, s in enumerate(OP_STAGES)}
    return dict(counts=d, flops=int(counts[:, :5].sum()), flops_by_stage={s: int(counts[i, :5].sum()) for i, s in enumerate(OP_STAGES)},
                iters=it, tau=tau, f=f)


Oracle.op_count = _op_count


def _op_count_rollout(self, P, horizon, q, v, w_des, vdot_des, normals, mu, mask, tau_ext=None, integ=None, warm=True):
    """Instrumented operation count of ONE oracle rollout of ONE state (1-D float64 inputs): dict(flops, flops_per_tick, iters_sum, counts)."""
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    k = self._keep
    counts = np.zeros(len(OP_KINDS), np.int64)
    ps = make_params_struct(P)
    fn = lib().wbco_op_count_rollout
    fn.restype = C.c_int
    it = fn(self.nb, _p(k[0]), _p(k[1]), _p(k[2]), _p(k[3]), _p(k[4]), _p(k[5]), _p(k[6]), self.nf, _p(k[7]), _p(k[8]), _p(k[9]), C.byref(ps),
            int(horizon), int(bool(warm)), _p(c(q)), _p(c(v)), _p(c(w_des)), _p(c(vdot_des)), _p(c(normals)), _p(c(mu)), int(mask), _p(c(tau_ext)),
            _p(c(integ)), _p(counts))
    assert it >= 0
    fl = int(counts[:5].sum())
    return dict(counts={kd: int(counts[j]) for j, kd in enumerate(OP_KINDS)}, flops=fl, flops_per_tick=fl / horizon, iters_sum=it)


Oracle.op_count_rollout = _op_count_rollout


def _step_timed(self, P, q, v, w_des, vdot_des, normals, mu, mask, tau_prev=None, f_prev=None, reps=1, nthreads=1):
    """`reps` passes of step() over the batch inside ONE OpenMP region (observer state zero, per-thread scratch): returns the wall
    seconds between the region's two barriers -- thread start-up and per-call fork/join are outside."""
    dt = q.dtype
    N = q.shape[0]
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dt)
    q, v, w_des, vdot_des, normals, mu, tau_prev, f_prev = map(c, (q, v, w_des, vdot_des, normals, mu, tau_prev, f_prev))
    mask = np.ascontiguousarray(mask, dtype=np.int32)
    tau, f, status = np.empty((N, self.nj), dt), np.empty((N, 3 * self.nf), dt), np.empty(N, np.int32)
    sec = C.c_double(0.0)
    getattr(lib(), "wbco_step_timed_" + self._suf(dt))(self.h, C.byref(make_params_struct(P)), N, _p(q), _p(v), _p(w_des), _p(vdot_des),
                                                       _p(normals), _p(mu), _p(mask), _p(tau_prev), _p(f_prev), _p(tau), _p(f), _p(status),
                                                       int(reps), int(nthreads), C.byref(sec))
    return sec.value


Oracle.step_timed = _step_timed


def qp_solve(H, g, Cm, d, max_iter=100, tol=1e-9, warm=None, want_active=False):
    """warm: boolean [m] guess of the active set (see qp_solve_gi); want_active: also return the final active set."""
    dt = H.dtype
    n, m = len(g), len(d)
    H, g, Cm, d = (np.ascontiguousarray(a, dtype=dt) for a in (H, g, Cm, d))
    x = np.zeros(n, dt)
    lam = np.zeros(max(m, 1), dt)
    st = C.c_int(0)
    fn = getattr(lib(), "wbco_qp_solve_" + Oracle._suf(dt))
    fn.restype = C.c_int
    ct = C.c_double if dt == np.float64 else C.c_float
    wf = None if warm is None else np.ascontiguousarray(warm, dtype=np.uint8)
    af = np.zeros(max(m, 1), np.uint8)
    it = fn(n, m, _p(H), _p(g), _p(Cm), _p(d), int(max_iter), ct(tol), _p(x), _p(lam), C.byref(st), _p(wf), _p(af))
    if want_active:
        return x, lam[:m], st.value, it, af[:m].astype(bool)
    return x, lam[:m], st.value, it


def qp_general(H, g, Cm, d, meq=0, max_iter=200, tol=1e-9):
    """qp_general.hpp: min 1/2 x'Hx + g'x  s.t.  Cm[:meq] x = d[:meq], Cm[meq:] x >= d[meq:]   (n <= 36, m <= 64).
    One problem (H[n,n]) or a batch (H[N,n,n], ...; fp64 only).  Returns x, lambda, status, iters."""
    H = np.asarray(H)
    if H.ndim == 3:
        N, n = H.shape[0], H.shape[1]
        m = np.asarray(d).shape[1]
        c = lambda a: np.ascontiguousarray(a, np.float64)
        H, g, Cm, d = c(H), c(g), c(Cm).reshape(N, max(m, 0) * n), c(d)
        x, lam = np.zeros((N, n)), np.zeros((N, max(m, 1)))
        st, it = np.zeros(N, np.int32), np.zeros(N, np.int32)
        lib().wbco_qp_general_batch_f64(N, n, m, int(meq), _p(H), _p(g), _p(Cm), _p(d), int(max_iter), C.c_double(tol), _p(x), _p(lam) if m else _p(lam),
                                        _p(st), _p(it), 8)
        return x, lam[:, :m], st, it
    dt = H.dtype if H.dtype in (np.float32, np.float64) else np.float64
    n, m = len(g), len(d)
    H, g, Cm, d = (np.ascontiguousarray(a, dtype=dt) for a in (H, g, Cm, d))
    x = np.zeros(n, dt)
    lam = np.zeros(max(m, 1), dt)
    st = C.c_int(0)
    fn = getattr(lib(), "wbco_qp_general_" + Oracle._suf(dt))
    fn.restype = C.c_int
    ct = C.c_double if dt == np.float64 else C.c_float
    it = fn(n, m, int(meq), _p(H), _p(g), _p(Cm), _p(d), int(max_iter), ct(tol), _p(x), _p(lam), C.byref(st))
    return x, lam[:m], st.value, it


def qp_assemble(P, nf, mask, pb, pf, normals, mu, b):
    dt = np.float64
    ns = bin(mask).count("1")
    n, m = 3 * ns, 6 * ns
    H = np.zeros((max(n, 1), max(n, 1)), dt)
    g = np.zeros(max(n, 1), dt)
    Cm = np.zeros((max(m, 1), max(n, 1)), dt)
    d = np.zeros(max(m, 1), dt)
    mo = C.c_int(0)
    ps = make_params_struct(P)
    c = lambda a: np.ascontiguousarray(a, dtype=dt)
    fn = lib().wbco_qp_assemble_f64
    fn.restype = C.c_int
    nn = fn(C.byref(ps), nf, int(mask), _p(c(pb)), _p(c(pf)), _p(c(normals)), _p(c(mu)), _p(c(b)), _p(H), _p(g),
            _p(Cm), _p(d), C.byref(mo))
    assert nn == n and mo.value == m
    return H[:n, :n].copy() if n else np.zeros((0, 0)), g[:n], Cm[:m, :n].copy() if n else np.zeros((0, 0)), d[:m]

"""MI355X-native batched whole-body-control hot path (host-side Python binding).

Thin ctypes layer over lib/libwbc_hip.so (the C-ABI in include/wbc_hip.h).  PyTorch is used only
for device memory and streams.  There is NO CPU fallback: if the HIP library is missing or no
gfx950 device is present, construction fails loudly.

Mirrors the reference controller's shape (/root/reference/README.md:60: the controller is
started with the URDF path; README.md:11: observer + GRF optimisation per tick):
    model  = Model.from_urdf(path)
    solver = Solver(model, params, dtype="f64", device=0, max_batch=N)
    out    = solver.step(q=..., v=..., w_des=..., ...)      # one control tick for N states
Batch tensors are component-major: shape [ncomp, N] (see include/wbc_hip.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WBC_LIB") or os.path.join(_HERE, "lib", "libwbc_hip.so")
SYNTHETIC_URDF = os.path.join(_HERE, "assets", "synthetic_quadruped.urdf")
WBC_MAXV = 32
F64, F32 = 0, 1
ABI_VERSION = 9   # include/wbc_hip.h: wbc_abi_version()

_lib = None


class WbcError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        detail = lib().wbc_last_error().decode() if _lib is not None else ""
        super().__init__("%s failed: status %d (%s) %s" % (where, code, lib().wbc_strerror(code).decode(), detail))


def build_library(force=False):
    """Compile the HIP library in-tree (hipcc --offload-arch=gfx950)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-s"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


class Params(C.Structure):
    _fields_ = [("S", C.c_double * 6), ("alpha", C.c_double), ("fn_min", C.c_double), ("fn_max", C.c_double),
                ("mu_scale", C.c_double), ("dt", C.c_double), ("observer_order", C.c_int), ("max_iter", C.c_int),
                ("qp_tol", C.c_double), ("K1", C.c_double * WBC_MAXV), ("K2", C.c_double * WBC_MAXV)]

    @staticmethod
    def default(dtype="f64"):
        p = Params()
        lib().wbc_params_default(C.byref(p), F64 if dtype == "f64" else F32)
        return p

    @staticmethod
    def from_dict(d, dtype="f64"):
        p = Params.default(dtype)
        for i in range(6):
            p.S[i] = float(d["S"][i])
        for k in ("alpha", "fn_min", "fn_max", "mu_scale", "dt", "qp_tol"):
            setattr(p, k, float(d[k]))
        p.observer_order, p.max_iter = int(d["observer_order"]), int(d["max_iter"])
        for i in range(min(len(d["K1"]), WBC_MAXV)):
            p.K1[i] = float(d["K1"][i])
            p.K2[i] = float(d["K2"][i])
        return p


class RefParams(C.Structure):
    """wbc_ref_params: gains of the CoM reference generator (include/wbc_hip.h)."""
    _fields_ = [("kp_com", C.c_double * 3), ("kd_com", C.c_double * 3), ("kp_rot", C.c_double * 3), ("kd_rot", C.c_double * 3),
                ("kp_joint", C.c_double), ("kd_joint", C.c_double), ("inertia_nom", C.c_double * 3),
                ("q_nom", C.c_double * WBC_MAXV)]

    @staticmethod
    def default():
        g = RefParams()
        lib().wbc_ref_params_default(C.byref(g))
        return g

    @staticmethod
    def from_dict(d):
        g = RefParams.default()
        for k in ("kp_com", "kd_com", "kp_rot", "kd_rot", "inertia_nom"):
            for i in range(3):
                getattr(g, k)[i] = float(d[k][i])
        g.kp_joint, g.kd_joint = float(d["kp_joint"]), float(d["kd_joint"])
        for i, x in enumerate(d["q_nom"]):
            g.q_nom[i] = float(x)
        return g


PLAN_WORDS = 12


class SolverOptions(C.Structure):
    """wbc_solver_options (include/wbc_hip.h): the kernel-selection switches of a solver.  The library itself never reads
    the environment; Solver(options=None) builds them from the defaults overridden by the WBC_* variables below -- a
    convenience of THIS binding for the A/B scripts under tools/ and bench.py."""
    _fields_ = [("struct_size", C.c_size_t), ("fused_max", C.c_longlong), ("rollout_persistent", C.c_int),
                ("rollout_spw", C.c_int), ("obs_split_min", C.c_longlong), ("one_zerocopy", C.c_int), ("timing_mode", C.c_int), ("qp_tile", C.c_int), ("obs_split_serial", C.c_int), ("qp_lane", C.c_int), ("f32_pack2", C.c_int), ("keep_structural", C.c_int), ("rollout_warm", C.c_int),
                ("multi_threads", C.c_int), ("multi_spin_us", C.c_int), ("obs_colaunch", C.c_int), ("tile_tick", C.c_int), ("fused_pair", C.c_int)]
    ENV = {"WBC_FUSED_MAX": ("fused_max", int), "WBC_ROLLOUT_PERSISTENT": ("rollout_persistent", int),
           "WBC_ROLLOUT_SPW": ("rollout_spw", int), "WBC_OBS_SPLIT_MIN": ("obs_split_min", int),
           "WBC_ONE_ZEROCOPY": ("one_zerocopy", int), "WBC_QP_TILE": ("qp_tile", int), "WBC_OBS_SPLIT_SERIAL": ("obs_split_serial", int), "WBC_QP_LANE": ("qp_lane", int), "WBC_F32_PACK2": ("f32_pack2", int), "WBC_KEEP_STRUCTURAL": ("keep_structural", int), "WBC_ROLLOUT_WARM": ("rollout_warm", int),
           "WBC_MULTI_THREADS": ("multi_threads", int), "WBC_MULTI_SPIN_US": ("multi_spin_us", int), "WBC_OBS_COLAUNCH": ("obs_colaunch", int), "WBC_TILE_TICK": ("tile_tick", int), "WBC_FUSED_PAIR": ("fused_pair", int),
           "WBC_TIMING": ("timing_mode", lambda v: 1 if v == "pair" else 0)}

    @staticmethod
    def default():
        o = SolverOptions()
        lib().wbc_solver_options_default(C.byref(o))
        return o

    @staticmethod
    def make(options=None):
        """options: None (defaults + WBC_* environment), a dict of field overrides, or a SolverOptions."""
        if isinstance(options, SolverOptions):
            return options
        o = SolverOptions.default()
        if options is None:
            for var, (field, conv) in SolverOptions.ENV.items():
                if var in os.environ:
                    setattr(o, field, conv(os.environ[var]))
        else:
            for k, v in options.items():
                if k not in dict(SolverOptions._fields_) or k == "struct_size":
                    raise KeyError("unknown solver option %r" % k)
                setattr(o, k, int(v))
        return o


class _BatchIn(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "mask", "tau_prev", "f_prev")]


class _BatchOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("tau", "f", "status", "iters", "M", "h", "Jc", "pf")]


class TickPlan(C.Structure):
    """wbc_tick_plan (include/wbc_hip.h): which kernels a tick of N states runs."""
    _fields_ = [("struct_size", C.c_size_t), ("fused", C.c_int), ("front", C.c_int), ("qp", C.c_int), ("qp_tile", C.c_int),
                ("qp_body", C.c_int), ("sweep_pack2", C.c_int), ("sweep_block", C.c_int), ("qp_warm", C.c_int)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "struct_size"}


def plan_tick(N, dtype="f64", observer_order=0, options=None, want_mats=True, want_pf=True, warm=False):
    """wbc_plan_tick: the kernels a tick of N states runs with these options (dict of wbc_tick_plan's fields); needs no device."""
    pl = TickPlan()
    pl.struct_size = C.sizeof(TickPlan)
    o = SolverOptions.make({} if options is None else options)
    _check(lib().wbc_plan_tick(F64 if dtype == "f64" else F32, int(observer_order), C.byref(o), int(N), int(want_mats), int(want_pf), int(warm), C.byref(pl)),
           "wbc_plan_tick")
    return pl.as_dict()


def dispatch_thresholds(dtype="f64", observer_order=0, options=None, want_mats=True, warm=False):
    """wbc_dispatch_thresholds: the batch sizes at which the tick's kernels change (warm: those of a warm-started tick), ascending; needs no device."""
    out = (C.c_size_t * 16)()
    n = C.c_int(0)
    o = SolverOptions.make({} if options is None else options)
    _check(lib().wbc_dispatch_thresholds(F64 if dtype == "f64" else F32, int(observer_order), C.byref(o), (1 if want_mats else 0) | (2 if warm else 0), out, 16, C.byref(n)),
           "wbc_dispatch_thresholds")
    return [int(out[i]) for i in range(n.value)]


class _ObsState(C.Structure):
    _fields_ = [("integ", C.c_void_p), ("r", C.c_void_p)]


def lib():
    """Load libwbc_hip.so; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        # torch first: its bundled libamdhip64.so.7 must be THE HIP runtime of the process, so that device
        # pointers and streams handed over from torch are valid inside the library (same SONAME => shared).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the WBC hot path)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.wbc_strerror.restype = C.c_char_p
        L.wbc_last_error.restype = C.c_char_p
        L.wbc_model_joint_name.restype = C.c_char_p
        L.wbc_model_foot_link.restype = C.c_char_p
        L.wbc_model_total_mass.restype = C.c_double
        L.wbc_model_total_mass.argtypes = [C.c_void_p]
        L.wbc_model_joint_name.argtypes = [C.c_void_p, C.c_int]
        L.wbc_model_foot_link.argtypes = [C.c_void_p, C.c_int]
        L.wbc_model_free.argtypes = [C.c_void_p]
        L.wbc_solver_destroy.argtypes = [C.c_void_p]
        L.wbc_solver_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
        L.wbc_solver_create_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
        L.wbc_solver_options_default.argtypes = [C.c_void_p]
        L.wbc_solver_options_default.restype = None
        L.wbc_observer_init.argtypes = [C.c_void_p] * 5
        L.wbc_one_map.argtypes = [C.c_void_p, C.c_void_p]
        L.wbc_one_tick.argtypes = [C.c_void_p]
        L.wbc_shard_range.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.wbc_multi_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.wbc_multi_destroy.argtypes = [C.c_void_p]
        L.wbc_multi_destroy.restype = None
        L.wbc_multi_size.argtypes = [C.c_void_p]
        L.wbc_multi_device.argtypes = [C.c_void_p, C.c_int]
        L.wbc_multi_solver.argtypes = [C.c_void_p, C.c_int]
        L.wbc_multi_solver.restype = C.c_void_p
        L.wbc_multi_stream.argtypes = [C.c_void_p, C.c_int]
        L.wbc_multi_stream.restype = C.c_void_p
        L.wbc_multi_rccl_ranks.argtypes = [C.c_void_p]
        L.wbc_multi_set_params.argtypes = [C.c_void_p, C.c_void_p]
        L.wbc_multi_step_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wbc_multi_step_batch_warm.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wbc_multi_rollout_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wbc_multi_allgather_tau.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.wbc_multi_synchronize.argtypes = [C.c_void_p]
        L.wbc_multi_allgather_tau_async.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.wbc_multi_gather_wait.argtypes = [C.c_void_p, C.c_int]
        L.wbc_multi_step_host.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.wbc_dynamics_batch.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 9
        L.wbc_step_batch.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 4
        L.wbc_integrate_batch.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 9
        L.wbc_rollout_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 6
        L.wbc_ref_params_default.argtypes = [C.c_void_p]
        L.wbc_ref_params_default.restype = None
        L.wbc_solver_set_ref_params.argtypes = [C.c_void_p, C.c_void_p]
        L.wbc_reference_batch.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 3 + [C.c_double] + [C.c_void_p] * 4
        L.wbc_compute_reference.argtypes = [C.c_void_p] * 4 + [C.c_double] + [C.c_void_p] * 3
        L.wbc_rollout_tracking_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 8
        L.wbc_plan_tick.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.wbc_solver_plan_tick.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.wbc_step_batch_warm.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 6
        L.wbc_dispatch_thresholds.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.wbc_solver_invalidate_structural.argtypes = [C.c_void_p]
        L.wbc_qp_dense_batch.argtypes = [C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_double] + [C.c_void_p] * 5
        # this binding's struct layouts and array sizes are those of ONE header version: refuse any other library
        if L.wbc_abi_version() != ABI_VERSION:
            raise RuntimeError("%s is ABI %d, this binding is written against ABI %d: rebuild the library (python -c 'import __graft_entry__ as g; g.build()')"
                               % (LIB_PATH, L.wbc_abi_version(), ABI_VERSION))
        L.wbc_solver_collect_timing_n.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.wbc_multi_tick_gather.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.wbc_multi_issue_threads.argtypes = [C.c_void_p]
        L.wbc_multi_set_peer_copies.argtypes = [C.c_void_p, C.c_int]
        L.wbc_multi_gather_pushes.argtypes = [C.c_void_p]
        L.wbc_multi_host_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.wbc_multi_probe_issue.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _check(code, where):
    if code != 0:
        raise WbcError(code, where)


class Model:
    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        nb, nq, nv, nj, nf = (C.c_int() for _ in range(5))
        _check(lib().wbc_model_dims(self._h, C.byref(nb), C.byref(nq), C.byref(nv), C.byref(nj), C.byref(nf)), "wbc_model_dims")
        self.nb, self.nq, self.nv, self.nj, self.nf = nb.value, nq.value, nv.value, nj.value, nf.value

    @staticmethod
    def from_urdf(path, foot_links=None):
        h = C.c_void_p()
        if foot_links:
            arr = (C.c_char_p * len(foot_links))(*[s.encode() for s in foot_links])
            rc = lib().wbc_model_load_urdf(path.encode(), arr, len(foot_links), C.byref(h))
        else:
            rc = lib().wbc_model_load_urdf(path.encode(), None, 0, C.byref(h))
        _check(rc, "wbc_model_load_urdf")
        return Model(h.value)

    @staticmethod
    def from_flat(flat):
        h = C.c_void_p()
        c = lambda k, t: np.ascontiguousarray(flat[k], dtype=t)
        a = [c("parent", np.int32), c("Rt", np.float64), c("rt", np.float64), c("axis", np.float64),
             c("mass", np.float64), c("com", np.float64), c("Ic", np.float64), c("foot_body", np.int32),
             c("foot_off", np.float64), c("gravity", np.float64)]
        p = [x.ctypes.data_as(C.c_void_p) for x in a]
        rc = lib().wbc_model_from_flat(int(flat["nb"]), p[0], p[1], p[2], p[3], p[4], p[5], p[6], len(a[7]), p[7], p[8],
                                       p[9], C.byref(h))
        _check(rc, "wbc_model_from_flat")
        return Model(h.value)

    def flat(self):
        nb, nf = self.nb, self.nf
        o = dict(nb=nb, parent=np.zeros(nb, np.int32), Rt=np.zeros((nb, 9)), rt=np.zeros((nb, 3)), axis=np.zeros((nb, 3)),
                 mass=np.zeros(nb), com=np.zeros((nb, 3)), Ic=np.zeros((nb, 6)), foot_body=np.zeros(nf, np.int32),
                 foot_off=np.zeros((nf, 3)), gravity=np.zeros(3))
        p = lambda k: o[k].ctypes.data_as(C.c_void_p)
        _check(lib().wbc_model_get_flat(self._h, p("parent"), p("Rt"), p("rt"), p("axis"), p("mass"), p("com"), p("Ic"),
                                        p("foot_body"), p("foot_off"), p("gravity")), "wbc_model_get_flat")
        o["joint_names"] = [(lib().wbc_model_joint_name(self._h, j) or b"").decode() for j in range(self.nj)]
        o["foot_links"] = [(lib().wbc_model_foot_link(self._h, k) or b"").decode() for k in range(self.nf)]
        return o

    @property
    def total_mass(self):
        return lib().wbc_model_total_mass(self._h)

    def __del__(self):
        try:
            lib().wbc_model_free(self._h)
        except Exception:
            pass


class Solver:
    """Device-side context: one per GPU (and per stream)."""

    def __init__(self, model, params=None, dtype="f64", device=0, max_batch=4096, options=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible to torch: the WBC hot path has no CPU fallback")
        self.torch = torch
        self.model = model
        self.dtype = dtype
        self.tdtype = torch.float64 if dtype == "f64" else torch.float32
        self.device = torch.device("cuda", device)
        self.max_batch = int(max_batch)
        self.params = params if params is not None else Params.default(dtype)
        h = C.c_void_p()
        self.options = SolverOptions.make(options)
        _check(lib().wbc_solver_create_ex(model._h, C.byref(self.params), F64 if dtype == "f64" else F32, device,
                                          self.max_batch, C.byref(self.options), C.byref(h)), "wbc_solver_create_ex")
        self._h = h

    def set_params(self, params):
        _check(lib().wbc_solver_set_params(self._h, C.byref(params)), "wbc_solver_set_params")
        self.params = params

    def __del__(self):
        try:
            lib().wbc_solver_destroy(self._h)
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _ptr(self, t, rows, N, dtype=None):
        if t is None:
            return None
        assert t.is_cuda and t.is_contiguous(), "batch tensors must be contiguous CUDA tensors"
        assert t.dtype == (dtype or self.tdtype), (t.dtype, dtype or self.tdtype)
        assert t.numel() == rows * N, (tuple(t.shape), rows, N)
        return C.c_void_p(t.data_ptr())

    def empty(self, rows, N, dtype=None):
        return self.torch.empty((rows, N), dtype=dtype or self.tdtype, device=self.device)

    def dynamics(self, q, v, want=("M", "h", "Jc", "pf"), out=None):
        """Dynamics sweep: returns dict of [ncomp, N] tensors for the names in `want`."""
        m = self.model
        N = q.shape[1]
        rows = dict(M=m.nv * (m.nv + 1) // 2, h=m.nv, Jc=3 * m.nf * m.nv, pf=3 * m.nf, p=m.nv, beta=m.nv)
        out = dict(out or {})
        for k in want:
            if k not in out:
                out[k] = self.empty(rows[k], N)
        g = lambda k: self._ptr(out.get(k), rows[k], N)
        _check(lib().wbc_dynamics_batch(self._h, N, self._ptr(q, m.nq, N), self._ptr(v, m.nv, N), g("M"), g("h"), g("Jc"),
                                        g("pf"), g("p"), g("beta"), self._stream()), "wbc_dynamics_batch")
        return out

    def step(self, q, v, w_des, vdot_des, normals, mu, mask, tau_prev=None, f_prev=None, obs_integ=None, obs_r=None,
             out=None, want_mats=False, _prepared=False, active_in=None, warm=False):
        """One control tick.  Observer state tensors are updated in place.  Returns dict(tau, f, status, iters[, M, h, Jc, pf]).
        warm=True (or active_in given): wbc_step_batch_warm -- the QPs start from the int32 [N] tensor active_in (None: cold) and the
        final active sets are returned as out["active"] (pass out={"active": active_in} to update a set in place)."""
        torch = self.torch
        m = self.model
        N = q.shape[1]
        out = dict(out or {})
        rows = dict(tau=m.nj, f=3 * m.nf, M=m.nv * (m.nv + 1) // 2, h=m.nv, Jc=3 * m.nf * m.nv, pf=3 * m.nf)
        for k in ("tau", "f"):
            if k not in out:
                out[k] = self.empty(rows[k], N)
        for k in ("status", "iters"):
            if k not in out:
                out[k] = torch.empty(N, dtype=torch.int32, device=self.device)
        if want_mats:
            if "M" not in out or "Jc" not in out:
                # keep_structural identifies "the same buffers" by address, and the caching allocator hands a just-freed address out
                # again: buffers allocated HERE are always written in full (only a caller that passes out= keeps the constants)
                lib().wbc_solver_invalidate_structural(self._h)
            for k in ("M", "h", "Jc", "pf"):
                if k not in out:
                    out[k] = self.empty(rows[k], N)
        bi = _BatchIn(self._ptr(q, m.nq, N), self._ptr(v, m.nv, N), self._ptr(w_des, 6, N), self._ptr(vdot_des, m.nv, N),
                      self._ptr(normals, 3 * m.nf, N), self._ptr(mu, m.nf, N), self._ptr(mask, 1, N, torch.int32),
                      self._ptr(tau_prev, m.nj, N), self._ptr(f_prev, 3 * m.nf, N))
        g = lambda k: self._ptr(out.get(k), rows[k], N)
        bo = _BatchOut(g("tau"), g("f"), self._ptr(out["status"], 1, N, torch.int32),
                       self._ptr(out["iters"], 1, N, torch.int32), g("M"), g("h"), g("Jc"), g("pf"))
        ob = _ObsState(self._ptr(obs_integ, m.nv, N), self._ptr(obs_r, m.nv, N))
        warm = warm or active_in is not None
        if warm and "active" not in out:
            out["active"] = torch.zeros(N, dtype=torch.int32, device=self.device)
        if _prepared:   # prepare_step(): hand back the validated argument structs instead of launching
            return out, (N, bi, bo, ob, (q, v, w_des, vdot_des, normals, mu, mask, tau_prev, f_prev, obs_integ, obs_r, active_in), warm)
        if warm:
            _check(lib().wbc_step_batch_warm(self._h, N, C.byref(bi), C.byref(bo), C.byref(ob), self._ptr(active_in, 1, N, torch.int32),
                                             self._ptr(out["active"], 1, N, torch.int32), self._stream()), "wbc_step_batch_warm")
        else:
            _check(lib().wbc_step_batch(self._h, N, C.byref(bi), C.byref(bo), C.byref(ob), self._stream()), "wbc_step_batch")
        return out

    def prepare_step(self, *args, **kw):
        """Same arguments as step(); validates them and builds the C argument structs ONCE.  Returns (tick, out):
        tick() launches one control tick on the current stream with nothing but the C call in it (a control loop or a
        benchmark that reuses its buffers: ~3 us of host time per tick instead of ~15), out is step()'s dict."""
        out, (N, bi, bo, ob, keep, warm) = self.step(*args, _prepared=True, **kw)
        h, rbi, rbo, rob, stream_of = self._h, C.byref(bi), C.byref(bo), C.byref(ob), self._stream
        if warm:   # (a prepared warm tick carries its sets from call to call in out["active"]; the first call starts from active_in, or cold)
            import torch
            fnw = lib().wbc_step_batch_warm
            a_in = keep[-1]
            p_out = self._ptr(out["active"], 1, N, torch.int32)
            state = {"first": self._ptr(a_in, 1, N, torch.int32) if a_in is not None else None, "n": 0}

            def tick(_keep=(keep, bi, bo, ob, out)):
                rc = fnw(h, N, rbi, rbo, rob, state["first"] if state["n"] == 0 else p_out, p_out, stream_of())
                state["n"] += 1
                if rc:
                    _check(rc, "wbc_step_batch_warm")
            return tick, out
        fn = lib().wbc_step_batch

        def tick(_keep=(keep, bi, bo, ob, out)):   # the tensors and structs stay alive as long as the closure does
            rc = fn(h, N, rbi, rbo, rob, stream_of())
            if rc:
                _check(rc, "wbc_step_batch")
        return tick, out

    def plan_tick(self, N, want_mats=True, want_pf=True, warm=False):
        """wbc_solver_plan_tick: which kernels a tick of N states runs on THIS solver (dict of wbc_tick_plan's fields)."""
        pl = TickPlan()
        pl.struct_size = C.sizeof(TickPlan)
        _check(lib().wbc_solver_plan_tick(self._h, int(N), int(want_mats), int(want_pf), int(warm), C.byref(pl)), "wbc_solver_plan_tick")
        return pl.as_dict()

    def invalidate_structural(self):
        """keep_structural: the next tick writes M / Jc in full again (their buffers were freed, reallocated or overwritten)."""
        _check(lib().wbc_solver_invalidate_structural(self._h), "wbc_solver_invalidate_structural")

    def integrate(self, q, v, M, h, Jc, tau, f, tau_ext=None):
        """Forward dynamics with the planned GRFs + semi-implicit Euler; q, v advance IN PLACE (one dt)."""
        m = self.model
        N = q.shape[1]
        _check(lib().wbc_integrate_batch(self._h, N, self._ptr(q, m.nq, N), self._ptr(v, m.nv, N),
                                         self._ptr(M, m.nv * (m.nv + 1) // 2, N), self._ptr(h, m.nv, N),
                                         self._ptr(Jc, 3 * m.nf * m.nv, N), self._ptr(tau, m.nj, N), self._ptr(f, 3 * m.nf, N), self._ptr(tau_ext, m.nv, N),
                                         self._stream()), "wbc_integrate_batch")

    def rollout(self, horizon, q, v, w_des, vdot_des, normals, mu, mask, out, obs_integ=None, obs_r=None, tau_ext=None,
                tau_traj=None):
        """`horizon` dependent ticks; q, v advance IN PLACE; `out` must hold tau, f (previous outputs or zeros), status,
        iters, M, h, Jc (e.g. the dict a previous step(..., want_mats=True) returned)."""
        torch = self.torch
        m = self.model
        N = q.shape[1]
        rows = dict(tau=m.nj, f=3 * m.nf, M=m.nv * (m.nv + 1) // 2, h=m.nv, Jc=3 * m.nf * m.nv, pf=3 * m.nf)
        bi = _BatchIn(self._ptr(q, m.nq, N), self._ptr(v, m.nv, N), self._ptr(w_des, 6, N), self._ptr(vdot_des, m.nv, N),
                      self._ptr(normals, 3 * m.nf, N), self._ptr(mu, m.nf, N), self._ptr(mask, 1, N, torch.int32), None, None)
        g = lambda k: self._ptr(out.get(k), rows[k], N)
        bo = _BatchOut(g("tau"), g("f"), self._ptr(out["status"], 1, N, torch.int32),
                       self._ptr(out.get("iters"), 1, N, torch.int32), g("M"), g("h"), g("Jc"), g("pf"))
        ob = _ObsState(self._ptr(obs_integ, m.nv, N), self._ptr(obs_r, m.nv, N))
        tt = None
        if tau_traj is not None:
            assert tau_traj.is_cuda and tau_traj.is_contiguous() and tau_traj.numel() == horizon * m.nj * N
            tt = C.c_void_p(tau_traj.data_ptr())
        _check(lib().wbc_rollout_batch(self._h, N, int(horizon), C.byref(bi), C.byref(bo), C.byref(ob),
                                       self._ptr(tau_ext, m.nv, N), tt, self._stream()), "wbc_rollout_batch")
        return out

    def set_ref_params(self, g):
        """g: RefParams or dict (see RefParams.from_dict)."""
        if isinstance(g, dict):
            g = RefParams.from_dict(g)
        _check(lib().wbc_solver_set_ref_params(self._h, C.byref(g)), "wbc_solver_set_ref_params")

    def reference(self, q, v, plan, t=0.0, out=None, want_com=False):
        """CoM reference generator: plan [12, N] -> dict(w_des [6, N], vdot_des [nv, N][, com [6, N]])."""
        torch = self.torch
        m = self.model
        N = q.shape[1]
        out = {} if out is None else out
        for k, rows in (("w_des", 6), ("vdot_des", m.nv)) + ((("com", 6),) if want_com else ()):
            if k not in out:
                out[k] = torch.empty((rows, N), dtype=self.tdtype, device=q.device)
        _check(lib().wbc_reference_batch(self._h, N, self._ptr(q, m.nq, N), self._ptr(v, m.nv, N),
                                         self._ptr(plan, PLAN_WORDS, N), C.c_double(t), self._ptr(out["w_des"], 6, N),
                                         self._ptr(out["vdot_des"], m.nv, N),
                                         self._ptr(out["com"], 6, N) if want_com else None, self._stream()),
               "wbc_reference_batch")
        return out

    def rollout_tracking(self, horizon, q, v, plan, normals, mu, mask, out, w_des, vdot_des, obs_integ=None, obs_r=None,
                         tau_ext=None, tau_traj=None, com_traj=None):
        """rollout() with the planner in the loop: w_des / vdot_des are scratch buffers regenerated every tick."""
        torch = self.torch
        m = self.model
        N = q.shape[1]
        rows = dict(tau=m.nj, f=3 * m.nf, M=m.nv * (m.nv + 1) // 2, h=m.nv, Jc=3 * m.nf * m.nv, pf=3 * m.nf)
        bi = _BatchIn(self._ptr(q, m.nq, N), self._ptr(v, m.nv, N), self._ptr(w_des, 6, N), self._ptr(vdot_des, m.nv, N),
                      self._ptr(normals, 3 * m.nf, N), self._ptr(mu, m.nf, N), self._ptr(mask, 1, N, torch.int32), None, None)
        g = lambda k: self._ptr(out.get(k), rows[k], N)
        bo = _BatchOut(g("tau"), g("f"), self._ptr(out["status"], 1, N, torch.int32),
                       self._ptr(out.get("iters"), 1, N, torch.int32), g("M"), g("h"), g("Jc"), g("pf"))
        ob = _ObsState(self._ptr(obs_integ, m.nv, N), self._ptr(obs_r, m.nv, N))
        tt = ct = None
        if tau_traj is not None:
            assert tau_traj.is_cuda and tau_traj.is_contiguous() and tau_traj.numel() == horizon * m.nj * N
            tt = C.c_void_p(tau_traj.data_ptr())
        if com_traj is not None:
            assert com_traj.is_cuda and com_traj.is_contiguous() and com_traj.numel() == horizon * 6 * N
            ct = C.c_void_p(com_traj.data_ptr())
        _check(lib().wbc_rollout_tracking_batch(self._h, N, int(horizon), C.byref(bi), C.byref(bo), C.byref(ob),
                                                self._ptr(tau_ext, m.nv, N), self._ptr(plan, PLAN_WORDS, N), tt, ct,
                                                self._stream()), "wbc_rollout_tracking_batch")
        return out

    def compute_torques(self, q, v, w_des, vdot_des, normals, mu, mask, tau_prev=None, f_prev=None, obs_integ=None,
                        obs_r=None):
        """Single-robot host-array call (numpy float64 in/out): the reference's one-robot tick shape."""
        d = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        q, v, w_des, vdot_des, normals, mu, tau_prev, f_prev = map(d, (q, v, w_des, vdot_des, normals, mu, tau_prev, f_prev))
        if obs_integ is not None:
            assert obs_integ.dtype == np.float64 and obs_integ.flags.c_contiguous
            assert obs_r.dtype == np.float64 and obs_r.flags.c_contiguous
        tau = np.zeros(self.model.nj)
        f = np.zeros(3 * self.model.nf)
        st = C.c_int(0)
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        _check(lib().wbc_compute_torques(self._h, p(q), p(v), p(w_des), p(vdot_des), p(normals), p(mu), int(mask),
                                         p(tau_prev), p(f_prev), p(obs_integ), p(obs_r), p(tau), p(f), C.byref(st)),
               "wbc_compute_torques")
        return tau, f, st.value

    def one_image(self):
        """wbc_one_map: numpy views of the solver's pinned single-robot image (fp64 solvers).  Write q, v, w_des, vdot_des, normals,
        mu, mask[0] (and tau_prev, f_prev, obs_integ, obs_r with the observer on) in place, call one_tick(), read tau, f, status[0]."""
        class _Img(C.Structure):
            _fields_ = [(k, C.POINTER(C.c_double)) for k in ("q", "v", "w_des", "vdot_des", "normals", "mu", "tau_prev", "f_prev", "obs_integ", "obs_r", "tau", "f")] + \
                       [(k, C.POINTER(C.c_int)) for k in ("mask", "status", "iters")]
        img = _Img()
        _check(lib().wbc_one_map(self._h, C.byref(img)), "wbc_one_map")
        m = self.model
        sizes = dict(q=m.nq, v=m.nv, w_des=6, vdot_des=m.nv, normals=3 * m.nf, mu=m.nf, tau_prev=m.nj, f_prev=3 * m.nf, obs_integ=m.nv,
                     obs_r=m.nv, tau=m.nj, f=3 * m.nf, mask=1, status=1, iters=1)
        return {k: np.ctypeslib.as_array(getattr(img, k), shape=(n,)) for k, n in sizes.items()}

    def one_tick(self):
        rc = lib().wbc_one_tick(self._h)
        if rc:
            _check(rc, "wbc_one_tick")

    def observer_init(self, q, v):
        """Observer start-up of the single-robot loop: returns (integ = M(q) v, r = 0) as numpy float64 arrays."""
        d = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        q, v = d(q), d(v)
        integ, r = np.zeros(self.model.nv), np.ones(self.model.nv)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _check(lib().wbc_observer_init(self._h, p(q), p(v), p(integ), p(r)), "wbc_observer_init")
        return integ, r

    def compute_reference(self, q, v, plan, t=0.0):
        """Single-robot host-array planner call (numpy float64): returns (w_des[6], vdot_des[nv], com[6])."""
        d = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        q, v, plan = d(q), d(v), d(plan)
        w, vd, com = np.zeros(6), np.zeros(self.model.nv), np.zeros(6)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _check(lib().wbc_compute_reference(self._h, p(q), p(v), p(plan), C.c_double(t), p(w), p(vd), p(com)),
               "wbc_compute_reference")
        return w, vd, com

    def qp_handover(self):
        """states of the last two-kernel tick that the per-lane QP kernel handed to the dense active-set kernel (synchronises)"""
        c = C.c_int(0)
        _check(lib().wbc_solver_qp_handover(self._h, C.byref(c)), "wbc_solver_qp_handover")
        return c.value

    def enable_timing(self, on=1):
        """0 = off, 1 = HIP events around every kernel, k > 1 = around the kernels of every k-th tick."""
        _check(lib().wbc_solver_enable_timing(self._h, int(on)), "wbc_solver_enable_timing")

    def collect_timing(self):
        ms = (C.c_double * 6)()
        cnt = (C.c_int * 6)()
        _check(lib().wbc_solver_collect_timing_n(self._h, ms, cnt, 6), "wbc_solver_collect_timing_n")
        names = ("dyn", "qp", "rnea", "fused", "qp_lane", "rollout")  # fused = one-kernel tick of small batches; dyn = fused sweep (mass_jac kernel with WBC_SWEEP=split); rnea = rnea_step front half
        out = {}
        for i, n in enumerate(names):
            out[n + "_ms"] = ms[i]
            out[n + "_launches"] = cnt[i]
        return out


def shard_range(n_total, n_shards, shard):
    """wbc_shard_range: (start, count) of a shard's contiguous slice."""
    st, cnt = C.c_size_t(), C.c_size_t()
    _check(lib().wbc_shard_range(n_total, n_shards, shard, C.byref(st), C.byref(cnt)), "wbc_shard_range")
    return st.value, cnt.value


def qp_dense_batch(H, g, Cm=None, d=None, meq=0, max_iter=200, tol=None, want_lambda=True, stream=None):
    """wbc_qp_dense_batch: N dense QPs of one run-time size, one per wavefront (csrc/qp_general.hip.hpp).
        min 1/2 x'Hx + g'x   s.t.   Cm[:, :meq] x = d[:, :meq],   Cm[:, meq:] x >= d[:, meq:]
    H [N, n, n], g [N, n], Cm [N, m, n], d [N, m]: contiguous CUDA tensors of one dtype (float64 / float32), n <= 36, m <= 64.
    Returns dict(x [N, n], lam [N, m] | None, status [N] int32 (0 ok, 1 iteration limit, 2 infeasible, 3 H not PD), iters [N]).
    Enqueued on `stream` (default: torch's current stream); does not synchronise."""
    import torch
    if H.dtype not in (torch.float64, torch.float32):
        raise TypeError("H must be float64 or float32")
    N, n = int(H.shape[0]), int(H.shape[1])
    m = 0 if Cm is None else int(Cm.shape[1])
    ts = [H, g] + ([Cm, d] if m else [])
    for t in ts:
        if not (t.is_cuda and t.is_contiguous() and t.dtype == H.dtype and int(t.shape[0]) == N):
            raise ValueError("qp_dense_batch: contiguous CUDA tensors of one dtype and one batch size")
    if tuple(H.shape) != (N, n, n) or tuple(g.shape) != (N, n) or (m and (tuple(Cm.shape) != (N, m, n) or tuple(d.shape) != (N, m))):
        raise ValueError("qp_dense_batch: shapes H [N,n,n], g [N,n], Cm [N,m,n], d [N,m]")
    if tol is None:
        tol = 1e-9 if H.dtype == torch.float64 else 1e-4
    x = torch.empty((N, n), dtype=H.dtype, device=H.device)
    lam = torch.empty((N, m), dtype=H.dtype, device=H.device) if (want_lambda and m) else None
    status = torch.empty(N, dtype=torch.int32, device=H.device)
    iters = torch.empty(N, dtype=torch.int32, device=H.device)
    st = (stream if stream is not None else torch.cuda.current_stream(H.device)).cuda_stream
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    with torch.cuda.device(H.device):
        _check(lib().wbc_qp_dense_batch(0 if H.dtype == torch.float64 else 1, N, n, m, int(meq), p(H), p(g), p(Cm) if m else None, p(d) if m else None,
                                        int(max_iter), float(tol), p(x), p(lam), p(status), p(iters), C.c_void_p(st)), "wbc_qp_dense_batch")
    return {"x": x, "lam": lam, "status": status, "iters": iters}


GATHER_NONE, GATHER_RCCL, GATHER_PEER_COPY = 0, 1, 2


class MultiSolver:
    """wbc_multi_*: ONE process, one solver per device of the node (the C-ABI path a C++ host uses; the torchrun path of
    bench.py / sharding.py is one process per GPU instead).  devices may repeat a device for NONE / PEER_COPY gathers."""

    def __init__(self, model, params=None, dtype="f64", devices=(0,), max_batch_total=4096, gather="none", options=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible to torch: the WBC hot path has no CPU fallback")
        self.torch, self.model, self.dtype = torch, model, dtype
        self.tdtype = torch.float64 if dtype == "f64" else torch.float32
        self.devices = [int(d) for d in devices]
        self.params = params if params is not None else Params.default(dtype)
        self.options = SolverOptions.make(options)
        self.max_batch_total = int(max_batch_total)
        backend = dict(none=GATHER_NONE, rccl=GATHER_RCCL, peer=GATHER_PEER_COPY)[gather]
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        _check(lib().wbc_multi_create(model._h, C.byref(self.params), F64 if dtype == "f64" else F32, arr, len(self.devices),
                                      self.max_batch_total, backend, C.byref(self.options), C.byref(h)), "wbc_multi_create")
        self._h = h
        self.n = lib().wbc_multi_size(self._h)

    def __del__(self):
        try:
            lib().wbc_multi_destroy(self._h)
        except Exception:
            pass

    @property
    def rccl_ranks(self):
        return lib().wbc_multi_rccl_ranks(self._h)

    def stream(self, k):
        return lib().wbc_multi_stream(self._h, k)

    def synchronize(self):
        _check(lib().wbc_multi_synchronize(self._h), "wbc_multi_synchronize")

    def sync_torch_streams(self):
        """Wait for torch's current stream on every shard device: the shard streams (wbc_multi_stream) do not synchronise
        with torch's streams or the null stream, so buffers produced there must be complete before the first tick."""
        for dev in sorted(set(self.devices)):
            self.torch.cuda.current_stream(self.torch.device("cuda", dev)).synchronize()

    def scatter(self, full, rows, n_total, dtype=None):
        """host/any-device [rows, n_total] tensor -> list of per-shard contiguous [rows, count_k] tensors on devices[k]"""
        outs = []
        for k, dev in enumerate(self.devices):
            st, cnt = shard_range(n_total, self.n, k)
            x = full[..., st:st + cnt].contiguous().to(self.torch.device("cuda", dev))
            outs.append(x.to(dtype) if dtype is not None else x)
        return outs

    def prepare_step(self, n_total, ins, obs=None, want_mats=False, warm=False):
        """ins: dict name -> list of per-shard tensors (q, v, w_des, vdot_des, normals, mu, mask[, tau_prev, f_prev]);
        obs: (list integ, list r) or None.  Returns (tick, outs): tick() enqueues one control tick on every shard
        (wbc_multi_step_batch), outs is a list of per-shard output dicts.  warm: wbc_multi_step_batch_warm -- every shard carries
        its active sets in outs[k]["active"] (int32 [count_k], zero = cold) from tick to tick."""
        torch, m = self.torch, self.model
        BI, BO, OS = (_BatchIn * self.n)(), (_BatchOut * self.n)(), (_ObsState * self.n)()
        outs, keep = [], []
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        for k, dev in enumerate(self.devices):
            _, cnt = shard_range(n_total, self.n, k)
            d = torch.device("cuda", dev)
            o = dict(tau=torch.zeros((m.nj, cnt), dtype=self.tdtype, device=d), f=torch.zeros((3 * m.nf, cnt), dtype=self.tdtype, device=d),
                     status=torch.zeros(cnt, dtype=torch.int32, device=d), iters=torch.zeros(cnt, dtype=torch.int32, device=d))
            if want_mats:
                o.update(M=torch.empty((m.nv * (m.nv + 1) // 2, cnt), dtype=self.tdtype, device=d), h=torch.empty((m.nv, cnt), dtype=self.tdtype, device=d),
                         Jc=torch.empty((3 * m.nf * m.nv, cnt), dtype=self.tdtype, device=d), pf=torch.empty((3 * m.nf, cnt), dtype=self.tdtype, device=d))
            g = lambda name: ins[name][k] if name in ins and ins[name] is not None else None
            for name in ("q", "v", "w_des", "vdot_des", "normals", "mu", "mask", "tau_prev", "f_prev"):
                t = g(name)
                assert t is None or (t.is_cuda and t.is_contiguous() and t.device == d and t.shape[-1] == cnt), (name, k)
            BI[k] = _BatchIn(p(g("q")), p(g("v")), p(g("w_des")), p(g("vdot_des")), p(g("normals")), p(g("mu")), p(g("mask")),
                             p(g("tau_prev")), p(g("f_prev")))
            BO[k] = _BatchOut(p(o["tau"]), p(o["f"]), p(o["status"]), p(o["iters"]), p(o.get("M")), p(o.get("h")), p(o.get("Jc")), p(o.get("pf")))
            if obs is not None:
                OS[k] = _ObsState(p(obs[0][k]), p(obs[1][k]))
            if warm:
                o["active"] = torch.zeros(max(cnt, 1), dtype=torch.int32, device=d)[:cnt]
            outs.append(o)
        ACT = (C.c_void_p * self.n)(*[o["active"].data_ptr() if warm else None for o in outs])
        keep = (ins, obs, outs, BI, BO, OS, ACT)
        fn, h, has_obs = lib().wbc_multi_step_batch, self._h, obs is not None
        fn_warm = lib().wbc_multi_step_batch_warm
        # The shard streams are library-created non-blocking streams: nothing orders them behind the torch streams that
        # produced these buffers (scatter()'s copies, the zero-fills above, a caller's observer state).  One-off cost.
        self.sync_torch_streams()

        def tick(_keep=keep):
            rc = fn_warm(h, n_total, BI, BO, OS if has_obs else None, ACT) if warm else fn(h, n_total, BI, BO, OS if has_obs else None)
            if rc:
                _check(rc, "wbc_multi_step_batch_warm" if warm else "wbc_multi_step_batch")
        tick.capi = (BI, BO, OS, ACT, has_obs)   # (for prepare_tick_gather)
        return tick, outs

    def allgather_tau(self, n_total, outs, tau_all=None):
        """every device receives all torques: returns list (per device) of [n_shards, nj * count_0] tensors"""
        torch, m = self.torch, self.model
        _, c0 = shard_range(n_total, self.n, 0)
        if tau_all is None:
            tau_all = [torch.zeros((self.n, m.nj * c0), dtype=self.tdtype, device=torch.device("cuda", d)) for d in self.devices]
            self.sync_torch_streams()   # the zero-fills run on torch's streams, the gather on the shard streams
        loc = (C.c_void_p * self.n)(*[o["tau"].data_ptr() for o in outs])
        allp = (C.c_void_p * self.n)(*[t.data_ptr() for t in tau_all])
        _check(lib().wbc_multi_allgather_tau(self._h, n_total, loc, allp), "wbc_multi_allgather_tau")
        return tau_all

    def allgather_tau_async(self, n_total, outs, tau_all, slot):
        """wbc_multi_allgather_tau_async: the gather of `outs` (slot 0 / 1 of a double-buffered tau) on the gather streams, beside the next tick"""
        loc = (C.c_void_p * self.n)(*[o["tau"].data_ptr() for o in outs])
        allp = (C.c_void_p * self.n)(*[t.data_ptr() for t in tau_all])
        _check(lib().wbc_multi_allgather_tau_async(self._h, n_total, loc, allp, int(slot)), "wbc_multi_allgather_tau_async")
        return tau_all

    def prepare_tick_gather(self, n_total, ticks, tau_alls, warm=False):
        """wbc_multi_tick_gather as a closure: ticks = [(BI, BO, OS, ACT) per slot] as kept by prepare_step (its _keep tuple), tau_alls = per slot the
        per-device gather buffers.  Returns run(slot): gather_wait(slot) + tick into the slot's tau + overlapped gather of the slot, ONE call."""
        fn, h = lib().wbc_multi_tick_gather, self._h
        slots = []
        for (BI, BO, OS, ACT, has_obs), tau_all in zip(ticks, tau_alls):
            allp = (C.c_void_p * self.n)(*[t.data_ptr() for t in tau_all])
            slots.append((BI, BO, OS if has_obs else None, ACT if warm else None, allp))

        def run(slot, _keep=(ticks, tau_alls, slots)):
            BI, BO, OS, ACT, allp = slots[slot]
            rc = fn(h, n_total, BI, BO, OS, ACT, allp, slot)
            if rc:
                _check(rc, "wbc_multi_tick_gather")
        return run

    @property
    def issue_threads(self):
        return lib().wbc_multi_issue_threads(self._h)

    def set_peer_copies(self, on):
        """wbc_multi_set_peer_copies: 1 = the peer gather never stores through peer mappings (for gather buffers from virtual-memory pools)"""
        _check(lib().wbc_multi_set_peer_copies(self._h, 1 if on else 0), "wbc_multi_set_peer_copies")

    @property
    def gather_pushes(self):
        """1 when the next peer gather would use the push kernel"""
        return lib().wbc_multi_gather_pushes(self._h)

    def host_stats(self, reset=True):
        """(calls, seconds) the caller has spent inside the tick / rollout / gather entry points since the last reset"""
        calls, sec = C.c_ulonglong(), C.c_double()
        _check(lib().wbc_multi_host_stats(self._h, C.byref(calls), C.byref(sec), 1 if reset else 0), "wbc_multi_host_stats")
        return calls.value, sec.value

    def probe_issue(self, iters=2000):
        """microseconds per EMPTY ticket through the issue threads (serial issue: per empty loop over the shards)"""
        sec = C.c_double()
        _check(lib().wbc_multi_probe_issue(self._h, int(iters), C.byref(sec)), "wbc_multi_probe_issue")
        return sec.value / iters * 1e6

    def gather_wait(self, slot):
        """the shard streams wait (on the device) for the last gather of `slot`: before the tick that overwrites that slot's tau"""
        _check(lib().wbc_multi_gather_wait(self._h, int(slot)), "wbc_multi_gather_wait")

    def step_host(self, q, v, w_des, vdot_des, normals, mu, mask, tau_prev=None, f_prev=None, obs_integ=None, obs_r=None):
        """Host-resident batch (numpy, component-major [ncomp, n_total] in the solver's dtype): scatter, tick, gather.
        Observer state arrays are updated in place.  Returns dict(tau, f, status, iters)."""
        nd = np.float64 if self.dtype == "f64" else np.float32
        n_total = q.shape[1]
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=nd)
        q, v, w_des, vdot_des, normals, mu, tau_prev, f_prev = map(c, (q, v, w_des, vdot_des, normals, mu, tau_prev, f_prev))
        mask = np.ascontiguousarray(mask, dtype=np.int32)
        for a in (obs_integ, obs_r):
            assert a is None or (a.dtype == nd and a.flags.c_contiguous)
        m = self.model
        out = dict(tau=np.zeros((m.nj, n_total), nd), f=np.zeros((3 * m.nf, n_total), nd), status=np.zeros(n_total, np.int32),
                   iters=np.zeros(n_total, np.int32))
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        bi = _BatchIn(p(q), p(v), p(w_des), p(vdot_des), p(normals), p(mu), p(mask), p(tau_prev), p(f_prev))
        bo = _BatchOut(p(out["tau"]), p(out["f"]), p(out["status"]), p(out["iters"]), None, None, None, None)
        ob = _ObsState(p(obs_integ), p(obs_r))
        _check(lib().wbc_multi_step_host(self._h, n_total, C.byref(bi), C.byref(bo), C.byref(ob)), "wbc_multi_step_host")
        return out

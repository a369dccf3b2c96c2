// C-ABI implementation (include/wbc_hip.h): host side of the HIP path.  Pure host code -- the kernels live in the
// k_*.hip translation units behind launch.hpp.  No CPU compute fallback exists: without a usable gfx950 device every
// solver entry point fails with WBC_E_NODEVICE/WBC_E_HIP.
#include "../../include/wbc_hip.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "device_types.hpp"
#include "host_internal.hpp"
#include "launch.hpp"
#include "model.hpp"

using namespace wbc;

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield");
#endif
}

static thread_local std::string g_err;
int wbc::fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) return fail(WBC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

// Every entry point runs on the solver's device and leaves the caller's current device as it found it.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = (err == hipSuccess); }
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define ON_DEVICE(s_)                                                                                     \
  DeviceGuard guard_((s_)->device);                                                                       \
  if (guard_.err != hipSuccess) return fail(WBC_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard_.err))

struct wbc_model { FlatModel fm; };

constexpr size_t TIMING_MAX_SPANS = 4096;   // bounded ring: samples beyond it are dropped until the next collect
constexpr int ONE_SCRATCH = 200;        // first scalar of the helpers' scratch region of the single-robot image (88 scalars)
constexpr int ONE_SCALARS = 288;        // scalars in front of the image's ints (the tick uses the first 161)

// thresholds between kernel variants after the options are applied (resolve_options)
constexpr long long WBC_OBS_SPLIT_MIN_NOMATS_F64 = 16384;
constexpr long long WBC_OBS_SPLIT_MIN_NOMATS_F32 = 32768;
constexpr long long WBC_WARM_LANE_MIN_F64 = 53248;
constexpr long long WBC_WARM_LANE_MIN_F32 = 36864;
constexpr long long WBC_TT_WARM_MAX_F32 = 81920;   // warm fp32 observer-on ticks below this run the (cold, set-reporting) tile tick: plan_tick
constexpr long long WBC_COLAUNCH_MIN_F32 = 12289;
constexpr long long WBC_COLAUNCH_MAX_F32 = 32768;
constexpr long long WBC_COLAUNCH_MIN_F64 = 12289;
constexpr long long WBC_COLAUNCH_MAX_F64 = 14336;
struct Resolved { size_t fused_max, fused_max_noobs, fused_max_obs, obs_split_min, obs_split_min_nomats, tile_min, lane_min, warm_tile_min, warm_lane_min, colaunch_min, colaunch_max, stile_min, stile_max, tt_min, tt_max, tt_first_min, tt_max_obs, tt_max_noobs32, pair_min, pair_max; };

struct wbc_solver {
  int dtype = WBC_F64;
  int device = 0;
  size_t max_batch = 0;
  wbc_params params;
  wbc_solver_options opt;
  int leg_body[4][3];
  void* d_model = nullptr;  // DevModel<T>
  void* d_ws = nullptr;     // WS_LDS_WORDS * max_batch * sizeof(T): the 66 step words + 18 words of rhat (separate observer kernel)
  int in_rollout = 0;       // inside the per-tick loop of wbc_rollout_batch, past its first tick
  int* d_todo = nullptr;    // 4 + max_batch ints: states the per-lane QP kernel hands to the dense active-set kernel (count, workgroups done, count of the last tick, pad; indices)
  int* d_aset = nullptr;    // max_batch ints: the active sets that wbc_rollout_batch's per-tick launches carry from tick to tick (rollout_warm)
  QpJidx jmap;
  unsigned long long jpack = 0;   // jmap as nibbles (pack_jidx): the dynamics bodies' joint indices, a kernel argument
  Resolved rz{};            // resolved thresholds (resolve_options)
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  void* d_ref = nullptr;     // DevRefParams<T>, set by wbc_solver_set_ref_params
  // N=1 convenience buffers
  void* d_one = nullptr;
  void* h_one = nullptr;       // pinned host image of d_one: the single-robot calls move it with ONE copy each way
  void* h_one_dev = nullptr;   // device address of h_one (mapped): one_zerocopy lets the N = 1 kernels read / write it directly
  size_t one_bytes = 0;
  // wbc_solver_options.keep_structural: the M / Jc buffers (and N) of the last call that wrote their structural constants
  const void* kept_M = nullptr; const void* kept_Jc = nullptr; size_t kept_N = 0;
  unsigned one_seq = 0;        // completion tickets of the flag-polled single-robot ticks (one_zerocopy >= 2)
  bool one_flag_ok = true;     // cleared when the stream write-value / flag kernel is refused: back to hipStreamSynchronize
  // timing: a ring of event pairs allocated by wbc_solver_enable_timing (never inside a tick)
  bool timing = false;
  int timing_period = 1;   // instrument every timing_period-th tick
  unsigned long long calls = 0;
  bool sample_now = false;
  std::vector<hipEvent_t> ev_pool;   // 2 * TIMING_MAX_SPANS once timing has been enabled
  struct Span { int kind; hipEvent_t a, b; };
  std::vector<Span> spans;
  size_t dropped = 0;
};

// ------------------------------------------------------------------------------------------ model
extern "C" int wbc_model_load_urdf(const char* path, const char* const* foot_links, int n_foot_links, wbc_model** out) {
  if (!path || !out || n_foot_links < 0 || (n_foot_links > 0 && !foot_links)) return fail(WBC_E_INVALID, "null argument");
  *out = nullptr;
  std::vector<std::string> feet;
  for (int i = 0; i < n_foot_links; ++i) {
    if (!foot_links[i]) return fail(WBC_E_INVALID, "null foot link name");
    feet.emplace_back(foot_links[i]);
  }
  wbc_model* m = new (std::nothrow) wbc_model;
  if (!m) return fail(WBC_E_INVALID, "out of memory");
  std::string err;
  int rc;
  try {
    rc = load_urdf(path, feet, m->fm, err);
  } catch (const std::exception& e) {
    rc = WBC_E_PARSE; err = e.what();
  }
  if (rc != WBC_OK) { delete m; return fail(rc, err); }
  *out = m;
  return WBC_OK;
}

extern "C" int wbc_model_from_flat(int nb, const int* parent, const double* Rt, const double* rt, const double* axis,
                                   const double* mass, const double* com, const double* Ic, int nf,
                                   const int* foot_body, const double* foot_off, const double* gravity,
                                   wbc_model** out) {
  if (!out || nb < 1 || nb > 64 || nf < 0 || nf > 16 || !parent || !Rt || !rt || !axis || !mass || !com || !Ic ||
      (nf > 0 && (!foot_body || !foot_off)))
    return fail(WBC_E_INVALID, "bad argument");
  *out = nullptr;
  for (int i = 0; i < nb; ++i)
    if ((i == 0 && parent[i] != -1) || (i > 0 && (parent[i] < 0 || parent[i] >= i)))
      return fail(WBC_E_INVALID, "parent[] must satisfy parent[0] = -1, 0 <= parent[i] < i");
  for (int k = 0; k < nf; ++k)
    if (foot_body[k] < 0 || foot_body[k] >= nb) return fail(WBC_E_INVALID, "foot_body out of range");
  wbc_model* m = new (std::nothrow) wbc_model;
  if (!m) return fail(WBC_E_INVALID, "out of memory");
  FlatModel& f = m->fm;
  f.nb = nb;
  f.parent.assign(parent, parent + nb);
  f.Rt.assign(Rt, Rt + 9 * nb);
  f.rt.assign(rt, rt + 3 * nb);
  f.axis.assign(axis, axis + 3 * nb);
  f.mass.assign(mass, mass + nb);
  f.com.assign(com, com + 3 * nb);
  f.Ic.assign(Ic, Ic + 6 * nb);
  f.foot_body.assign(foot_body, foot_body + nf);
  f.foot_off.assign(foot_off, foot_off + 3 * nf);
  if (gravity) for (int k = 0; k < 3; ++k) f.gravity[k] = gravity[k];
  f.joint_names.assign(nb - 1, "");
  f.foot_links.assign(nf, "");
  f.body_names.assign(nb, "");
  *out = m;
  return WBC_OK;
}

extern "C" void wbc_model_free(wbc_model* m) { delete m; }

extern "C" int wbc_model_dims(const wbc_model* m, int* nb, int* nq, int* nv, int* nj, int* nf) {
  if (!m) return fail(WBC_E_INVALID, "null model");
  if (nb) *nb = m->fm.nb;
  if (nq) *nq = m->fm.nq();
  if (nv) *nv = m->fm.nv();
  if (nj) *nj = m->fm.nj();
  if (nf) *nf = m->fm.nf();
  return WBC_OK;
}

extern "C" int wbc_model_get_flat(const wbc_model* m, int* parent, double* Rt, double* rt, double* axis, double* mass,
                                  double* com, double* Ic, int* foot_body, double* foot_off, double* gravity) {
  if (!m) return fail(WBC_E_INVALID, "null model");
  const FlatModel& f = m->fm;
  auto cp = [](auto* dst, const auto& v) { if (dst) std::memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
  cp(parent, f.parent); cp(Rt, f.Rt); cp(rt, f.rt); cp(axis, f.axis); cp(mass, f.mass); cp(com, f.com); cp(Ic, f.Ic);
  cp(foot_body, f.foot_body); cp(foot_off, f.foot_off);
  if (gravity) for (int k = 0; k < 3; ++k) gravity[k] = f.gravity[k];
  return WBC_OK;
}

extern "C" const char* wbc_model_joint_name(const wbc_model* m, int j) {
  if (!m || j < 0 || j >= (int)m->fm.joint_names.size()) return nullptr;
  return m->fm.joint_names[j].c_str();
}
extern "C" const char* wbc_model_foot_link(const wbc_model* m, int k) {
  if (!m || k < 0 || k >= (int)m->fm.foot_links.size()) return nullptr;
  return m->fm.foot_links[k].c_str();
}
extern "C" double wbc_model_total_mass(const wbc_model* m) {
  if (!m) return 0.0;
  double s = 0;
  for (double x : m->fm.mass) s += x;
  return s;
}

extern "C" void wbc_params_default(wbc_params* p, int dtype) {
  if (!p) return;
  for (int i = 0; i < 6; ++i) p->S[i] = 1.0;
  p->alpha = 1e-3; p->fn_min = 0.0; p->fn_max = 400.0; p->mu_scale = 1.0; p->dt = 1e-3;
  p->observer_order = 0; p->max_iter = 100;
  p->qp_tol = (dtype == WBC_F32) ? 1e-3 : 1e-9;
  for (int i = 0; i < WBC_MAXV; ++i) { p->K1[i] = 50.0; p->K2[i] = 200.0; }
}

// ------------------------------------------------------------------------------------------ solver
template <class T> static void build_dev_model(const FlatModel& f, const int leg_body[4][3], DevModel<T>& d) {
  std::memset(&d, 0, sizeof(d));
  auto put = [&](int leg, int idx, double v) { d.cst[idx * 4 + leg] = (T)v; };
  for (int l = 0; l < 4; ++l) {
    for (int k = 0; k < 3; ++k) {
      const int b = leg_body[l][k], o = JOINT_WORDS * k;
      const double* Rt = &f.Rt[9 * b];
      const double* ax = &f.axis[3 * b];
      double A0[9], A2[9];
      // A0 = Rt a a^T ; A2 = Rt [a]x
      double Ra[3];
      for (int i = 0; i < 3; ++i) Ra[i] = Rt[3 * i] * ax[0] + Rt[3 * i + 1] * ax[1] + Rt[3 * i + 2] * ax[2];
      const double K[9] = {0, -ax[2], ax[1], ax[2], 0, -ax[0], -ax[1], ax[0], 0};
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          A0[3 * i + j] = Ra[i] * ax[j];
          double s = 0;
          for (int q = 0; q < 3; ++q) s += Rt[3 * i + q] * K[3 * q + j];
          A2[3 * i + j] = s;
        }
      for (int e = 0; e < 9; ++e) { put(l, o + e, A0[e]); put(l, o + 9 + e, Rt[e] - A0[e]); put(l, o + 18 + e, A2[e]); }
      for (int e = 0; e < 3; ++e) { put(l, o + 27 + e, f.rt[3 * b + e]); put(l, o + 30 + e, ax[e]); }
      const double m = f.mass[b], *c = &f.com[3 * b], *I = &f.Ic[6 * b];
      put(l, o + 33, m);
      for (int e = 0; e < 3; ++e) put(l, o + 34 + e, m * c[e]);
      const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
      const double Io[6] = {I[0] + m * (cc - c[0] * c[0]), I[1] - m * c[0] * c[1], I[2] - m * c[0] * c[2],
                            I[3] + m * (cc - c[1] * c[1]), I[4] - m * c[1] * c[2], I[5] + m * (cc - c[2] * c[2])};
      for (int e = 0; e < 6; ++e) put(l, o + 37 + e, Io[e]);
      d.jidx[l][k] = b - 1;
    }
    for (int e = 0; e < 3; ++e) put(l, 129 + e, f.foot_off[3 * l + e]);
  }
  const double m = f.mass[0], *c = &f.com[0], *I = &f.Ic[0];
  const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  d.base_m = (T)m;
  for (int e = 0; e < 3; ++e) d.base_h[e] = (T)(m * c[e]);
  const double Io[6] = {I[0] + m * (cc - c[0] * c[0]), I[1] - m * c[0] * c[1], I[2] - m * c[0] * c[2],
                        I[3] + m * (cc - c[1] * c[1]), I[4] - m * c[1] * c[2], I[5] + m * (cc - c[2] * c[2])};
  for (int e = 0; e < 6; ++e) d.base_Io[e] = (T)Io[e];
  for (int e = 0; e < 3; ++e) d.grav[e] = (T)f.gravity[e];
  // structural zeros of the packed mass matrix
  int nz = 0;
  auto mi = [](int i, int j) { if (i > j) { int t = i; i = j; j = t; } return i * 18 - i * (i - 1) / 2 + (j - i); };
  for (int l1 = 0; l1 < 4; ++l1)
    for (int l2 = l1 + 1; l2 < 4; ++l2)
      for (int ka = 0; ka < 3; ++ka)
        for (int kb = 0; kb < 3; ++kb) d.zidx[nz++] = mi(6 + d.jidx[l1][ka], 6 + d.jidx[l2][kb]);
  const int bz[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 4}, {2, 5}};
  for (auto& e : bz) d.zidx[nz++] = mi(e[0], e[1]);
  for (; nz < 64; ++nz) d.zidx[nz] = -1;
}

template <class T> static DevParams<T> to_dev_params(const wbc_params& p) {
  DevParams<T> d;
  for (int i = 0; i < 6; ++i) { d.S[i] = (T)p.S[i]; d.sS[i] = (T)std::sqrt(p.S[i]); }
  d.sqrt_alpha = (T)std::sqrt(p.alpha); d.rsqrt_alpha = (T)(1.0 / std::sqrt(p.alpha));
  d.alpha = (T)p.alpha; d.fn_min = (T)p.fn_min; d.fn_max = (T)p.fn_max; d.mu_scale = (T)p.mu_scale;
  d.dt = (T)p.dt; d.qp_tol = (T)p.qp_tol;
  d.observer_order = p.observer_order; d.max_iter = p.max_iter;
  for (int i = 0; i < 18; ++i) { d.K1[i] = (T)p.K1[i]; d.K2[i] = (T)p.K2[i]; }
  return d;
}

static int check_params(const wbc_params* p) {
  if (!p) return fail(WBC_E_INVALID, "null params");
  if (!(p->alpha > 0)) return fail(WBC_E_INVALID, "alpha must be > 0 (strict convexity of the GRF QP)");
  if (!(p->fn_max >= p->fn_min)) return fail(WBC_E_INVALID, "fn_max < fn_min");
  if (p->observer_order < 0 || p->observer_order > 2) return fail(WBC_E_INVALID, "observer_order must be 0, 1 or 2");
  if (p->max_iter < 1) return fail(WBC_E_INVALID, "max_iter must be >= 1");
  for (int i = 0; i < 6; ++i) if (!(p->S[i] >= 0)) return fail(WBC_E_INVALID, "S must be >= 0");
  return WBC_OK;
}

int wbc::check_params_public(const wbc_params* p) { return check_params(p); }

extern "C" void wbc_solver_options_default(wbc_solver_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->struct_size = sizeof(*o);
  o->fused_max = -1;
  o->rollout_persistent = 1;
  o->rollout_spw = 0;
  o->obs_split_min = -1;   // auto
  o->one_zerocopy = 3;   // measured p50 of a one-robot tick on the image (us): 0: 23.0, 1: 20.6, 2: 18.8, 3: 17.1
  o->timing_mode = WBC_TIMING_DISPATCH;
  o->qp_tile = 0;
  o->obs_split_serial = 1;
  o->qp_lane = 0;
  o->f32_pack2 = 0;
  o->keep_structural = 0;
  o->rollout_warm = 1;
  o->multi_threads = 0;
  o->multi_spin_us = 200;
  o->obs_colaunch = 0;
  o->tile_tick = 0;
  o->fused_pair = 0;
}

// ------------------------------------------------------------------------------------------ which kernels run a tick
// The measured switches between kernel variants, in ONE place: step_impl launches what plan_tick says, wbc_plan_tick /
// wbc_dispatch_thresholds report it (tests/test_gpu_parity.py straddles every switch with it, so a moved threshold moves the test).
static Resolved resolve_options(int dtype, const wbc_solver_options& o) {
  Resolved r;
  // rollouts of at most fused_max states run as ONE persistent launch (rollout_kernel); ticks of at most fused_max_noobs / fused_max_obs states run as
  // ONE kernel (fused_tick.hip.hpp): it still wins with two rounds of workgroups -- since round 3 also with the observer on in fp64
  // (M steps/s, two-kernel -> fused: 5 120 states 175 -> 233, 6 144: 220 -> 287, 8 192: 258 -> 331) -- and loses from 12 288 on
  // End of round 4 (the observer-on tick's QP no longer waits for rhat, warm ticks end with the dynamics roles): the one-launch tick still wins with THREE
  // rounds of workgroups.  M steps/s, two-kernel -> fused: fp64 observer on 9 216: 291 -> 318, 10 240: 317 -> 351, 12 288: 347 -> 377, but 13 312: 384 -> 353;
  // fp32 observer on 9 216: 281 -> 344, 12 288: 380 -> 406, 13 312: 360 -> 376, 14 336: 385 -> 347; fp64 observer off 9 216: 295 -> 301, 11 264: 309 -> 328,
  // 12 288: 348 -> 342 (warm ticks in a closed loop, kernels in us: observer off 10 240: 30.9 -> 27.6, 12 288: 32.7 -> 30.2; on 10 240: 34.8 -> 32.7).
  r.fused_max = 4096; r.fused_max_noobs = 11264; r.fused_max_obs = 12288;
  if (o.fused_max >= 0) r.fused_max = r.fused_max_noobs = r.fused_max_obs = (size_t)o.fused_max;
  // observer as its own kernel before the sweep (the all-in-one observer sweep runs one wavefront per SIMD).  Measured on
  // MI355X, front half of the tick, all-in-one -> observer kernel + observer-free sweep (us): fp64 464 -> 112 + 279 at
  // 262 144 states, 111 -> 34 + 57 at 65 536, but 48 -> 25 + 33 at 32 768; fp32 298 -> 57 + 168 at 262 144, 45 -> 17 + 27 at
  // 65 536 (a tie per tick), 26 -> 13 + 16 at 32 768.  At the final kernels (observer-free sweep without the w_des forwarding,
  // both observer forms with their inputs requested up front) per tick, all-in-one -> split (M steps/s): fp64 437 -> 472 at
  // 65 536, 428 -> 446 at 49 152, 405 -> 439 at 40 960, 451 -> 460 at 32 768, 360 -> 398 at 24 576, 308 -> 341 at 20 480, but
  // 330 -> 289 at 16 384; fp32 855 -> 934 at 98 304, 784 -> 798 at 65 536, 681 -> 717 at 49 152, 594 -> 629 at 40 960, but
  // 625 -> 587 at 32 768.  Round 3 (M steps/s): fp32 570 -> 593 at 34 816, 599 -> 614 at 36 864, 612 -> 638 at 38 912 (past 32 768 states the
  // all-in-one fp32 sweep needs a second round of wavefronts: 23.7 -> 37 us).  Default: fp64 from 20 480 states on, fp32 from 33 792.
  r.obs_split_min = (size_t)-1;
  if (o.obs_split_min >= 0) r.obs_split_min = (size_t)o.obs_split_min;
  else if (o.obs_split_min == -1) r.obs_split_min = dtype == WBC_F32 ? 33792 : 20480;
  // ticks without M / h / Jc outputs: the all-in-one observer form of rnea_step against observer kernel + observer-free rnea_step.  Measured
  // (tools/ab_sweep.sh with AB_EXTRA=--no-mats, M steps/s, all-in-one -> split): fp64 trot batch 12 288: 460 -> 409, 16 384: 393 -> 449, 24 576: 528 -> 581,
  // 32 768: 542 -> 648, 65 536: 580 -> 792, 131 072: 649 -> 940, 262 144: 606 -> 967 (rnea_step with the observer inside 332 us; observer kernel 100 +
  // observer-free rnea_step 68); fp32 24 576: 688 -> 637, 32 768: 720 -> 759, 65 536: 966 -> 1 163, 131 072: 1 160 -> 1 441, 262 144: 1 149 -> 1 539.
  // (The two kernels on two streams lose to one after the other at every size, as with the sweep.)
  r.obs_split_min_nomats = o.obs_split_min >= 0 ? (size_t)o.obs_split_min
                           : (o.obs_split_min == -1 ? (size_t)(dtype == WBC_F32 ? WBC_OBS_SPLIT_MIN_NOMATS_F32 : WBC_OBS_SPLIT_MIN_NOMATS_F64) : (size_t)-1);
  // tiles dealt by predicted work (auto): fp64 from 14 336 states, fp32 from 30 720; per-lane QP pair (auto): fp64 from 106 496, fp32 from 212 992
  r.tile_min = dtype == WBC_F32 ? 30720 : 14336;
  r.lane_min = dtype == WBC_F32 ? 212992 : 106496;
  // staged tiles (round 6, fp32 solvers; qp_stile_kernel): ONE workgroup of twelve wavefronts per CU holds the CU's share of the batch and its inputs in LDS.
  // Measured on MI355X (profiles/r06b_ab_staged_tiles.log; QP stage in us, one-wavefront workgroups or gathered tiles -> staged): fp32 trot batch 16 384: 17.5 -> 16.2,
  // 20 480: 19.9 -> 17.8, 24 576: 20.1 -> 15.6, 28 672: 22.5 -> 17.1, 32 768 (configs[3]'s shard): 22.6 -> 17.2, 40 960: 26.5 -> 20.8, 49 152: 26.9 -> 21.6
  r.stile_min = dtype == WBC_F32 ? (size_t)WBC_STILE_MIN_F32 : (size_t)-1;
  r.stile_max = dtype == WBC_F32 ? wbc::STILE_MAX_STATES : 0;
  // the whole tick as one launch of 64 / 96 / 128-state workgroups (round 6; tile_tick_kernel): fp32, observer on, M / h / Jc outputs, even N.  Measured on MI355X
  // (profiles/r06c_ab_tile_tick.log; M steps/s, sweep_obs / observer + sweep -> staged or gathered tiles / per-lane pair against the tile tick): 12 800: 445 -> 492,
  // 16 384: 521 -> 613, 24 576: 750 -> 843, 32 768 (configs[3]'s shard): 913 -> 1 059, 36 864 (a second, nearly empty round of workgroups): 694 -> 719, 49 152: 851 -> 892,
  // 65 536: 910 -> 1 123, 98 304: 1 045 -> 1 132, 131 072: 1 002 -> 1 038, 196 608: 927 -> 1 078, 262 144: 993 -> 1 049 -- every size measured, so: no upper limit
  // (auto only while the kernel-selection options of the two-launch tick are at auto themselves: a caller who names a QP kernel or a front half gets it)
  const bool tt_auto_ok = o.qp_tile == 0 && o.qp_lane == 0 && o.obs_colaunch == 0 && o.obs_split_min == -1;
  r.tt_min = (size_t)-1; r.tt_max = 0; r.tt_first_min = (size_t)-1; r.tt_max_noobs32 = 0;
  // (fp32 below the one-launch tick's limit, profiles/r06o_tile_tick_f32_small.log, one-launch -> tile tick: 8 192: 347 -> 335 but 8 704: 337 -> 353, 10 240: 357 -> 434, 12 288: 413 -> 490:
  //  from 8 194 states the tile tick goes in front of the one-launch tick while fused_max is at auto -- tt_first_min, as for fp64 below)
  if (dtype == WBC_F32 && (o.tile_tick > 0 || (o.tile_tick == 0 && tt_auto_ok))) {
    r.tt_min = o.tile_tick > 0 ? 2 : (size_t)WBC_TILE_TICK_MIN; r.tt_max = (size_t)-1;
    if (o.tile_tick == 0 && o.fused_max < 0) r.tt_first_min = r.tt_min;
    // observer off (fp32): forced only.  Measured on the standing batch (profiles/r06q_tile_tick_f32_noobs.log; M steps/s, default -> tile tick): 16 384: 490 -> 484,
    // 24 576: 610 -> 633, 32 768: 688 -> 800, but 49 152: 807 -> 738, 10 240: 343 -> 307, 262 144: 1 254 -> 850 -- no range worth a default
    r.tt_max_noobs32 = o.tile_tick > 0 ? (size_t)-1 : 0;
  }
  // ... and of fp64 observer-off batches (32 ... 112-state workgroups: NS = 2 ... 7 sweep wavefronts; small tiles get helper wavefronts for the QP stage).  fp64 QPs of
  // the standing batch iterate 2.5 times per state against the trot batches' 0.5, so the QP stage weighs more and the gain is small, and only while the batch is ONE
  // round of workgroups (profiles/r06d_ab_tile_tick_f64.log; M steps/s, default -> tile tick): 11 264: 333 -> 355, 12 288: 347 -> 379, 16 384: 425 -> 475, 24 576: 531 -> 550,
  // 28 672: 541 -> 568; but 8 192: 343 -> 287 (the fused tick's role split overlaps QP and dynamics), 32 768: 602 -> 376 (a second round), 262 144: 790 -> 550
  // With tau_partial handed over in LDS (profiles/r06j_ab_tile_tick_lds_handover.log, r06k_tile_tick_f64_range.log, r06m_tile_tick_f64_small.log; default -> tile tick): 9 216: 303 -> 324,
  // 10 240: 319 -> 350, 11 264: 333 -> 378, 12 288: 348 -> 404, 16 384: 427 -> 501, 24 576: 532 -> 575, 28 672: 546 -> 596; 8 192 (two full rounds of the one-launch tick): 342 -> 305.
  // So from 8 193 states on the tile tick also goes IN FRONT of the one-launch tick (tt_first_min) -- while the caller leaves fused_max at auto.
  r.tt_max_obs = 0;
  if (dtype == WBC_F64 && (o.tile_tick > 0 || (o.tile_tick == 0 && tt_auto_ok))) {
    r.tt_min = o.tile_tick > 0 ? 2 : (size_t)WBC_TILE_TICK_MIN_F64; r.tt_max = o.tile_tick > 0 ? (size_t)-1 : (size_t)7 * 16 * 256;
    if (o.tile_tick == 0 && o.fused_max < 0) r.tt_first_min = r.tt_min;
    // observer on (NS sweep + NS observer wavefronts of 16 states: 32 / 48 / 64-state workgroups, 64 in rounds beyond 16 384 states), profiles/r06n_tile_tick_f64_obs.log,
    // M steps/s default -> tile tick: 8 192: 358 -> 326 (one-launch tick, two full rounds) but 8 704: 311 -> 350, 10 240: 358 -> 398, 12 288: 388 -> 461; against the
    // two-launch ticks 12 800: 355 -> 399, 16 384: 416 -> 567, 24 576: 499 -> 525, 32 768: 532 -> 639, 49 152: 614 -> 656, 65 536: 553 -> 578, 98 304: 557 -> 597,
    // 131 072: 580 -> 586, 196 608: 571 -> 598, but 262 144: 611 -> 574
    r.tt_max_obs = o.tile_tick > 0 ? (size_t)-1 : (size_t)WBC_TILE_TICK_MAX_F64_OBS;
  }
  // the one-launch tick as 32-state, twelve-wavefront workgroups (fused_pair_kernel, fused_tick.hip.hpp): observer off, cold, M / h / Jc outputs, N >= 64.
  // Both halves of a pair are resident together, so a round of workgroups is 8 192 states instead of 4 096 -- at the price of the rnea role's spill (168 registers).
  r.pair_min = (size_t)-1; r.pair_max = 0;
  // fp32 (observer off): the pair holds 168 registers WITHOUT a spill (the six-wavefront workgroup 152), a pair lasts 15.3 us against 12.4 -- ahead of every other plan from one round of
  // 16-state workgroups up to two rounds of pairs (profiles/r06zzz_ab_pair_f32.log, M steps/s default -> pair: 5 000: 245 -> 283, 6 144: 285 -> 352, 8 192: 364 -> 457, 10 240: 341 -> 364,
  // 12 288: 369 -> 432, 16 384: 485 -> 527; 24 576: 608 -> 507-549)
  if (o.fused_pair > 0) { r.pair_min = 64; r.pair_max = 65536; }   // (from 64 states: the QP wavefronts of a tail workgroup test their unshifted slots 16 p + 16 .. 31 against N, fused_tick.hip.hpp)
  else if (o.fused_pair == 0 && o.fused_max < 0 && o.tile_tick == 0 && tt_auto_ok) { r.pair_min = (size_t)WBC_FUSED_PAIR_MIN; r.pair_max = (size_t)(dtype == WBC_F32 ? WBC_FUSED_PAIR_MAX_F32 : WBC_FUSED_PAIR_MAX); }
  r.warm_tile_min = dtype == WBC_F32 ? r.tile_min : 24576;   // (warm ticks: the one-wavefront kernel with the block set-up up to here; plan_tick)
  r.warm_lane_min = dtype == WBC_F32 ? WBC_WARM_LANE_MIN_F32 : WBC_WARM_LANE_MIN_F64;   // (measured: tools/warm_loop.py with WARM_LOOP_LANE=1; plan_tick)
  // observer update + observer-free sweep as the two roles of ONE launch (sweep_obs_kernel, observer.hip.hpp): while both roles' wavefronts are resident
  // together -- fp32 with two states per lane: 2 x N / 32 <= 2 048 wavefronts, i.e. up to 32 768 states, BASELINE's configs[3] shard; fp64 (228 registers,
  // 21 kB of LDS per one-wavefront workgroup: seven per CU, 1 792 on the device) up to 14 336 states.  Measured (profiles/r05b_ab_colaunch_*.log, M steps/s,
  // all-in-one observer sweep -> two roles): fp32 12 290: 351 -> 393, 16 384: 445 -> 498, 20 480: 498 -> 551, 24 576: 586 -> 651, 28 672: 639 -> 679,
  // 32 768: 732 -> 779 (the two bodies as two kernels one after the other: 315 / 411 / 467 / 536 / 594 / 630); fp64 12 800: 321 -> 352, 13 312: 385 -> 420,
  // 14 336: 386 -> 448, but 16 384 (two rounds): 418 -> 405.
  r.colaunch_min = (size_t)-1; r.colaunch_max = 0;
  if (o.obs_colaunch > 0) { r.colaunch_min = 0; r.colaunch_max = (size_t)-1; }
  else if (o.obs_colaunch == 0) { r.colaunch_min = dtype == WBC_F32 ? WBC_COLAUNCH_MIN_F32 : WBC_COLAUNCH_MIN_F64; r.colaunch_max = dtype == WBC_F32 ? WBC_COLAUNCH_MAX_F32 : WBC_COLAUNCH_MAX_F64; }
  return r;
}

struct TickPlan { int fused, front, qp, tile, qp_body, pack2, sweep_block, qp_warm; bool obs_split, lane; };
static TickPlan plan_tick(int dtype, int observer_order, const wbc_solver_options& o, const Resolved& r, size_t N, bool mats, bool pf, bool warm = false) {
  TickPlan p{};
  const bool ob = observer_order > 0, f32 = dtype == WBC_F32;
  // (warm ticks: behind the one-launch warm tick and below the warm per-lane pair the COLD tile tick -- its staged tiles report the sets -- beats the two-launch warm plans:
  //  closed loops of 16 384 drifting states, wall us per tick, warm one-wavefront kernels -> tile tick: fp64 observer off 41.8 -> 38.8, on 44.8 -> 37.1, fp32 37.8 -> 34.4;
  //  profiles/r06zz_warm_loop_large.log, r06t_warm_loop_tile_tick.log)
  //  fp32 against the warm per-lane pair: 49 152: 75.6 -> 66.5, 65 536: 80.8 -> 76.6, 81 920: 94.0 / 95.4, 98 304: 101 / 108 -- so up to 81 920 states; fp64: up to warm_lane_min)
  const bool tt_warm_ok = !warm || (N < (f32 ? (size_t)WBC_TT_WARM_MAX_F32 : r.warm_lane_min) && o.qp_lane <= 0);
  const bool tt64 = mats && !ob && !f32 && tt_warm_ok && N >= r.tt_min && N <= r.tt_max;   // fp64, observer off (configs[1]'s shape): sweep wavefronts, then the staged QP tile of their states
  // fp64, observer on (configs[2]'s shape): NS sweep + NS observer wavefronts, then the staged QP tile; 64-state workgroups in rounds beyond 16 384 states
  const bool tt64o = mats && ob && !f32 && tt_warm_ok && N >= r.tt_min && N <= r.tt_max_obs;
  // fp32, observer on (configs[3]'s shape), even batches: packed sweep + observer wavefronts, staged QP tile
  const bool tt32 = mats && ob && f32 && (N & 1) == 0 && o.f32_pack2 >= 0 && N >= r.tt_min && N <= r.tt_max && tt_warm_ok;
  const bool tt32n = mats && !ob && f32 && (N & 1) == 0 && o.f32_pack2 >= 0 && N >= r.tt_min && N <= r.tt_max_noobs32 && tt_warm_ok;   // ... observer off
  if (mats && !ob && !warm && N >= r.pair_min && N <= r.pair_max) {   // observer off, cold: 32-state workgroups of the one-launch tick
    p.fused = 3;
    return p;
  }
  if (tt32n && !warm && N >= r.tt_first_min) {
    p.fused = 2; p.front = 0; p.pack2 = 1; p.sweep_block = 64; p.qp = 1; p.tile = wbc::tile_tick_states(N); p.qp_body = 2;
    return p;
  }
  if (tt32 && !warm && N >= r.tt_first_min) {
    p.fused = 2; p.front = 4; p.obs_split = true; p.pack2 = 1; p.sweep_block = 64; p.qp = 1; p.tile = wbc::tile_tick_states(N); p.qp_body = 2;
    return p;
  }
  if (tt64 && !warm && N >= r.tt_first_min) {
    p.fused = 2; p.front = 0; p.sweep_block = 64; p.qp = 1; p.tile = wbc::tile_tick_states_f64(N); p.qp_body = 2;
    return p;
  }
  if (tt64o && !warm && N >= r.tt_first_min) {
    p.fused = 2; p.front = 4; p.obs_split = true; p.sweep_block = 64; p.qp = 1; p.tile = wbc::tile_tick_states_f64_obs(N); p.qp_body = 2;
    return p;
  }
  if ((mats || !pf) && N <= (ob ? r.fused_max_obs : r.fused_max_noobs)) {
    // small batch: one launch, 16 states per workgroup, rnea_step | mass_jac | [observer] | QP as wavefront roles and the
    // workspace through LDS (fused_tick.hip.hpp)
    p.fused = 1;
    p.qp_warm = warm;
    return p;
  }
  if (tt64) {
    p.fused = 2; p.front = 0; p.sweep_block = 64; p.qp = 1; p.tile = wbc::tile_tick_states_f64(N); p.qp_body = 2;
    return p;
  }
  if (tt64o) {
    p.fused = 2; p.front = 4; p.obs_split = true; p.sweep_block = 64; p.qp = 1; p.tile = wbc::tile_tick_states_f64_obs(N); p.qp_body = 2;
    return p;
  }
  if (tt32n) {
    p.fused = 2; p.front = 0; p.pack2 = 1; p.sweep_block = 64; p.qp = 1; p.tile = wbc::tile_tick_states(N); p.qp_body = 2;
    return p;
  }
  if (tt32) {
    // mid-size fp32 observer-on batch: sweep | observer roles and the staged QP tile of the same 128 states per workgroup, one launch (tile_tick.hip.hpp).
    // (warm ticks: only where the cold tiles are the plan anyway -- the kernel reports the sets; from warm_lane_min on the warm per-lane pair stays faster:
    //  230 against 250 us at 262 144 states)
    p.fused = 2; p.front = 4; p.obs_split = true; p.pack2 = 1; p.sweep_block = 64; p.qp = 1; p.tile = wbc::tile_tick_states(N); p.qp_body = 2;
    return p;
  }
  // Large batches solve the QPs ONE STATE PER LANE first (qp_lane_kernel: semismooth Newton on the residual wrench, 64 QPs
  // per wavefront, no cross-lane traffic); the few per cent it does not finish go through a device-side list to the dense
  // active-set kernel.  No host read: the list length stays on the device, the second launch is grid-stride over it.
  // End of round 3, tiles with the predictor hand-over and the scalar weights against the per-lane pair, M steps/s.  fp64 standing batch: a tie from
  // 53 248 to 98 304 states (605 / 605, 608 / 605, 606 / 604, 621 / 625, 638 / 639), then the pair: 636 / 690 at 114 688, 644 / 730 at 131 072; fp64 trot
  // batch: tiles 531 / 483 at 53 248, 533 / 463 at 57 344, 537 / 503 at 65 536, 554 / 520 at 81 920, 546 / 534 at 98 304, pair 527 / 554 at 114 688.
  // fp32 trot batch: tiles 1 043 / 919 at 98 304, 992 / 909 at 131 072, 934 / 911 at 196 608, pair 907 / 951 at 229 376.
  // Hence the default: fp64 from 106 496 states on, fp32 from 212 992 (history of the threshold: docs/DESIGN_R04.md 4.3a).
  // wbc_step_batch_warm (dependent ticks), beyond the fused size.  Below warm_tile_min (fp64 24 576 states, fp32 the cold tile_min) the
  // one-wavefront kernel with the block set-up (qp_struct16.hip.hpp); from warm_lane_min on the per-lane kernel started from the previous FACES (one Newton step confirms them;
  // qp_lane.hip.hpp) with the hand-over list behind it; in between the COLD tiles, which only report the sets (qp_warm = 0): the warm
  // one-wavefront kernel holds 232-252 registers and loses to them there, and the per-lane pair has a floor of two dependent launches (a
  // wavefront of the lane kernel 14-19 us, then the hardest handed-over state's cold solve, 11-15 us).  Closed loops of drifting states on
  // MI355X, tick in us, cold tiles / warm one-wavefront / warm per-lane: fp64 observer on 36 864: 82.6 / 91.0 / 89.7, 49 152: 102.4 / 110.5 /
  // 104.8, 57 344: 124.2 / 133.3 / 121.4, 65 536: 135 / 148 / 129, 262 144: 474 / 533 / 414; fp64 observer off 32 768: 63.8 / 61.9 / 71.7,
  // 49 152: 88.3 / 85.5 / 87.9, 65 536: 120 / 115 / 108, 98 304: 168 / 170 / 153; fp32 (QP stage alone) 36 864: 35.5 / 36.3 / 32.4,
  // 57 344: 42.7 / 50.6 / 34.2, 98 304: 64.9 / 80.6 / 41.4, whole tick 262 144: 300 / 378 / 230.  wbc_solver_options.qp_lane = -1 / 1 forces either.
  // Since the set-up drops rows with negative multipliers instead of restarting cold, the warm one-wavefront kernel against the cold tiles
  // (tick in us, cold / warm one-wavefront): fp64 observer off 16 384: 45.4 / 42.2, 24 576: 52.9 / 49.4, 32 768: 61.8 / 58.0, 49 152: 82.6 / 78.7;
  // fp64 observer on (a trot batch whose cold solves need 0.9 iterations) 16 384: 46.8 / 45.6, 24 576: 61.8 / 62.2, 32 768: 71.3 / 76.4,
  // 49 152: 98.2 / 105.0; fp32 16 384: 45.2 / 43.4, 24 576: 53.0 / 52.6, 32 768: 58.0 / 57.8, 49 152: 78.5 / 82.3 -- it depends on how hard
  // the batch's cold solves are; the default takes it up to 24 576 fp64 states, where it wins or ties on both.
  p.lane = warm ? (o.qp_lane > 0 || (o.qp_lane == 0 && N >= r.warm_lane_min)) : (o.qp_lane > 0 || (o.qp_lane == 0 && N >= r.lane_min));
  if (!mats) {   // no M, h, Jc wanted: the CRBA-free rnea_step kernel is the front half -- large observer-on batches: observer kernel + observer-free rnea_step
    p.obs_split = ob && N >= r.obs_split_min_nomats;
    p.front = p.obs_split ? 3 : 1;
  }
  else if (ob && N >= r.colaunch_min && N <= r.colaunch_max && (!f32 || o.obs_colaunch > 0 || ((N & 1) == 0 && o.f32_pack2 >= 0))) { p.front = 4; p.obs_split = true; }   // ... as the two roles of one launch
  else if (ob && N >= r.obs_split_min) { p.front = 2; p.obs_split = true; }   // observer kernel + observer-free sweep
  else p.front = 0;
  if (p.front != 1 && p.front != 3) {   // what the dyn_sweep launcher picks (k_sweep.hip)
    const bool obs_variant = p.front == 0 && ob;
    p.pack2 = f32 && (N & 1) == 0 && o.f32_pack2 >= 0 && (o.f32_pack2 > 0 || N >= (size_t)WBC_PACK2_MIN_STATES || p.front == 4);   // (the two-role launch packs every even fp32 batch)
    const size_t threads = ((N + (p.pack2 ? 2 : 1) - 1) / (p.pack2 ? 2 : 1)) * 4;
    p.sweep_block = (!obs_variant && p.front != 4 && threads >= wbc::BIG_GRID_THREADS) ? 256 : 64;
  }
  // two-kernel ticks deal tiles of states to the wavefronts by predicted work (qp_tile_kernel).  Measured on MI355X, fp64,
  // QP kernel alone, one-wavefront workgroups -> tiles: 34.7 -> 32.7 us at 12 288 states (tiles of 32), 62.4 -> 48.3 at
  // 32 768, 109.6 -> 93.1 at 65 536, 408 -> 350 at 262 144 (tiles of 64; 128 and 256 lose to the workgroup lifetime);
  // below 12 288 states the tiles do not fill the device
  // (round 3, structured QP body, QP stage in us.  fp64 standing batch: one-wave 19.0 / tiles-of-32 19.6 at 12 288; 22.2 / 20.0 / 64: 23.9 at
  //  16 384; 31.2 / 24.6 / 30.3 at 24 576; 32.4 / 31.9 / 29.9 at 28 672; 35.9 / 34.7 / 30.4 at 32 768.  fp32 -- whose tile kernel holds 180
  //  registers, two workgroups per CU, where the one-wave kernel runs four wavefronts per SIMD -- trot batch: 17.0 / 20.8 / 20.5 at 16 384,
  //  19.6 / 31.2 / 20.5 at 24 576, 22.1 / 33.4 / 22.0 at 28 672, 24.3 / 35.7 / 22.7 at 32 768; standing batch 31.3 / 44.1 / 37.2 at 24 576)
  // ONE ROUND OF RESIDENT WORKGROUPS: the tile kernel keeps three workgroups on a CU (146 registers in fp64, 159 in fp32; 768 on the device),
  // and a launch with a few workgroups more than that runs a second, nearly empty round -- QP stage at 36 864 fp64 states: tiles of
  // 64 (576 workgroups, 2.25 per CU) 37.8 us, of 48 (768) 30.8 us; fp32 at 40 960: 64 -> 36.7, 80 (512 workgroups) -> 27.2.  So the tile is the
  // smallest size (steps of 4 / 8: k_qp.hip) that fits the batch into one round.
  int tile = warm ? (o.qp_tile == 0 && N >= r.warm_tile_min && !(f32 && N > 65536) ? 0 : -1) : o.qp_tile;   // (warm: the auto tiles only, and never the fp32 12 x 12 body, which reports no set)
  bool staged = false;
  if (tile == 0) {
    if (f32) {
      if (N >= (warm ? r.warm_tile_min : r.stile_min) && N <= r.stile_max) { tile = (int)(((N + 255) / 256 + 3) / 4 * 4); staged = true; }   // one workgroup per CU
      else if (N >= r.tile_min && N <= 65536) { tile = (int)(((N + 767) / 768 + 7) / 8 * 8); tile = tile < 64 ? 64 : tile; }   // (159 registers since the QP weights stay scalar: three workgroups per CU, but 64-state tiles at two per CU beat 44-state ones at three: 22.2 vs 24.1 us at 32 768)
      else if (N > 65536) {                    // (beyond: the leaner fp32 body, FOUR workgroups per CU -- k_qp.hip: one round is 1 024 tiles)
        tile = (int)(((N + 1023) / 1024 + 7) / 8 * 8);
        tile = tile > 128 ? 64 : tile;         // (more than one round of 128-state tiles: many rounds of 64-state ones)
      }
    } else if (N >= r.tile_min) {
      tile = (int)(((N + 767) / 768 + 3) / 4 * 4);
      tile = tile < 32 ? 32 : (tile > 64 ? 64 : tile);
    }
  }
  if (tile < 0) tile = 0;
  if (p.lane) { p.qp = 2; p.tile = 0; }
  else { p.qp = tile > 0 ? 1 : 0; p.tile = tile; }
  p.qp_warm = warm && p.qp != 1;
  if (!warm && o.qp_tile > 0 && f32 && tile <= wbc::STILE_MAX_TILE && tile % 4 == 0 && N < (size_t)WBC_F32_DENSE_TILE_MIN) staged = true;   // (an explicit qp_tile: staged where the kernel exists)
  p.qp_body = (p.qp == 1 && staged) ? 2 : ((p.qp == 1 && f32 && tile >= 64 && tile <= 128 && N >= (size_t)WBC_F32_DENSE_TILE_MIN) ? 1 : 0);
  return p;
}

static int options_from_caller(const wbc_solver_options* opt, wbc_solver_options& o) {
  wbc_solver_options_default(&o);
  if (opt) {
    if (opt->struct_size == 0 || opt->struct_size > sizeof(o)) return fail(WBC_E_INVALID, "wbc_solver_options.struct_size is not set (call wbc_solver_options_default first)");
    std::memcpy(&o, opt, opt->struct_size);   // a caller built against an older, shorter struct keeps the newer defaults
    o.struct_size = sizeof(o);
  }
  return WBC_OK;
}

static void plan_to_public(const TickPlan& p, wbc_tick_plan* out) {
  wbc_tick_plan t;
  std::memset(&t, 0, sizeof(t));
  t.struct_size = sizeof(t);
  t.fused = p.fused; t.front = p.front; t.qp = p.qp; t.qp_tile = p.tile; t.qp_body = p.qp_body; t.sweep_pack2 = p.pack2; t.sweep_block = p.sweep_block;
  t.qp_warm = p.qp_warm;
  const size_t n = out->struct_size && out->struct_size < sizeof(t) ? out->struct_size : sizeof(t);
  std::memcpy(out, &t, n);
  out->struct_size = n;
}

extern "C" int wbc_plan_tick(int dtype, int observer_order, const wbc_solver_options* opt, size_t N, int with_mats, int with_pf, int warm,
                             wbc_tick_plan* plan) {
  if (!plan || (dtype != WBC_F64 && dtype != WBC_F32) || observer_order < 0 || observer_order > 2) return fail(WBC_E_INVALID, "bad argument");
  wbc_solver_options o;
  const int rc = options_from_caller(opt, o);
  if (rc) return rc;
  plan_to_public(plan_tick(dtype, observer_order, o, resolve_options(dtype, o), N, with_mats != 0, with_pf != 0, warm != 0), plan);
  return WBC_OK;
}

// the batch sizes N at which plan(N) differs from plan(N - 1), ascending (the one-round tile SIZE is not a switch of kernel family
// and is left out: it steps every 3 072 / 6 144 / 8 192 states)
extern "C" int wbc_dispatch_thresholds(int dtype, int observer_order, const wbc_solver_options* opt, int flags, size_t* out, int cap, int* n) {
  const int with_mats = flags & WBC_PLAN_WITH_MATS;
  const bool warm = (flags & WBC_PLAN_WARM) != 0;
  if (!n || (cap > 0 && !out) || (dtype != WBC_F64 && dtype != WBC_F32) || observer_order < 0 || observer_order > 2) return fail(WBC_E_INVALID, "bad argument");
  wbc_solver_options o;
  const int rc = options_from_caller(opt, o);
  if (rc) return rc;
  const Resolved r = resolve_options(dtype, o);
  // candidates: every constant the planner compares N with (+ 1 where the comparison is <=); kept when the plan really changes there
  const size_t cand[] = {r.pair_min, r.pair_max ? r.pair_max + 1 : (size_t)-1, (size_t)WBC_TT_WARM_MAX_F32, r.tt_min, r.tt_max + 1, r.tt_max_obs + 1, r.tt_max_noobs32 + 1, r.stile_min, r.stile_max + 1, r.fused_max_noobs + 1, r.fused_max_obs + 1, r.tile_min, r.obs_split_min, r.lane_min, r.warm_tile_min, r.warm_lane_min, r.obs_split_min_nomats, (size_t)WBC_PACK2_MIN_STATES, (size_t)WBC_F32_DENSE_TILE_MIN,
                         wbc::BIG_GRID_THREADS / 4, wbc::BIG_GRID_THREADS / 2, (size_t)65537, r.colaunch_min, r.colaunch_max + 1};
  size_t keep[24]; int k = 0;
  for (size_t c : cand) {
    if (c < 2 || c == (size_t)-1 || c > ((size_t)1 << 21)) continue;
    // an odd N never packs: compare like with like (both even) for the fp32 sweep, plain neighbours otherwise
    auto same = [&](size_t a, size_t b) {
      const TickPlan x = plan_tick(dtype, observer_order, o, r, a, with_mats != 0, true, warm), y = plan_tick(dtype, observer_order, o, r, b, with_mats != 0, true, warm);
      return x.fused == y.fused && x.front == y.front && x.qp == y.qp && x.qp_body == y.qp_body && x.pack2 == y.pack2 && x.sweep_block == y.sweep_block && x.qp_warm == y.qp_warm;
    };
    const bool changes = (c % 2 == 0) ? !same(c - 2, c) : !same(c - 1, c + 1);
    if (!changes) continue;
    bool dup = false;
    for (int i = 0; i < k; ++i) dup = dup || keep[i] == c;
    if (!dup && k < 24) keep[k++] = c;
  }
  for (int i = 0; i < k; ++i) for (int j = i + 1; j < k; ++j) if (keep[j] < keep[i]) { const size_t t = keep[i]; keep[i] = keep[j]; keep[j] = t; }
  // (an even candidate right behind a kept odd one is the same switch seen from the even side -- the plans of fp32 batches are compared even with even)
  for (int i = 1; i < k; ++i) if (keep[i] % 2 == 0 && keep[i - 1] + 1 == keep[i]) { for (int j = i; j + 1 < k; ++j) keep[j] = keep[j + 1]; --k; --i; }
  *n = k;
  for (int i = 0; i < k && i < cap; ++i) out[i] = keep[i];
  return WBC_OK;
}

extern "C" int wbc_solver_create_ex(const wbc_model* m, const wbc_params* p, int dtype, int device, size_t max_batch,
                                    const wbc_solver_options* opt, wbc_solver** out) {
  if (!m || !out || max_batch == 0 || (dtype != WBC_F64 && dtype != WBC_F32)) return fail(WBC_E_INVALID, "bad argument");
  *out = nullptr;
  if (max_batch > ((size_t)1 << 21))  // keeps every component-major array below 4 GiB (32-bit lane offsets)
    return fail(WBC_E_CAPACITY, "max_batch above 2^21 states per solver: shard the batch over more solvers");
  int rc = check_params(p);
  if (rc) return rc;
  wbc_solver_options o;
  rc = options_from_caller(opt, o);
  if (rc) return rc;
  if (o.rollout_spw != 0 && o.rollout_spw != 4 && o.rollout_spw != 16) return fail(WBC_E_INVALID, "rollout_spw must be 0 (auto), 4 or 16");
  {   // the tile sizes the kernels of this scalar type exist for (k_qp.hip)
    const int ok64[] = {0, -1, 32, 36, 40, 44, 48, 52, 56, 60, 64, 128, 256, 512};
    const int ok32[] = {0, -1, 32, 36, 40, 44, 48, 52, 56, 60, 64, 72, 80, 88, 96, 104, 112, 120, 128, 256, 512};
    bool found = false;
    if (dtype == WBC_F64) { for (int v : ok64) found = found || o.qp_tile == v; }
    else { for (int v : ok32) found = found || o.qp_tile == v; found = found || (o.qp_tile > 0 && o.qp_tile <= wbc::STILE_MAX_TILE && o.qp_tile % 4 == 0); }
    if (!found) return fail(WBC_E_INVALID, "qp_tile: 0 (auto), -1 (off), 32, 64, 128, 256, 512; fp64 also 36 ... 60 in steps of 4, fp32 also every multiple of 4 up to 192");
  }
  if (o.timing_mode != WBC_TIMING_DISPATCH && o.timing_mode != WBC_TIMING_EVENT_PAIR) return fail(WBC_E_INVALID, "bad timing_mode");
  if (o.f32_pack2 < -1 || o.f32_pack2 > 1) return fail(WBC_E_INVALID, "f32_pack2 must be -1, 0 or 1");
  if (o.keep_structural != 0 && o.keep_structural != 1) return fail(WBC_E_INVALID, "keep_structural must be 0 or 1");
  if (o.one_zerocopy < 0 || o.one_zerocopy > 3) return fail(WBC_E_INVALID, "one_zerocopy must be 0 ... 3");
  if (o.rollout_warm != 0 && o.rollout_warm != 1) return fail(WBC_E_INVALID, "rollout_warm must be 0 or 1");
  if (o.obs_colaunch < -1 || o.obs_colaunch > 1) return fail(WBC_E_INVALID, "obs_colaunch must be -1, 0 or 1");
  if (o.tile_tick < -1 || o.tile_tick > 1) return fail(WBC_E_INVALID, "tile_tick must be -1, 0 or 1");
  int leg_body[4][3];
  std::string err;
  rc = quadruped_topology(m->fm, leg_body, err);
  if (rc) return fail(rc, err);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(WBC_E_NODEVICE, "no HIP device: the WBC hot path has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(WBC_E_INVALID, "device index out of range");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(WBC_E_NODEVICE, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
  wbc_solver* s = new (std::nothrow) wbc_solver;
  if (!s) return fail(WBC_E_INVALID, "out of memory");
  s->dtype = dtype; s->device = device; s->max_batch = max_batch; s->params = *p; s->opt = o;
  s->rz = resolve_options(dtype, o);
  std::memcpy(s->leg_body, leg_body, sizeof(leg_body));
  for (int l = 0; l < 4; ++l) for (int k = 0; k < 3; ++k) s->jmap.j[3 * l + k] = leg_body[l][k] - 1;
  s->jpack = pack_jidx(s->jmap.j);
  const size_t ts = dtype == WBC_F64 ? 8 : 4;
  DeviceGuard guard(device);
  hipError_t e = guard.err;
  if (e == hipSuccess) {
    if (dtype == WBC_F64) {
      DevModel<double> dm; build_dev_model(m->fm, leg_body, dm);
      e = hipMalloc(&s->d_model, sizeof(dm));
      if (e == hipSuccess) e = hipMemcpy(s->d_model, &dm, sizeof(dm), hipMemcpyHostToDevice);
    } else {
      DevModel<float> dm; build_dev_model(m->fm, leg_body, dm);
      e = hipMalloc(&s->d_model, sizeof(dm));
      if (e == hipSuccess) e = hipMemcpy(s->d_model, &dm, sizeof(dm), hipMemcpyHostToDevice);
    }
  }
  if (e == hipSuccess) e = hipMalloc(&s->d_ws, (size_t)WS_LDS_WORDS * max_batch * ts);
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_todo, (max_batch + 4) * sizeof(int));
  if (e == hipSuccess) e = hipMemset(s->d_todo, 0, 4 * sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_aset, max_batch * sizeof(int));
  if (e == hipSuccess) e = hipMemset(s->d_aset, 0, max_batch * sizeof(int));
  // N = 1 image: q19 v18 w6 a18 n12 mu4 tp12 fp12 integ18 r18 tau12 f12 (scalars; 161 of ONE_TICK_SCALARS), a scratch region for
  // the start-up / planner helpers (wbc_observer_init, wbc_compute_reference: they must not touch what a caller keeps in the
  // image between ticks, wbc_one_map), then the ints mask | status | iters | completion ticket
  s->one_bytes = ONE_SCALARS * sizeof(double) + 4 * sizeof(int);
  if (e == hipSuccess) e = hipMalloc(&s->d_one, s->one_bytes);
  if (e == hipSuccess) e = hipHostMalloc(&s->h_one, s->one_bytes, hipHostMallocMapped);
  if (e == hipSuccess) e = hipHostGetDevicePointer(&s->h_one_dev, s->h_one, 0);
  // (the completion ticket lives in this image: a recycled pinned block must not look like ticket 1 of a solver whose first tick is still running)
  if (e == hipSuccess) { std::memset(s->h_one, 0, s->one_bytes); e = hipMemset(s->d_one, 0, s->one_bytes); }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->aux, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming);
  // kernels with more than 64 kB of dynamic LDS (staged QP tiles, tile tick): the limit is raised here, not inside a tick (which may be under stream capture)
  if (e == hipSuccess) e = dtype == WBC_F64 ? k_qp_prepare<double>() : k_qp_prepare<float>();
  if (e == hipSuccess) e = dtype == WBC_F64 ? k_tile_prepare<double>() : k_tile_prepare<float>();
  if (e != hipSuccess) {
    std::string msg = std::string("device allocation failed: ") + hipGetErrorString(e);
    wbc_solver_destroy(s);
    return fail(WBC_E_HIP, msg);
  }
  *out = s;
  return WBC_OK;
}

extern "C" int wbc_solver_create(const wbc_model* m, const wbc_params* p, int dtype, int device, size_t max_batch,
                                 wbc_solver** out) {
  return wbc_solver_create_ex(m, p, dtype, device, max_batch, nullptr, out);
}

extern "C" void wbc_solver_destroy(wbc_solver* s) {
  if (!s) return;
  DeviceGuard guard(s->device);
  if (s->d_model) (void)hipFree(s->d_model);
  if (s->d_ws) (void)hipFree(s->d_ws);
  if (s->d_todo) (void)hipFree(s->d_todo);
  if (s->d_aset) (void)hipFree(s->d_aset);
  if (s->d_one) (void)hipFree(s->d_one);
  if (s->h_one) (void)hipHostFree(s->h_one);
  if (s->d_ref) (void)hipFree(s->d_ref);
  if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
  if (s->ev_join) (void)hipEventDestroy(s->ev_join);
  if (s->aux) (void)hipStreamDestroy(s->aux);
  for (hipEvent_t ev : s->ev_pool) (void)hipEventDestroy(ev);
  delete s;
}

extern "C" int wbc_solver_set_params(wbc_solver* s, const wbc_params* p) {
  if (!s) return fail(WBC_E_INVALID, "null solver");
  int rc = check_params(p);
  if (rc) return rc;
  s->params = *p;
  return WBC_OK;
}

extern "C" int wbc_solver_device(const wbc_solver* s) { return s ? s->device : -1; }

extern "C" int wbc_solver_plan_tick(const wbc_solver* s, size_t N, int with_mats, int with_pf, int warm, wbc_tick_plan* plan) {
  if (!s || !plan) return fail(WBC_E_INVALID, "null argument");
  plan_to_public(plan_tick(s->dtype, s->params.observer_order, s->opt, s->rz, N, with_mats != 0, with_pf != 0, warm != 0), plan);
  return WBC_OK;
}

// keep_structural: forget which M / Jc buffers hold their structural constants (the caller freed, reallocated or overwrote them)
extern "C" int wbc_solver_invalidate_structural(wbc_solver* s) {
  if (!s) return fail(WBC_E_INVALID, "null solver");
  s->kept_M = nullptr; s->kept_Jc = nullptr; s->kept_N = 0;
  return WBC_OK;
}

// diagnostics: how many states of the LAST two-kernel tick the per-lane QP kernel handed to the dense kernel (synchronises)
extern "C" int wbc_solver_qp_handover(wbc_solver* s, int* count) {
  if (!s || !count) return fail(WBC_E_INVALID, "null argument");
  ON_DEVICE(s);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(count, s->d_todo + 2, sizeof(int), hipMemcpyDeviceToHost));   // [2]: what qp_list_kernel saw before it emptied the list
  return WBC_OK;
}

// ---- timing: spans borrow event pairs from a ring that wbc_solver_enable_timing allocated; nothing is created inside a tick
struct SpanScope {   // one instrumented kernel launch
  wbc_solver* s;
  LaunchCtx L;
  bool active = false;
  hipError_t err = hipSuccess;
  SpanScope(wbc_solver* s_, int kind, hipStream_t st) : s(s_) {
    L.st = st;
    L.f32_pack2 = s->opt.f32_pack2;
    if (!s->timing || !s->sample_now) return;
    if (s->spans.size() >= TIMING_MAX_SPANS) { ++s->dropped; return; }
    const size_t i = 2 * s->spans.size();
    wbc_solver::Span sp{kind, s->ev_pool[i], s->ev_pool[i + 1]};
    if (s->opt.timing_mode == WBC_TIMING_DISPATCH) { L.ev_start = sp.a; L.ev_stop = sp.b; }   // the dispatch's own timestamps
    else err = hipEventRecord(sp.a, st);
    s->spans.push_back(sp);
    active = true;
  }
  hipError_t end() {   // after the launch
    if (active && s->opt.timing_mode != WBC_TIMING_DISPATCH) return hipEventRecord(s->spans.back().b, L.st);
    return hipSuccess;
  }
  void cancel() { if (active) { s->spans.pop_back(); active = false; } }   // the launch failed
};
static void timing_tick(wbc_solver* s) {  // once per API call: is this tick instrumented?
  s->sample_now = s->timing && (s->calls++ % (unsigned long long)s->timing_period) == 0;
}

extern "C" int wbc_solver_enable_timing(wbc_solver* s, int on) {
  if (!s) return fail(WBC_E_INVALID, "null solver");
  if (on && s->ev_pool.size() != 2 * TIMING_MAX_SPANS) {
    ON_DEVICE(s);
    // built in a local vector and swapped in only when complete: a pool left short by a failed hipEventCreate would be
    // indexed past its end by the next SpanScope
    std::vector<hipEvent_t> pool;
    pool.reserve(2 * TIMING_MAX_SPANS);
    for (size_t i = 0; i < 2 * TIMING_MAX_SPANS; ++i) {
      hipEvent_t ev;
      const hipError_t e = hipEventCreate(&ev);
      if (e != hipSuccess) {
        for (hipEvent_t x : pool) (void)hipEventDestroy(x);
        return fail(WBC_E_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
      }
      pool.push_back(ev);
    }
    for (hipEvent_t x : s->ev_pool) (void)hipEventDestroy(x);
    s->ev_pool.swap(pool);
    s->spans.reserve(TIMING_MAX_SPANS);
  }
  s->timing = on != 0;
  s->timing_period = on > 1 ? on : 1;  // on = k > 1: sample every k-th tick (keeps the event cost out of the rest)
  s->calls = 0;
  s->sample_now = false;
  s->spans.clear();
  s->dropped = 0;
  return WBC_OK;
}

// cap = entries of the caller's arrays: kinds beyond it are dropped, never written (the unsized call wrote as many entries as THIS
// build knows, whatever the caller was built against)
extern "C" int wbc_solver_collect_timing_n(wbc_solver* s, double* ms, int* launches, int cap) {
  if (!s || !ms || !launches || cap < 0) return fail(WBC_E_INVALID, "bad argument");
  const int n = cap < WBC_TIMING_KINDS ? cap : WBC_TIMING_KINDS;
  for (int k = 0; k < n; ++k) { ms[k] = 0; launches[k] = 0; }
  ON_DEVICE(s);
  for (auto& sp : s->spans) {
    HIP_TRY(hipEventSynchronize(sp.b));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, sp.a, sp.b));
    if (sp.kind < n) { ms[sp.kind] += t; launches[sp.kind]++; }
  }
  s->spans.clear();
  s->dropped = 0;
  return WBC_OK;
}
// the unsized call of ABI <= 6 writes the FIVE entries every version of it has had (kind 5, the persistent rollout kernel, needs the sized call)
extern "C" int wbc_solver_collect_timing(wbc_solver* s, double* ms, int* launches) { return wbc_solver_collect_timing_n(s, ms, launches, 5); }

// one instrumented launch: kind = index into the timing arrays (0 dyn_sweep, 1 QP (dense active set), 2 rnea_step / observer, 3 fused tick, 4 QP one state per lane, 5 persistent rollout)
#define TIMED_LAUNCH(kind_, stream_, what_, call_)                                                          \
  do {                                                                                                      \
    SpanScope sc_(s, kind_, stream_);                                                                       \
    if (sc_.err != hipSuccess) return fail(WBC_E_HIP, std::string("hipEventRecord: ") + hipGetErrorString(sc_.err)); \
    const LaunchCtx& L = sc_.L;                                                                             \
    hipError_t le_ = (call_);                                                                               \
    if (le_ != hipSuccess) { sc_.cancel(); return fail(WBC_E_HIP, std::string(what_ " launch: ") + hipGetErrorString(le_)); } \
    le_ = sc_.end();                                                                                        \
    if (le_ != hipSuccess) return fail(WBC_E_HIP, std::string("hipEventRecord: ") + hipGetErrorString(le_)); \
  } while (0)

template <class T> static const DevModel<T>* dev_model(const wbc_solver* s) { return (const DevModel<T>*)s->d_model; }

// keep_structural: `skip` = this call may leave the structural zeros / ones of M and Jc alone (same buffers and N as the call that
// last WROTE them).  The buffers are recorded only once the launch that writes them has been enqueued (written()); a call that
// fails before that forgets them, so the next call writes every word again.
struct KeepScope {
  wbc_solver* s; const void* M; const void* Jc; size_t N; int skip = 0; bool done = false;
  KeepScope(wbc_solver* s_, const void* M_, const void* Jc_, size_t N_) : s(s_), M(M_), Jc(Jc_), N(N_) {
    // (in_rollout: ticks 1 .. horizon-1 of ONE wbc_rollout_batch call write the buffers tick 0 of the same call wrote)
    if (M) skip = ((s->opt.keep_structural || s->in_rollout) && M == s->kept_M && Jc == s->kept_Jc && N == s->kept_N) ? 1 : 0;
  }
  void written() { if (M) { s->kept_M = M; s->kept_Jc = Jc; s->kept_N = N; } done = true; }
  ~KeepScope() { if (M && !done) s->kept_M = nullptr; }
};

template <class T>
static int dynamics_impl(wbc_solver* s, size_t N, const void* q, const void* v, void* M, void* h, void* Jc, void* pf,
                         void* p, void* beta, hipStream_t st) {
  SweepArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.jpack = s->jpack;
  a.N = N; a.q = (const T*)q; a.v = (const T*)v;
  a.M = (T*)M; a.h = (T*)h; a.Jc = (T*)Jc; a.pf = (T*)pf; a.p = (T*)p; a.beta = (T*)beta;
  const int mode = (M ? SW_MATS : 0) | ((p || beta) ? SW_OBS : 0);
  KeepScope keep(s, M, Jc, N);
  a.skip_consts = keep.skip;
  timing_tick(s);
  TIMED_LAUNCH(0, st, "dyn_sweep", k_dyn_sweep<T>(L, mode, dev_model<T>(s), to_dev_params<T>(s->params), a));
  keep.written();
  return WBC_OK;
}

extern "C" int wbc_dynamics_batch(wbc_solver* s, size_t N, const void* q, const void* v, void* M, void* h, void* Jc,
                                  void* pf, void* p, void* beta, void* stream) {
  if (!s || !q || !v) return fail(WBC_E_INVALID, "null argument");
  if (N == 0) return WBC_OK;
  if (N > s->max_batch) return fail(WBC_E_CAPACITY, "N exceeds the solver's max_batch");   // (also keeps the 32-bit lane offsets in range)
  if ((M || h || Jc) && !(M && h && Jc)) return fail(WBC_E_INVALID, "M, h, Jc must be given together");
  if (!M && !pf && !p && !beta) return fail(WBC_E_INVALID, "no output requested");
  ON_DEVICE(s);
  hipStream_t st = (hipStream_t)stream;
  return s->dtype == WBC_F64 ? dynamics_impl<double>(s, N, q, v, M, h, Jc, pf, p, beta, st)
                             : dynamics_impl<float>(s, N, q, v, M, h, Jc, pf, p, beta, st);
}

template <class T>
static int step_impl(wbc_solver* s, size_t N, const wbc_batch_in* in, const wbc_batch_out* out,
                     const wbc_observer_state* obs, hipStream_t st, bool warm_api = false, const int* aset_in = nullptr, int* aset_out = nullptr) {
  SweepArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.jpack = s->jpack;
  a.N = N; a.q = (const T*)in->q; a.v = (const T*)in->v;
  a.M = (T*)out->M; a.h = (T*)out->h; a.Jc = (T*)out->Jc; a.pf = (T*)out->pf;
  a.w_des = (const T*)in->w_des; a.vdot_des = (const T*)in->vdot_des;
  a.tau_prev = (const T*)in->tau_prev; a.f_prev = (const T*)in->f_prev;
  a.obs_integ = obs ? (T*)obs->integ : nullptr; a.obs_r = obs ? (T*)obs->r : nullptr;
  a.ws = (T*)s->d_ws;
  const bool mats = out->M != nullptr, ob = s->params.observer_order > 0;
  KeepScope keep(s, out->M, out->Jc, N);
  a.skip_consts = keep.skip;
  timing_tick(s);
  QpArgs<T> qa;
  qa.jpack = s->jpack;
  qa.N = N; qa.ws = (const T*)s->d_ws; qa.normals = (const T*)in->normals; qa.mu = (const T*)in->mu; qa.mask = in->mask;
  qa.tau = (T*)out->tau; qa.f = (T*)out->f; qa.status = out->status; qa.iters = out->iters;
  qa.aset_in = aset_in; qa.aset_out = aset_out;
  qa.rprev = (ob && obs) ? (const T*)obs->r : nullptr;
  // two-kernel tick with M/h/Jc outputs: the QP takes its geometry from Jc and the sweep skips those workspace words
  qa.Jc = mats ? (const T*)out->Jc : nullptr;
  qa.wdes = nullptr;
  a.ws_geom = mats ? 0 : 1;
  const DevParams<T> dp = to_dev_params<T>(s->params);
  // (what runs, and why: plan_tick.  A warm call WITHOUT a set to start from -- tick 0 of a per-tick-launch rollout, the first tick of a closed loop --
  //  is planned as the cold tick it is: between the warm and the cold per-lane thresholds the cold tiles are the faster kernels; they still report the set)
  //  -- unless the cold plan's QP kernel is the fp32 12 x 12 body, which reports no set: then the warm plan's kernels run, started from the empty set.)
  TickPlan pl = plan_tick(s->dtype, s->params.observer_order, s->opt, s->rz, N, mats, out->pf != nullptr, warm_api);
  if (warm_api && aset_in == nullptr) {
    const TickPlan pc = plan_tick(s->dtype, s->params.observer_order, s->opt, s->rz, N, mats, out->pf != nullptr, false);
    if (pc.qp_body == 0) pl = pc;
  }
  const bool warm = warm_api && aset_in != nullptr && pl.qp_warm;
  if (pl.fused == 1) {
    TIMED_LAUNCH(3, st, "fused tick", k_fused_tick<T>(L, ob, mats, dev_model<T>(s), dp, a, qa, s->jmap, warm));
    keep.written();
    return WBC_OK;
  }
  if (pl.fused == 3) {
    TIMED_LAUNCH(3, st, "fused pair tick", k_fused_pair<T>(L, dev_model<T>(s), dp, a, qa, s->jmap));
    keep.written();
    return WBC_OK;
  }
  if (pl.fused == 2) {   // (the roles leave w_des to the QP stage: SW_NOB)
    qa.wdes = (const T*)in->w_des;
    TIMED_LAUNCH(3, st, "tile tick", k_tile_tick<T>(L, ob, pl.tile, dev_model<T>(s), dp, a, qa, s->jmap));
    keep.written();
    return WBC_OK;
  }
  // front halves that do not change the target wrench leave it to the QP kernels to read the caller's w_des (QpArgs::wdes)
  const bool front_writes_b = ob && !pl.obs_split;   // the all-in-one observer forms: b = w_des - rhat_base
  qa.wdes = front_writes_b ? nullptr : (const T*)in->w_des;   // (those front halves run their SW_NOB / RS_NOB variants)
  a.qp_todo = pl.lane ? s->d_todo : nullptr;   // the front-half kernel empties the hand-over list (one thread; a kernel of its own took 4.7 us per tick)
  if (pl.front == 3) {   // no M, h, Jc wanted, large observer-on batch: observer kernel beside the observer-free rnea_step (as front == 2 below)
    const int mode = RS_STEP | RS_NOB | (out->pf ? RS_PF : 0);
    if (s->opt.obs_split_serial) {
      TIMED_LAUNCH(2, st, "observer", k_observer<T>(L, dev_model<T>(s), dp, a));
      TIMED_LAUNCH(0, st, "rnea_step", k_rnea_step<T>(L, mode, dev_model<T>(s), dp, a));
    } else {
      HIP_TRY(hipEventRecord(s->ev_fork, st));
      HIP_TRY(hipStreamWaitEvent(s->aux, s->ev_fork, 0));
      TIMED_LAUNCH(2, s->aux, "observer", k_observer<T>(L, dev_model<T>(s), dp, a));
      HIP_TRY(hipEventRecord(s->ev_join, s->aux));
      TIMED_LAUNCH(0, st, "rnea_step", k_rnea_step<T>(L, mode, dev_model<T>(s), dp, a));
      HIP_TRY(hipStreamWaitEvent(st, s->ev_join, 0));   // the QP needs rhat
    }
  } else if (pl.front == 1) {  // no M, h, Jc wanted: the CRBA-free rnea_step kernel is the whole front half
    const int mode = RS_STEP | (ob ? RS_OBS : RS_NOB) | (out->pf ? RS_PF : 0);
    TIMED_LAUNCH(2, st, "rnea_step", k_rnea_step<T>(L, mode, dev_model<T>(s), dp, a));
  } else if (pl.front == 4) {   // mid-size observer-on batch: the observer update and the observer-free sweep as the two roles of one launch
    TIMED_LAUNCH(0, st, "sweep_obs", k_sweep_obs<T>(L, dev_model<T>(s), dp, a));
  } else if (pl.front == 2) {
    // large observer-on batch: the observer update runs as its own light kernel in front of (option: beside, on the second
    // stream) a dyn_sweep WITHOUT the observer passes (252 instead of 370 VGPRs: two waves per SIMD, shared tables), which writes
    // M, h, Jc; rhat travels through 18 extra workspace words and the QP kernel completes b and tau_partial with it
    if (s->opt.obs_split_serial) {   // same stream, one after the other
      TIMED_LAUNCH(2, st, "observer", k_observer<T>(L, dev_model<T>(s), dp, a));
      TIMED_LAUNCH(0, st, "dyn_sweep", k_dyn_sweep<T>(L, SW_MATS | SW_STEP | SW_NOB, dev_model<T>(s), dp, a));
    } else {
      HIP_TRY(hipEventRecord(s->ev_fork, st));
      HIP_TRY(hipStreamWaitEvent(s->aux, s->ev_fork, 0));
      TIMED_LAUNCH(2, s->aux, "observer", k_observer<T>(L, dev_model<T>(s), dp, a));
      HIP_TRY(hipEventRecord(s->ev_join, s->aux));
      TIMED_LAUNCH(0, st, "dyn_sweep", k_dyn_sweep<T>(L, SW_MATS | SW_STEP | SW_NOB, dev_model<T>(s), dp, a));
      HIP_TRY(hipStreamWaitEvent(st, s->ev_join, 0));   // the QP needs rhat
    }
  } else {
    TIMED_LAUNCH(0, st, "dyn_sweep", k_dyn_sweep<T>(L, SW_MATS | SW_STEP | (ob ? SW_OBS : SW_NOB), dev_model<T>(s), dp, a));
  }
  keep.written();
  if (pl.lane) {
    TIMED_LAUNCH(4, st, "qp_lane", k_qp_lane<T>(L, pl.obs_split, dp, qa, s->jmap, s->d_todo, warm));
    TIMED_LAUNCH(1, st, "qp", k_qp<T>(L, pl.obs_split, 0, dp, qa, s->jmap, s->d_todo, warm));
    return WBC_OK;
  }
  TIMED_LAUNCH(1, st, "qp", k_qp<T>(L, pl.obs_split, pl.tile, dp, qa, s->jmap, nullptr, warm, pl.qp_body));
  return WBC_OK;
}

// argument checks of one tick (no HIP call): shared with wbc_multi_*, which validates every shard before it enqueues any
int wbc::check_step_args(const wbc_solver* s, size_t N, const wbc_batch_in* in, const wbc_batch_out* out,
                         const wbc_observer_state* obs, bool rollout) {
  if (!s || !in || !out) return fail(WBC_E_INVALID, "null argument");
  if (N == 0) return WBC_OK;
  if (N > s->max_batch) return fail(WBC_E_CAPACITY, "N exceeds the solver's max_batch");
  if (!in->q || !in->v || !in->w_des || !in->vdot_des || !in->normals || !in->mu || !in->mask)
    return fail(WBC_E_INVALID, "null input buffer");
  if (!out->tau || !out->f || !out->status) return fail(WBC_E_INVALID, "null output buffer");
  if ((out->M || out->h || out->Jc) && !(out->M && out->h && out->Jc))
    return fail(WBC_E_INVALID, "M, h, Jc must be given together");
  if (rollout && !out->M) return fail(WBC_E_INVALID, "rollouts need the M, h, Jc buffers (forward dynamics reads them)");
  if (s->params.observer_order > 0) {
    if (!obs || !obs->integ || !obs->r) return fail(WBC_E_INVALID, "observer on: observer state buffers required");
    if (!rollout && (!in->tau_prev || !in->f_prev)) return fail(WBC_E_INVALID, "observer on: tau_prev and f_prev required");
  }
  return WBC_OK;
}

extern "C" int wbc_step_batch(wbc_solver* s, size_t N, const wbc_batch_in* in, const wbc_batch_out* out,
                              const wbc_observer_state* obs, void* stream) {
  const int rc0 = check_step_args(s, N, in, out, obs, false);
  if (rc0) return rc0;
  if (N == 0) return WBC_OK;
  ON_DEVICE(s);
  hipStream_t st = (hipStream_t)stream;
  return s->dtype == WBC_F64 ? step_impl<double>(s, N, in, out, obs, st) : step_impl<float>(s, N, in, out, obs, st);
}

// One tick of a DEPENDENT sequence (a closed loop, a rollout driven by the caller): the GRF QP of every state starts from active_in
extern "C" int wbc_step_batch_warm(wbc_solver* s, size_t N, const wbc_batch_in* in, const wbc_batch_out* out,
                                   const wbc_observer_state* obs, const int* active_in, int* active_out, void* stream) {
  const int rc0 = check_step_args(s, N, in, out, obs, false);
  if (rc0) return rc0;
  if (N == 0) return WBC_OK;
  ON_DEVICE(s);
  hipStream_t st = (hipStream_t)stream;
  return s->dtype == WBC_F64 ? step_impl<double>(s, N, in, out, obs, st, true, active_in, active_out)
                             : step_impl<float>(s, N, in, out, obs, st, true, active_in, active_out);
}

template <class T>
static int integrate_impl(wbc_solver* s, size_t N, void* q, void* v, const void* M, const void* h, const void* Jc,
                          const void* tau, const void* f, const void* tau_ext, void* tau_traj, hipStream_t st) {
  IntegrateArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.N = N; a.q = (T*)q; a.v = (T*)v; a.M = (const T*)M; a.h = (const T*)h; a.Jc = (const T*)Jc;
  a.tau = (const T*)tau; a.f = (const T*)f; a.tau_ext = (const T*)tau_ext; a.tau_traj = (T*)tau_traj;
  a.dt = (T)s->params.dt;
  a.jpack = s->jpack;
  LaunchCtx L; L.st = st;
  hipError_t e = k_integrate<T>(L, dev_model<T>(s), a);
  if (e != hipSuccess) return fail(WBC_E_HIP, std::string("integrate launch: ") + hipGetErrorString(e));
  return WBC_OK;
}

extern "C" int wbc_integrate_batch(wbc_solver* s, size_t N, void* q, void* v, const void* M, const void* h,
                                   const void* Jc, const void* tau, const void* f, const void* tau_ext, void* stream) {
  if (!s || !q || !v || !M || !h || !Jc || !tau || !f) return fail(WBC_E_INVALID, "null argument");
  if (N == 0) return WBC_OK;
  if (N > s->max_batch) return fail(WBC_E_CAPACITY, "N exceeds the solver's max_batch");
  ON_DEVICE(s);
  hipStream_t st = (hipStream_t)stream;
  return s->dtype == WBC_F64 ? integrate_impl<double>(s, N, q, v, M, h, Jc, tau, f, tau_ext, nullptr, st)
                             : integrate_impl<float>(s, N, q, v, M, h, Jc, tau, f, tau_ext, nullptr, st);
}

// small batches: the whole horizon in ONE launch (rollout_kernel, fused_tick.hip.hpp)
template <class T>
static int rollout_persistent(wbc_solver* s, size_t N, int horizon, const wbc_batch_in* in, const wbc_batch_out* out,
                              const wbc_observer_state* obs, const void* tau_ext, void* tau_traj, hipStream_t st,
                              const void* plan = nullptr, void* com_traj = nullptr) {
  SweepArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.jpack = s->jpack;
  a.N = N; a.q = (const T*)in->q; a.v = (const T*)in->v;
  a.M = (T*)out->M; a.h = (T*)out->h; a.Jc = (T*)out->Jc; a.pf = (T*)out->pf;
  a.w_des = (const T*)in->w_des; a.vdot_des = (const T*)in->vdot_des;
  a.tau_prev = (const T*)out->tau; a.f_prev = (const T*)out->f;
  a.obs_integ = obs ? (T*)obs->integ : nullptr; a.obs_r = obs ? (T*)obs->r : nullptr;
  a.ws = (T*)s->d_ws;
  a.skip_consts = 0;   // (tick 0 writes M / Jc in full; the later ticks of the launch leave their structural zeros / ones alone: rollout_kernel)
  s->kept_M = nullptr;
  QpArgs<T> qa;
  qa.jpack = s->jpack;
  qa.N = N; qa.ws = (const T*)s->d_ws; qa.normals = (const T*)in->normals; qa.mu = (const T*)in->mu; qa.mask = in->mask;
  qa.tau = (T*)out->tau; qa.f = (T*)out->f; qa.status = out->status; qa.iters = out->iters; qa.Jc = nullptr; qa.wdes = nullptr;
  qa.aset_in = nullptr; qa.aset_out = nullptr;   // (every tick of the launch but the first starts from the previous tick's set, kept in registers)
  // (cold rollouts with the planner in the loop keep the QP waiting for rhat: the speculative start lost there, 26.8 -> 29.0 us per tick)
  qa.rprev = (s->params.observer_order > 0 && obs && !plan) ? (const T*)obs->r : nullptr;
  a.ws_geom = 1;
  IntegrateArgs<T> ia;
  std::memset(&ia, 0, sizeof(ia));
  ia.N = N; ia.q = (T*)in->q; ia.v = (T*)in->v; ia.M = (const T*)out->M; ia.h = (const T*)out->h; ia.Jc = (const T*)out->Jc;
  ia.tau = (const T*)out->tau; ia.f = (const T*)out->f; ia.tau_ext = (const T*)tau_ext; ia.tau_traj = (T*)tau_traj;
  ia.dt = (T)s->params.dt;
  ia.jpack = s->jpack;
  // states per workgroup: 4 while that still fits one workgroup per CU (a tick then waits for the slowest of 4 QPs, not 16)
  const int spw = (s->opt.rollout_spw == 4 || (s->opt.rollout_spw == 0 && N <= 1024)) ? 4 : 16;
  RefArgs<T> ra;
  std::memset(&ra, 0, sizeof(ra));
  ra.jpack = s->jpack;
  ra.N = N; ra.q = (const T*)in->q; ra.v = (const T*)in->v; ra.plan = (const T*)plan; ra.t = (T)0;
  ra.w_des = (T*)in->w_des; ra.vdot_des = (T*)in->vdot_des; ra.com = (T*)com_traj;
  timing_tick(s);
  TIMED_LAUNCH(5, st, "rollout", k_rollout<T>(L, s->params.observer_order > 0, plan != nullptr, spw, dev_model<T>(s), to_dev_params<T>(s->params), a, qa,
                                              s->jmap, ia, horizon, (const DevRefParams<T>*)s->d_ref, ra, s->opt.rollout_warm != 0));
  return WBC_OK;
}

static bool rollout_as_one_launch(const wbc_solver* s, size_t N) { return N <= s->rz.fused_max && s->opt.rollout_persistent; }

extern "C" int wbc_rollout_batch(wbc_solver* s, size_t N, int horizon, const wbc_batch_in* in, const wbc_batch_out* out,
                                 const wbc_observer_state* obs, const void* tau_ext, void* tau_traj, void* stream) {
  if (!s || !in || !out) return fail(WBC_E_INVALID, "null argument");
  if (horizon < 1) return fail(WBC_E_INVALID, "horizon must be >= 1");
  if (!out->M || !out->h || !out->Jc) return fail(WBC_E_INVALID, "rollouts need the M, h, Jc buffers (forward dynamics reads them)");
  if (N == 0) return WBC_OK;   // empty shard
  if (rollout_as_one_launch(s, N)) {
    if (N > s->max_batch) return fail(WBC_E_CAPACITY, "N exceeds the solver's max_batch");
    if (!in->q || !in->v || !in->w_des || !in->vdot_des || !in->normals || !in->mu || !in->mask)
      return fail(WBC_E_INVALID, "null input buffer");
    if (!out->tau || !out->f || !out->status) return fail(WBC_E_INVALID, "null output buffer");
    if (s->params.observer_order > 0 && (!obs || !obs->integ || !obs->r))
      return fail(WBC_E_INVALID, "observer on: observer state buffers required");
    ON_DEVICE(s);
    hipStream_t st0 = (hipStream_t)stream;
    return s->dtype == WBC_F64 ? rollout_persistent<double>(s, N, horizon, in, out, obs, tau_ext, tau_traj, st0)
                               : rollout_persistent<float>(s, N, horizon, in, out, obs, tau_ext, tau_traj, st0);
  }
  wbc_batch_in tick = *in;
  tick.tau_prev = out->tau;  // the previous tick's outputs are this tick's tau_prev / f_prev: the sweep reads them
  tick.f_prev = out->f;      // before the QP kernel of the same tick overwrites them
  const size_t ts = s->dtype == WBC_F64 ? 8 : 4;
  const size_t nj = 12;
  ON_DEVICE(s);
  s->kept_M = nullptr;   // tick 0 writes M, Jc in full whatever an earlier call left there
  // rollout_warm with per-tick launches: where the planner's warm tick really starts from the sets (fused launch, warm one-wavefront kernel,
  // warm per-lane pair); in between the cold tiles are the faster kernels (plan_tick)
  const bool warm_ticks = s->opt.rollout_warm && plan_tick(s->dtype, s->params.observer_order, s->opt, s->rz, N, true, out->pf != nullptr, true).qp_warm;
  for (int t = 0; t < horizon; ++t) {
    s->in_rollout = t > 0;
    // tick t > 0 starts its QPs from the active sets tick t - 1 left in d_aset
    int rc = warm_ticks ? wbc_step_batch_warm(s, N, &tick, out, obs, t > 0 ? s->d_aset : nullptr, s->d_aset, stream)
                                 : wbc_step_batch(s, N, &tick, out, obs, stream);
    s->in_rollout = 0;
    if (rc) return rc;
    void* traj = tau_traj ? (void*)((char*)tau_traj + (size_t)t * nj * N * ts) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    rc = s->dtype == WBC_F64
             ? integrate_impl<double>(s, N, (void*)in->q, (void*)in->v, out->M, out->h, out->Jc, out->tau, out->f, tau_ext, traj, st)
             : integrate_impl<float>(s, N, (void*)in->q, (void*)in->v, out->M, out->h, out->Jc, out->tau, out->f, tau_ext, traj, st);
    if (rc) return rc;
  }
  return WBC_OK;
}

// ------------------------------------------------------------------------------------------ CoM reference generator
extern "C" void wbc_ref_params_default(wbc_ref_params* g) {
  if (!g) return;
  std::memset(g, 0, sizeof(*g));
  const double kp[3] = {100, 100, 150}, kd[3] = {20, 20, 25}, kr[3] = {200, 200, 100}, dr[3] = {25, 25, 15};
  const double in[3] = {0.8, 1.85, 2.05};  // composite inertia of the synthetic quadruped in its nominal stance
  for (int i = 0; i < 3; ++i) { g->kp_com[i] = kp[i]; g->kd_com[i] = kd[i]; g->kp_rot[i] = kr[i]; g->kd_rot[i] = dr[i]; g->inertia_nom[i] = in[i]; }
  g->kp_joint = 200; g->kd_joint = 28;
}

template <class T> static int upload_ref(wbc_solver* s, const wbc_ref_params* g) {
  DevRefParams<T> d;
  for (int i = 0; i < 3; ++i) {
    d.kp_com[i] = (T)g->kp_com[i]; d.kd_com[i] = (T)g->kd_com[i]; d.kp_rot[i] = (T)g->kp_rot[i]; d.kd_rot[i] = (T)g->kd_rot[i];
    d.inertia_nom[i] = (T)g->inertia_nom[i];
  }
  d.kp_joint = (T)g->kp_joint; d.kd_joint = (T)g->kd_joint;
  for (int i = 0; i < 12; ++i) d.q_nom[i] = (T)g->q_nom[i];
  if (!s->d_ref) HIP_TRY(hipMalloc(&s->d_ref, sizeof(DevRefParams<double>)));
  HIP_TRY(hipMemcpy(s->d_ref, &d, sizeof(d), hipMemcpyHostToDevice));
  return WBC_OK;
}

extern "C" int wbc_solver_set_ref_params(wbc_solver* s, const wbc_ref_params* g) {
  if (!s || !g) return fail(WBC_E_INVALID, "null argument");
  for (int i = 0; i < 3; ++i)
    if (!(g->inertia_nom[i] >= 0)) return fail(WBC_E_INVALID, "inertia_nom must be non-negative");
  ON_DEVICE(s);
  return s->dtype == WBC_F64 ? upload_ref<double>(s, g) : upload_ref<float>(s, g);
}

template <class T>
static int reference_impl(wbc_solver* s, size_t N, const void* q, const void* v, const void* plan, double t, void* w_des,
                          void* vdot_des, void* com, hipStream_t st) {
  RefArgs<T> a;
  std::memset(&a, 0, sizeof(a));   // (simg, refimg, planimg, skip_out: set by the rollout kernel only -- ADVICE r5)
  a.N = N; a.q = (const T*)q; a.v = (const T*)v; a.plan = (const T*)plan; a.t = (T)t;
  a.w_des = (T*)w_des; a.vdot_des = (T*)vdot_des; a.com = (T*)com;
  a.jpack = s->jpack;
  LaunchCtx L; L.st = st;
  hipError_t e = k_reference<T>(L, dev_model<T>(s), (const DevRefParams<T>*)s->d_ref, a);
  if (e != hipSuccess) return fail(WBC_E_HIP, std::string("reference launch: ") + hipGetErrorString(e));
  return WBC_OK;
}

extern "C" int wbc_reference_batch(wbc_solver* s, size_t N, const void* q, const void* v, const void* plan, double t,
                                   void* w_des, void* vdot_des, void* com, void* stream) {
  if (!s || !q || !v || !plan || !w_des || !vdot_des) return fail(WBC_E_INVALID, "null argument");
  if (!s->d_ref) return fail(WBC_E_INVALID, "call wbc_solver_set_ref_params first");
  if (N == 0) return WBC_OK;
  if (N > s->max_batch) return fail(WBC_E_CAPACITY, "N exceeds the solver's max_batch");
  ON_DEVICE(s);
  hipStream_t st = (hipStream_t)stream;
  return s->dtype == WBC_F64 ? reference_impl<double>(s, N, q, v, plan, t, w_des, vdot_des, com, st)
                             : reference_impl<float>(s, N, q, v, plan, t, w_des, vdot_des, com, st);
}

extern "C" int wbc_rollout_tracking_batch(wbc_solver* s, size_t N, int horizon, const wbc_batch_in* in,
                                          const wbc_batch_out* out, const wbc_observer_state* obs, const void* tau_ext,
                                          const void* plan, void* tau_traj, void* com_traj, void* stream) {
  if (!s || !in || !out || !plan) return fail(WBC_E_INVALID, "null argument");
  if (horizon < 1) return fail(WBC_E_INVALID, "horizon must be >= 1");
  if (!out->M || !out->h || !out->Jc) return fail(WBC_E_INVALID, "rollouts need the M, h, Jc buffers (forward dynamics reads them)");
  if (!in->q || !in->v || !in->w_des || !in->vdot_des) return fail(WBC_E_INVALID, "null input buffer");
  if (!s->d_ref) return fail(WBC_E_INVALID, "call wbc_solver_set_ref_params first");
  if (N == 0) return WBC_OK;   // empty shard
  if (rollout_as_one_launch(s, N)) {
    if (N > s->max_batch) return fail(WBC_E_CAPACITY, "N exceeds the solver's max_batch");
    if (!in->normals || !in->mu || !in->mask) return fail(WBC_E_INVALID, "null input buffer");
    if (!out->tau || !out->f || !out->status) return fail(WBC_E_INVALID, "null output buffer");
    if (s->params.observer_order > 0 && (!obs || !obs->integ || !obs->r))
      return fail(WBC_E_INVALID, "observer on: observer state buffers required");
    ON_DEVICE(s);
    hipStream_t st0 = (hipStream_t)stream;
    return s->dtype == WBC_F64 ? rollout_persistent<double>(s, N, horizon, in, out, obs, tau_ext, tau_traj, st0, plan, com_traj)
                               : rollout_persistent<float>(s, N, horizon, in, out, obs, tau_ext, tau_traj, st0, plan, com_traj);
  }
  wbc_batch_in tick = *in;
  tick.tau_prev = out->tau;
  tick.f_prev = out->f;
  const size_t ts = s->dtype == WBC_F64 ? 8 : 4;
  const size_t nj = 12;
  ON_DEVICE(s);
  const bool warm_ticks = s->opt.rollout_warm && plan_tick(s->dtype, s->params.observer_order, s->opt, s->rz, N, true, out->pf != nullptr, true).qp_warm;
  for (int t = 0; t < horizon; ++t) {
    void* com = com_traj ? (void*)((char*)com_traj + (size_t)t * 6 * N * ts) : nullptr;
    int rc = wbc_reference_batch(s, N, in->q, in->v, plan, (double)t * s->params.dt, (void*)in->w_des, (void*)in->vdot_des, com,
                                 stream);
    if (rc) return rc;
    if (t == 0) s->kept_M = nullptr;
    s->in_rollout = t > 0;
    rc = warm_ticks ? wbc_step_batch_warm(s, N, &tick, out, obs, t > 0 ? s->d_aset : nullptr, s->d_aset, stream)
                             : wbc_step_batch(s, N, &tick, out, obs, stream);
    s->in_rollout = 0;
    if (rc) return rc;
    void* traj = tau_traj ? (void*)((char*)tau_traj + (size_t)t * nj * N * ts) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    rc = s->dtype == WBC_F64
             ? integrate_impl<double>(s, N, (void*)in->q, (void*)in->v, out->M, out->h, out->Jc, out->tau, out->f, tau_ext, traj, st)
             : integrate_impl<float>(s, N, (void*)in->q, (void*)in->v, out->M, out->h, out->Jc, out->tau, out->f, tau_ext, traj, st);
    if (rc) return rc;
  }
  return WBC_OK;
}

// ---- single-robot tick on the solver's pinned image.  Layout (scalars of the solver's dtype, then ints):
static const int ONE_OFF[] = {0, 19, 37, 43, 61, 73, 77, 89, 101, 119, 137, 149, 161};  // q v w a n mu tp fp ig r tau f end
// ints behind the 200 scalars: [0] mask, [1] status, [2] iters, [3] completion ticket (one_zerocopy >= 2)

// Runs wbc_step_batch(N = 1) on the image and returns when tau, f, status (and the observer state) are in the HOST image.
//   one_zerocopy = 0: copy the image to the device scratch, tick, copy back, hipStreamSynchronize
//   one_zerocopy = 1: the kernels read / write the pinned image directly (mapped host memory), hipStreamSynchronize
//   one_zerocopy = 2: as 1, but completion is a 32-bit ticket that the STREAM writes into the image behind the tick
//                     (hipStreamWriteValue32) and the host polls in memory -- no runtime call on the wait path
//   one_zerocopy = 3: as 2 with a one-thread kernel writing the ticket (for stacks that refuse the stream write)
static int one_tick_on_image(wbc_solver* s) {
  const size_t ts = s->dtype == WBC_F64 ? 8 : 4;
  const int* off = ONE_OFF;
  unsigned char* hb = (unsigned char*)s->h_one;
  const int zcm = s->opt.one_zerocopy;
  const bool zc = zcm != 0;
  unsigned char* d = (unsigned char*)(zc ? s->h_one_dev : s->d_one);
  int* dints = (int*)(d + ONE_SCALARS * sizeof(double));
  int* hints = (int*)(hb + ONE_SCALARS * sizeof(double));
  if (!zc) HIP_TRY(hipMemcpyAsync(d, hb, s->one_bytes, hipMemcpyHostToDevice, nullptr));
  wbc_batch_in in;
  in.q = d + off[0] * ts; in.v = d + off[1] * ts; in.w_des = d + off[2] * ts; in.vdot_des = d + off[3] * ts;
  in.normals = d + off[4] * ts; in.mu = d + off[5] * ts; in.mask = dints;
  in.tau_prev = d + off[6] * ts; in.f_prev = d + off[7] * ts;
  wbc_batch_out out;
  std::memset(&out, 0, sizeof(out));
  out.tau = d + off[10] * ts; out.f = d + off[11] * ts; out.status = dints + 1; out.iters = dints + 2;
  wbc_observer_state os{d + off[8] * ts, d + off[9] * ts};
  int rc = wbc_step_batch(s, 1, &in, &out, &os, nullptr);
  if (rc) return rc;
  if (!zc) HIP_TRY(hipMemcpyAsync(hb, d, s->one_bytes, hipMemcpyDeviceToHost, nullptr));
  if (zcm >= 2 && s->one_flag_ok) {
    unsigned ticket = ++s->one_seq;
    if (ticket == 0) ticket = ++s->one_seq;                       // (0 is the "not yet" value below)
    __atomic_store_n((unsigned*)(hints + 3), 0u, __ATOMIC_RELEASE);   // before the launch: a match can only come from THIS tick's write
    hipError_t e = zcm == 2 ? hipStreamWriteValue32(nullptr, dints + 3, ticket, 0) : k_flag(nullptr, (unsigned*)(dints + 3), ticket);
    if (e == hipSuccess) {
      // stream order puts the ticket behind the tick's kernels, whose writes to the (fine-grained) image are released at
      // their end: ticket visible => outputs visible.  Bounded spin, then the runtime's wait (a stalled device must not hang us).
      const unsigned* flag = (const unsigned*)(hints + 3);
      const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(5);   // a healthy tick takes ~15 us
      for (;;) {
        for (int spin = 0; spin < 256; ++spin) {
          if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == ticket) return WBC_OK;
          cpu_relax();
        }
        if (std::chrono::steady_clock::now() > t_end) break;
      }
    } else {
      (void)hipGetLastError();
      s->one_flag_ok = false;   // refused on this stack: the plain wait from now on
    }
  }
  HIP_TRY(hipStreamSynchronize(nullptr));
  return WBC_OK;
}

extern "C" int wbc_compute_torques(wbc_solver* s, const double* q, const double* v, const double* w_des,
                                   const double* vdot_des, const double* normals, const double* mu, int mask,
                                   const double* tau_prev, const double* f_prev, double* obs_integ, double* obs_r,
                                   double* tau, double* f, int* status) {
  if (!s || !q || !v || !w_des || !vdot_des || !normals || !mu || !tau || !f || !status)
    return fail(WBC_E_INVALID, "null argument");
  const bool ob = s->params.observer_order > 0;
  if (ob && (!tau_prev || !f_prev || !obs_integ || !obs_r)) return fail(WBC_E_INVALID, "observer on: state required");
  ON_DEVICE(s);
  // host staging in the solver's dtype: a pinned image of the device scratch (doubles/floats, then mask | status | iters),
  // one asynchronous copy each way around the launch and one synchronisation (was four blocking copies from pageable memory)
  const int* off = ONE_OFF;
  unsigned char* hb = (unsigned char*)s->h_one;
  int* hints = (int*)(hb + ONE_SCALARS * sizeof(double));
  auto put = [&](int o, const double* src, int n) {
    for (int i = 0; i < n; ++i) {
      if (s->dtype == WBC_F64) ((double*)hb)[o + i] = src ? src[i] : 0.0;
      else ((float*)hb)[o + i] = src ? (float)src[i] : 0.0f;
    }
  };
  put(off[0], q, 19); put(off[1], v, 18); put(off[2], w_des, 6); put(off[3], vdot_des, 18); put(off[4], normals, 12);
  put(off[5], mu, 4); put(off[6], tau_prev, 12); put(off[7], f_prev, 12); put(off[8], obs_integ, 18); put(off[9], obs_r, 18);
  put(off[10], nullptr, 12); put(off[11], nullptr, 12);
  hints[0] = mask; hints[1] = 0; hints[2] = 0;
  int rc = one_tick_on_image(s);
  if (rc) return rc;
  auto get = [&](int o, double* dst, int n) {
    if (!dst) return;
    for (int i = 0; i < n; ++i)
      dst[i] = s->dtype == WBC_F64 ? ((double*)hb)[o + i] : (double)((float*)hb)[o + i];
  };
  get(off[10], tau, 12); get(off[11], f, 12);
  if (ob) { get(off[8], obs_integ, 18); get(off[9], obs_r, 18); }
  *status = hints[1];
  return WBC_OK;
}

// The single-robot loop WITHOUT staging copies: the caller keeps its state in the solver's pinned image and rewrites only
// what changed between ticks (fp64 solvers: the image's scalars are doubles).
extern "C" int wbc_one_map(wbc_solver* s, wbc_one_image* img) {
  if (!s || !img) return fail(WBC_E_INVALID, "null argument");
  if (s->dtype != WBC_F64) return fail(WBC_E_INVALID, "wbc_one_map: fp64 solvers only (the image holds the solver's scalar type)");
  double* hb = (double*)s->h_one;
  int* hints = (int*)((unsigned char*)s->h_one + ONE_SCALARS * sizeof(double));
  const int* off = ONE_OFF;
  img->q = hb + off[0]; img->v = hb + off[1]; img->w_des = hb + off[2]; img->vdot_des = hb + off[3]; img->normals = hb + off[4];
  img->mu = hb + off[5]; img->tau_prev = hb + off[6]; img->f_prev = hb + off[7]; img->obs_integ = hb + off[8]; img->obs_r = hb + off[9];
  img->tau = hb + off[10]; img->f = hb + off[11];
  img->mask = hints; img->status = hints + 1; img->iters = hints + 2;
  return WBC_OK;
}

extern "C" int wbc_one_tick(wbc_solver* s) {
  if (!s) return fail(WBC_E_INVALID, "null solver");
  if (s->dtype != WBC_F64) return fail(WBC_E_INVALID, "wbc_one_tick: fp64 solvers only");
  ON_DEVICE(s);
  return one_tick_on_image(s);
}

// Observer start-up for the single-robot host-pointer loop: integ(0) = p(0) = M(q) v, r(0) = 0 (see wbc_observer_state).
extern "C" int wbc_observer_init(wbc_solver* s, const double* q, const double* v, double* obs_integ, double* obs_r) {
  if (!s || !q || !v || !obs_integ || !obs_r) return fail(WBC_E_INVALID, "null argument");
  ON_DEVICE(s);
  const size_t ts = s->dtype == WBC_F64 ? 8 : 4;
  const int o = ONE_SCRATCH;   // scratch region of the pinned image (q at +0, v at +19, p = M v at +37): the tick's part stays as the caller left it
  unsigned char* hb = (unsigned char*)s->h_one;
  for (int i = 0; i < 19; ++i) { if (s->dtype == WBC_F64) ((double*)hb)[o + i] = q[i]; else ((float*)hb)[o + i] = (float)q[i]; }
  for (int i = 0; i < 18; ++i) { if (s->dtype == WBC_F64) ((double*)hb)[o + 19 + i] = v[i]; else ((float*)hb)[o + 19 + i] = (float)v[i]; }
  unsigned char* d = (unsigned char*)s->d_one;
  HIP_TRY(hipMemcpyAsync(d + o * ts, hb + o * ts, 37 * ts, hipMemcpyHostToDevice, nullptr));
  const unsigned long long calls0 = s->calls;   // start-up helper, not a tick: the every-k-th-tick sampling phase stays as it is
  const bool timing0 = s->timing;
  s->timing = false;
  int rc = wbc_dynamics_batch(s, 1, d + o * ts, d + (o + 19) * ts, nullptr, nullptr, nullptr, nullptr, d + (o + 37) * ts, nullptr, nullptr);
  s->timing = timing0; s->calls = calls0;
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(hb + (o + 37) * ts, d + (o + 37) * ts, 18 * ts, hipMemcpyDeviceToHost, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  for (int i = 0; i < 18; ++i) {
    obs_integ[i] = s->dtype == WBC_F64 ? ((double*)hb)[o + 37 + i] : (double)((float*)hb)[o + 37 + i];
    obs_r[i] = 0.0;
  }
  return WBC_OK;
}

extern "C" int wbc_compute_reference(wbc_solver* s, const double* q, const double* v, const double* plan, double t,
                                     double* w_des, double* vdot_des, double* com) {
  if (!s || !q || !v || !plan || !w_des || !vdot_des) return fail(WBC_E_INVALID, "null argument");
  if (!s->d_ref) return fail(WBC_E_INVALID, "call wbc_solver_set_ref_params first");
  ON_DEVICE(s);
  const size_t ts = s->dtype == WBC_F64 ? 8 : 4;
  const int o0 = ONE_SCRATCH;   // scratch region of the pinned image: the tick's part stays as the caller left it
  const int off[] = {o0 + 0, o0 + 19, o0 + 37, o0 + 49, o0 + 55, o0 + 73, o0 + 79};  // q v plan | w_des vdot_des com end
  unsigned char* hb = (unsigned char*)s->h_one;   // pinned staging, one copy each way
  auto put = [&](int o, const double* src, int n) {
    for (int i = 0; i < n; ++i) {
      if (s->dtype == WBC_F64) ((double*)hb)[o + i] = src[i];
      else ((float*)hb)[o + i] = (float)src[i];
    }
  };
  put(off[0], q, 19); put(off[1], v, 18); put(off[2], plan, PLAN_WORDS);
  unsigned char* d = (unsigned char*)s->d_one;
  HIP_TRY(hipMemcpyAsync(d + o0 * ts, hb + o0 * ts, 49 * ts, hipMemcpyHostToDevice, nullptr));
  int rc = wbc_reference_batch(s, 1, d + off[0] * ts, d + off[1] * ts, d + off[2] * ts, t, d + off[3] * ts, d + off[4] * ts,
                               d + off[5] * ts, nullptr);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(hb + (o0 + 49) * ts, d + (o0 + 49) * ts, 30 * ts, hipMemcpyDeviceToHost, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  auto get = [&](int o, double* dst, int n) {
    if (!dst) return;
    for (int i = 0; i < n; ++i)
      dst[i] = s->dtype == WBC_F64 ? ((double*)hb)[o + i] : (double)((float*)hb)[o + i];
  };
  get(off[3], w_des, 6); get(off[4], vdot_des, 18); get(off[5], com, 6);
  return WBC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// dense QPs of run-time size (qp_general.hip.hpp); stateless: no solver handle, the current device
template <class T>
static int qp_dense_launch(size_t N, int n, int m, int meq, const void* H, const void* g, const void* C, const void* d, int max_iter, double tol,
                           void* x, void* lambda, int* status, int* iters, hipStream_t st) {
  wbc::QpGeneralArgs<T> a;
  a.N = N; a.n = n; a.m = m; a.meq = meq; a.max_iter = max_iter; a.tol = (T)tol;
  a.H = (const T*)H; a.g = (const T*)g; a.C = (const T*)C; a.d = (const T*)d;
  a.x = (T*)x; a.lambda = (T*)lambda; a.status = status; a.iters = iters;
  wbc::LaunchCtx L;
  L.st = st;
  HIP_TRY(wbc::k_qp_general<T>(L, a));
  return WBC_OK;
}
extern "C" int wbc_qp_dense_batch(int dtype, size_t N, int n, int m, int meq, const void* H, const void* g, const void* C, const void* d,
                                  int max_iter, double tol, void* x, void* lambda, int* status, int* iters, void* hipStream) {
  if (dtype != WBC_F64 && dtype != WBC_F32) return fail(WBC_E_INVALID, "dtype must be WBC_F64 or WBC_F32");
  if (n < 1 || n > wbc::QPG_MAXN || m < 0 || m > wbc::QPG_MAXM || meq < 0 || meq > m)
    return fail(WBC_E_INVALID, "sizes: 1 <= n <= 36, 0 <= meq <= m <= 64");
  if (!H || !g || !x || !status || (m > 0 && (!C || !d))) return fail(WBC_E_INVALID, "null argument");
  if (max_iter < 0 || !(tol >= 0)) return fail(WBC_E_INVALID, "max_iter >= 0, tol >= 0");
  if (N == 0) return WBC_OK;
  if (N > (size_t)0x7FFFFFFF) return fail(WBC_E_CAPACITY, "N exceeds 2^31 - 1");
  hipStream_t st = (hipStream_t)hipStream;
  return dtype == WBC_F64 ? qp_dense_launch<double>(N, n, m, meq, H, g, C, d, max_iter, tol, x, lambda, status, iters, st)
                          : qp_dense_launch<float>(N, n, m, meq, H, g, C, d, max_iter, tol, x, lambda, status, iters, st);
}

extern "C" const char* wbc_strerror(int st) {
  switch (st) {
    case WBC_OK: return "ok";
    case WBC_E_INVALID: return "invalid argument";
    case WBC_E_IO: return "cannot read URDF";
    case WBC_E_PARSE: return "URDF parse error";
    case WBC_E_TOPOLOGY: return "unsupported robot topology";
    case WBC_E_NODEVICE: return "no usable gfx950 device";
    case WBC_E_HIP: return "HIP runtime error";
    case WBC_E_CAPACITY: return "batch exceeds solver capacity";
    default: return "unknown status";
  }
}
extern "C" const char* wbc_last_error(void) { return g_err.c_str(); }
extern "C" int wbc_abi_version(void) { return 9; }  // 9: wbc_solver_options.fused_pair, wbc_tick_plan.fused = 3 (fused_pair_kernel); 8: wbc_solver_options.tile_tick, wbc_tick_plan.fused = 2 / qp_body = 2 (staged QP tiles); 7: wbc_solver_collect_timing_n (the unsized call writes 5 entries again), wbc_solver_options.multi_threads / multi_spin_us, wbc_multi_tick_gather / wbc_multi_issue_threads / wbc_multi_host_stats; 6: wbc_plan_tick / wbc_solver_plan_tick / wbc_dispatch_thresholds, wbc_solver_invalidate_structural, warm start (wbc_step_batch_warm, wbc_multi_step_batch_warm, wbc_solver_options.rollout_warm, wbc_tick_plan.qp_warm, WBC_PLAN_* flags), wbc_multi_allgather_tau_async / wbc_multi_gather_wait; 5: wbc_qp_dense_batch; 4: wbc_one_map / wbc_one_tick, wbc_solver_options.f32_pack2 and one_zerocopy 2 / 3 (options struct grows at its end: struct_size keeps version-3 callers valid); 3: wbc_solver_options / wbc_solver_create_ex, wbc_observer_init, wbc_multi_* (2: integrate takes Jc, timing arrays have 4 entries, reference / tracking entry points)

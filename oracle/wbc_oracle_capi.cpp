// TEST INFRASTRUCTURE -- C-ABI (ctypes) face of the CPU oracle in wbc_oracle.hpp.
// PARITY UNPINNED (see wbc_oracle.hpp header).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg load this library; the product never does.
// Batch layout here is row-per-state (AoS, numpy-natural): x[N][ncomp].
#include "wbc_oracle.hpp"
#include "qp_general.hpp"
#include <chrono>
#include <omp.h>
#include <new>

using namespace wbco;

struct OracleHandle {
  Model<double> md;
  Model<float> mf;
};

extern "C" {

struct wbco_params {  // mirrors wbco::Params
  double S[6];
  double alpha, fn_min, fn_max, mu_scale, dt;
  int observer_order, max_iter;
  double qp_tol;
  double K1[MAXV], K2[MAXV];
};

static Params to_params(const wbco_params* p) {
  Params P;
  for (int i = 0; i < 6; ++i) P.S[i] = p->S[i];
  P.alpha = p->alpha; P.fn_min = p->fn_min; P.fn_max = p->fn_max; P.mu_scale = p->mu_scale; P.dt = p->dt;
  P.observer_order = p->observer_order; P.max_iter = p->max_iter; P.qp_tol = p->qp_tol;
  for (int i = 0; i < MAXV; ++i) { P.K1[i] = p->K1[i]; P.K2[i] = p->K2[i]; }
  return P;
}

struct wbco_ref_params {  // mirrors wbco::RefParams
  double kp_com[3], kd_com[3], kp_rot[3], kd_rot[3];
  double kp_joint, kd_joint;
  double inertia_nom[3];
  double q_nom[MAXV];
};

static RefParams to_ref(const wbco_ref_params* g) {
  RefParams G;
  for (int i = 0; i < 3; ++i) {
    G.kp_com[i] = g->kp_com[i]; G.kd_com[i] = g->kd_com[i]; G.kp_rot[i] = g->kp_rot[i]; G.kd_rot[i] = g->kd_rot[i];
    G.inertia_nom[i] = g->inertia_nom[i];
  }
  G.kp_joint = g->kp_joint; G.kd_joint = g->kd_joint;
  for (int i = 0; i < MAXV; ++i) G.q_nom[i] = g->q_nom[i];
  return G;
}

int wbco_maxv() { return MAXV; }

void* wbco_model_create(int nb, const int* parent, const double* Rt, const double* rt, const double* axis,
                        const double* mass, const double* com, const double* Ic, int nf, const int* foot_body,
                        const double* foot_off, const double* gravity) {
  if (nb < 1 || nb > MAXB || nf < 0 || nf > MAXF) return nullptr;
  OracleHandle* h = new (std::nothrow) OracleHandle;
  if (!h) return nullptr;
  model_from_flat(h->md, nb, parent, Rt, rt, axis, mass, com, Ic, nf, foot_body, foot_off, gravity);
  model_from_flat(h->mf, nb, parent, Rt, rt, axis, mass, com, Ic, nf, foot_body, foot_off, gravity);
  return h;
}
void wbco_model_destroy(void* h) { delete (OracleHandle*)h; }

#define DEF_API(SUF, T, MODEL)                                                                                  \
  void wbco_dynamics_##SUF(void* hh, int N, const T* q, const T* v, T* M, T* h, T* Jc, T* pf, T* p, T* beta,    \
                           int nthreads) {                                                                      \
    const Model<T>& m = ((OracleHandle*)hh)->MODEL;                                                             \
    const int nv = m.nv(), nq = nv + 1, nm = nv * (nv + 1) / 2, nf = m.nf;                                      \
    _Pragma("omp parallel for num_threads(nthreads) schedule(static)") for (int s = 0; s < N; ++s) {           \
      DynOut<T> o;                                                                                              \
      dynamics(m, q + (size_t)s * nq, v + (size_t)s * nv, o);                                                   \
      if (M) for (int e = 0; e < nm; ++e) M[(size_t)s * nm + e] = o.M[e];                                       \
      if (h) for (int e = 0; e < nv; ++e) h[(size_t)s * nv + e] = o.h[e];                                       \
      if (Jc) for (int e = 0; e < 3 * nf * nv; ++e) Jc[(size_t)s * 3 * nf * nv + e] = o.Jc[e];                  \
      if (pf) for (int e = 0; e < 3 * nf; ++e) pf[(size_t)s * 3 * nf + e] = o.pf[e];                            \
      if (p) for (int e = 0; e < nv; ++e) p[(size_t)s * nv + e] = o.p[e];                                       \
      if (beta) for (int e = 0; e < nv; ++e) beta[(size_t)s * nv + e] = o.beta[e];                              \
    }                                                                                                           \
  }                                                                                                             \
  void wbco_rnea_##SUF(void* hh, int N, const T* q, const T* v, const T* vdot, int with_gravity, T* out) {      \
    const Model<T>& m = ((OracleHandle*)hh)->MODEL;                                                             \
    const int nv = m.nv(), nq = nv + 1;                                                                         \
    for (int s = 0; s < N; ++s) {                                                                               \
      Kin<T> k;                                                                                                 \
      fk(m, q + (size_t)s * nq, k);                                                                             \
      rnea(m, k, v + (size_t)s * nv, vdot ? vdot + (size_t)s * nv : (const T*)nullptr, with_gravity != 0,       \
           out + (size_t)s * nv);                                                                               \
    }                                                                                                           \
  }                                                                                                             \
  void wbco_step_##SUF(void* hh, const wbco_params* pp, int N, const T* q, const T* v, const T* w_des,          \
                       const T* vdot_des, const T* normals, const T* mu, const int* mask, const T* tau_prev,    \
                       const T* f_prev, T* obs_integ, T* obs_r, T* tau, T* f, int* status, int* iters,          \
                       int nthreads, const unsigned* aset_in, unsigned* aset_out) {                            \
    const Model<T>& m = ((OracleHandle*)hh)->MODEL;                                                             \
    const Params P = to_params(pp);                                                                             \
    const int nv = m.nv(), nq = nv + 1, nj = m.nj(), nf = m.nf;                                                 \
    _Pragma("omp parallel for num_threads(nthreads) schedule(static)") for (int s = 0; s < N; ++s) {           \
      StepOut<T> o;                                                                                             \
      T zt[MAXV];                                                                                               \
      for (int i = 0; i < MAXV; ++i) zt[i] = 0;                                                                 \
      step(m, P, q + (size_t)s * nq, v + (size_t)s * nv, w_des + (size_t)s * 6, vdot_des + (size_t)s * nv,      \
           normals + (size_t)s * 3 * nf, mu + (size_t)s * nf, (unsigned)mask[s],                                \
           tau_prev ? tau_prev + (size_t)s * nj : zt, f_prev ? f_prev + (size_t)s * 3 * nf : zt,                \
           obs_integ ? obs_integ + (size_t)s * nv : (T*)nullptr, obs_r ? obs_r + (size_t)s * nv : (T*)nullptr,  \
           o, (DynOut<T>*)nullptr, aset_in ? aset_in + s : (const unsigned*)nullptr,                            \
           aset_out ? aset_out + s : (unsigned*)nullptr);                                                       \
      for (int e = 0; e < nj; ++e) tau[(size_t)s * nj + e] = o.tau[e];                                          \
      for (int e = 0; e < 3 * nf; ++e) f[(size_t)s * 3 * nf + e] = o.f[e];                                      \
      if (status) status[s] = o.status;                                                                         \
      if (iters) iters[s] = o.iters;                                                                            \
    }                                                                                                           \
  }                                                                                                             \
  /* the cpu_baseline leg of bench.py: `reps` passes of wbco_step over the batch inside ONE parallel region (thread start-up and   \
     the fork/join of a region per 4 096-state call are not what is being measured); static schedule, every thread keeps its       \
     scratch on its own stack; *seconds = wall time between two barriers of that region */                                        \
  void wbco_step_timed_##SUF(void* hh, const wbco_params* pp, int N, const T* q, const T* v, const T* w_des,    \
                             const T* vdot_des, const T* normals, const T* mu, const int* mask, const T* tau_prev, \
                             const T* f_prev, T* tau, T* f, int* status, int reps, int nthreads, double* seconds) { \
    const Model<T>& m = ((OracleHandle*)hh)->MODEL;                                                             \
    const Params P = to_params(pp);                                                                             \
    const int nv = m.nv(), nq = nv + 1, nj = m.nj(), nf = m.nf;                                                 \
    double t0 = 0, t1 = 0;                                                                                      \
    _Pragma("omp parallel num_threads(nthreads)") {                                                             \
      _Pragma("omp barrier")                                                                                    \
      _Pragma("omp master") t0 = omp_get_wtime();                                                               \
      for (int rep = 0; rep < reps; ++rep) {                                                                    \
        _Pragma("omp for schedule(static) nowait") for (int s = 0; s < N; ++s) {                               \
          StepOut<T> o;                                                                                         \
          T zt[MAXV], ig[MAXV], rr[MAXV];                                                                       \
          for (int i = 0; i < MAXV; ++i) zt[i] = ig[i] = rr[i] = 0;                                             \
          step(m, P, q + (size_t)s * nq, v + (size_t)s * nv, w_des + (size_t)s * 6, vdot_des + (size_t)s * nv,  \
               normals + (size_t)s * 3 * nf, mu + (size_t)s * nf, (unsigned)mask[s],                            \
               tau_prev ? tau_prev + (size_t)s * nj : zt, f_prev ? f_prev + (size_t)s * 3 * nf : zt, ig, rr, o); \
          for (int e = 0; e < nj; ++e) tau[(size_t)s * nj + e] = o.tau[e];                                      \
          for (int e = 0; e < 3 * nf; ++e) f[(size_t)s * 3 * nf + e] = o.f[e];                                  \
          status[s] = o.status;                                                                                 \
        }                                                                                                       \
      }                                                                                                         \
      _Pragma("omp barrier")                                                                                    \
      _Pragma("omp master") t1 = omp_get_wtime();                                                               \
    }                                                                                                           \
    *seconds = t1 - t0;                                                                                         \
  }                                                                                                             \
  /* horizon ticks of {step, forward dynamics with the planned GRFs, integration}; q, v, tau_prev, f_prev, obs in/out */ \
  void wbco_rollout_##SUF(void* hh, const wbco_params* pp, int N, int horizon, T* q, T* v, const T* w_des,              \
                          const T* vdot_des, const T* normals, const T* mu, const int* mask, const T* tau_ext,         \
                          T* tau_prev, T* f_prev, T* obs_integ, T* obs_r, T* tau_traj, int* status, int nthreads,       \
                          int warm, int* iters_sum) {                                                                  \
    const Model<T>& m = ((OracleHandle*)hh)->MODEL;                                                             \
    const Params P = to_params(pp);                                                                             \
    const int nv = m.nv(), nq = nv + 1, nj = m.nj(), nf = m.nf;                                                 \
    _Pragma("omp parallel for num_threads(nthreads) schedule(static)") for (int s = 0; s < N; ++s) {           \
      T zi[MAXV], zr[MAXV];                                                                                     \
      for (int i = 0; i < MAXV; ++i) zi[i] = zr[i] = 0;                                                         \
      rollout(m, P, horizon, q + (size_t)s * nq, v + (size_t)s * nv, w_des + (size_t)s * 6,                     \
              vdot_des + (size_t)s * nv, normals + (size_t)s * 3 * nf, mu + (size_t)s * nf, (unsigned)mask[s],   \
              tau_ext ? tau_ext + (size_t)s * nv : (const T*)nullptr, tau_prev + (size_t)s * nj,                 \
              f_prev + (size_t)s * 3 * nf, obs_integ ? obs_integ + (size_t)s * nv : zi,                          \
              obs_r ? obs_r + (size_t)s * nv : zr, tau_traj ? tau_traj + (size_t)s * horizon * nj : (T*)nullptr, \
              status ? status + s : (int*)nullptr, warm != 0, iters_sum ? iters_sum + s : (int*)nullptr);       \
    }                                                                                                           \
  }                                                                                                             \
  /* CoM reference generator (a11): plan [N][12] -> w_des [N][6], vdot_des [N][nv], optional com [N][6] */       \
  void wbco_reference_##SUF(void* hh, const wbco_ref_params* gg, int N, const T* q, const T* v, const T* plan,   \
                            T t, T* w_des, T* vdot_des, T* com) {                                               \
    const Model<T>& m = ((OracleHandle*)hh)->MODEL;                                                             \
    const RefParams G = to_ref(gg);                                                                             \
    const int nv = m.nv(), nq = nv + 1;                                                                         \
    for (int s = 0; s < N; ++s)                                                                                 \
      reference(m, G, q + (size_t)s * nq, v + (size_t)s * nv, plan + (size_t)s * PLAN_WORDS, t,                 \
                w_des + (size_t)s * 6, vdot_des + (size_t)s * nv, com ? com + (size_t)s * 6 : (T*)nullptr);     \
  }                                                                                                             \
  void wbco_rollout_tracking_##SUF(void* hh, const wbco_params* pp, const wbco_ref_params* gg, int N,           \
                                   int horizon, T* q, T* v, const T* plan, const T* normals, const T* mu,       \
                                   const int* mask, const T* tau_ext, T* tau_prev, T* f_prev, T* obs_integ,     \
                                   T* obs_r, T* tau_traj, T* com_traj, int* status, int nthreads, int warm) {   \
    const Model<T>& m = ((OracleHandle*)hh)->MODEL;                                                             \
    const Params P = to_params(pp);                                                                             \
    const RefParams G = to_ref(gg);                                                                             \
    const int nv = m.nv(), nq = nv + 1, nj = m.nj(), nf = m.nf;                                                 \
    _Pragma("omp parallel for num_threads(nthreads) schedule(static)") for (int s = 0; s < N; ++s) {           \
      T zi[MAXV], zr[MAXV];                                                                                     \
      for (int i = 0; i < MAXV; ++i) zi[i] = zr[i] = 0;                                                         \
      rollout_tracking(m, P, G, horizon, q + (size_t)s * nq, v + (size_t)s * nv, plan + (size_t)s * PLAN_WORDS, \
                       normals + (size_t)s * 3 * nf, mu + (size_t)s * nf, (unsigned)mask[s],                    \
                       tau_ext ? tau_ext + (size_t)s * nv : (const T*)nullptr, tau_prev + (size_t)s * nj,        \
                       f_prev + (size_t)s * 3 * nf, obs_integ ? obs_integ + (size_t)s * nv : zi,                 \
                       obs_r ? obs_r + (size_t)s * nv : zr,                                                     \
                       tau_traj ? tau_traj + (size_t)s * horizon * nj : (T*)nullptr,                            \
                       com_traj ? com_traj + (size_t)s * horizon * 6 : (T*)nullptr,                             \
                       status ? status + s : (int*)nullptr, warm != 0);                                         \
    }                                                                                                           \
  }                                                                                                             \
  void wbco_forward_dynamics_##SUF(int nv, int nj, int nf, const T* Mp, const T* h, const T* Jc, const T* tau,  \
                                   const T* f, const T* tau_ext, T* vdot) {                                     \
    forward_dynamics(nv, nj, nf, Mp, h, Jc, tau, f, tau_ext, vdot);                                             \
  }                                                                                                             \
  /* dense QP, row-major H[n*n], C[m*n]: min 1/2 x'Hx + g'x  s.t. Cx >= d */                                    \
  int wbco_qp_solve_##SUF(int n, int mm, const T* H, const T* g, const T* C, const T* d, int max_iter, T tol,   \
                          T* x, T* lambda, int* status, const unsigned char* warm, unsigned char* active_out) { \
    if (n > QPN || mm > QPM) { *status = -1; return 0; }                                                        \
    QP<T> qp;                                                                                                   \
    qp.n = n; qp.m = mm;                                                                                        \
    for (int i = 0; i < n; ++i) { qp.g[i] = g[i]; for (int j = 0; j < n; ++j) qp.H[i * QPN + j] = H[i * n + j]; } \
    for (int i = 0; i < mm; ++i) { qp.d[i] = d[i]; for (int j = 0; j < n; ++j) qp.C[i * QPN + j] = C[i * n + j]; } \
    bool wf[QPM], af[QPM];                                                                                      \
    for (int i = 0; i < mm; ++i) wf[i] = warm && warm[i];                                                       \
    const int it = qp_solve_gi(qp, max_iter, tol, x, lambda, status, warm ? wf : (const bool*)nullptr, af);     \
    if (active_out) for (int i = 0; i < mm; ++i) active_out[i] = af[i] ? 1 : 0;                                 \
    return it;                                                                                                  \
  }                                                                                                             \
  /* QP assembly for one state: returns n, writes m; H[QPN*QPN] etc. with leading dimension QPN */              \
  int wbco_qp_assemble_##SUF(const wbco_params* pp, int nf, int mask, const T* pb, const T* pf,                  \
                             const T* normals, const T* mu, const T* b, T* H, T* g, T* C, T* d, int* mout) {    \
    const Params P = to_params(pp);                                                                             \
    V3<T> pfv[MAXF];                                                                                            \
    for (int k = 0; k < nf; ++k) pfv[k] = V3<T>(pf[3 * k], pf[3 * k + 1], pf[3 * k + 2]);                       \
    QP<T> qp;                                                                                                   \
    qp_assemble(P, nf, (unsigned)mask, V3<T>(pb[0], pb[1], pb[2]), pfv, normals, mu, b, qp);                    \
    for (int i = 0; i < qp.n; ++i) { g[i] = qp.g[i]; for (int j = 0; j < qp.n; ++j) H[i * qp.n + j] = qp.H[i * QPN + j]; } \
    for (int i = 0; i < qp.m; ++i) { d[i] = qp.d[i]; for (int j = 0; j < qp.n; ++j) C[i * qp.n + j] = qp.C[i * QPN + j]; } \
    *mout = qp.m;                                                                                               \
    return qp.n;                                                                                                \
  }

DEF_API(f64, double, md)
DEF_API(f32, float, mf)

// general dense QP with equality rows (qp_general.hpp): C_i x = d_i for i < meq, C_i x >= d_i after
int wbco_qp_general_f64(int n, int m, int meq, const double* H, const double* g, const double* C, const double* d, int max_iter,
                        double tol, double* x, double* lambda, int* status) {
  return qp_solve_gi_general<double>(n, m, meq, H, g, C, d, max_iter, tol, x, lambda, status);
}
int wbco_qp_general_f32(int n, int m, int meq, const float* H, const float* g, const float* C, const float* d, int max_iter,
                        float tol, float* x, float* lambda, int* status) {
  return qp_solve_gi_general<float>(n, m, meq, H, g, C, d, max_iter, tol, x, lambda, status);
}
// ... over a batch (arrays of N problems of one size, problem-major), OpenMP over problems
void wbco_qp_general_batch_f64(int N, int n, int m, int meq, const double* H, const double* g, const double* C, const double* d,
                               int max_iter, double tol, double* x, double* lambda, int* status, int* iters, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int s = 0; s < N; ++s)
    iters[s] = qp_solve_gi_general<double>(n, m, meq, H + (size_t)s * n * n, g + (size_t)s * n, C + (size_t)s * m * n, d + (size_t)s * m,
                                           max_iter, tol, x + (size_t)s * n, lambda + (size_t)s * m, status + s);
}

// Per-QP wall time on one thread (SURVEY.md 8d "CPU: p50 of per-QP wall time over the batch"): for every state the
// dynamics run untimed, then {a7 assembly + a8 solve} is timed with steady_clock; ns_out[N], iters_out[N] (may be null).
void wbco_qp_time_f64(void* hh, const wbco_params* pp, int N, const double* q, const double* v, const double* w_des,
                      const double* normals, const double* mu, const int* mask, double* ns_out, int* iters_out) {
  const Model<double>& m = ((OracleHandle*)hh)->md;
  const Params P = to_params(pp);
  const int nv = m.nv(), nq = nv + 1, nf = m.nf;
  for (int s = 0; s < N; ++s) {
    DynOut<double> d;
    dynamics(m, q + (size_t)s * nq, v + (size_t)s * nv, d);
    V3<double> pf[MAXF];
    for (int f = 0; f < nf; ++f) pf[f] = V3<double>(d.pf[3 * f], d.pf[3 * f + 1], d.pf[3 * f + 2]);
    const double* qs = q + (size_t)s * nq;
    double x[QPN], lam[QPM];
    int status = 0;
    const auto t0 = std::chrono::steady_clock::now();
    QP<double> qp;
    qp_assemble(P, nf, (unsigned)mask[s], V3<double>(qs[0], qs[1], qs[2]), pf, normals + (size_t)s * 3 * nf,
                mu + (size_t)s * nf, w_des + (size_t)s * 6, qp);
    const int it = qp_solve_gi(qp, P.max_iter, P.qp_tol, x, lam, &status);
    const auto t1 = std::chrono::steady_clock::now();
    ns_out[s] = std::chrono::duration<double, std::nano>(t1 - t0).count() + 0.0 * x[0];
    if (iters_out) iters_out[s] = it;
  }
}

}  // extern "C"

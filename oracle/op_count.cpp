// TEST INFRASTRUCTURE -- instrumented operation count of one control step of the CPU oracle (SURVEY.md 8d:
// "algorithmic flops / step: replace the estimate with an instrumented op count from the oracle").
// The oracle is templated on its scalar; here it is instantiated with a scalar that counts every arithmetic
// operation it performs, per stage.  The counts describe THIS restatement (dense loops over structural zeros
// included), i.e. an upper bound on what a structure-exploiting kernel needs.  Never used by the product.
#include <cmath>
#include <limits>

namespace wbco_count {
struct Counters { long long add, mul, div, sqrt_, trig, cmp; };
static thread_local Counters g_c;

struct Cnt {
  double v;
  Cnt() : v(0) {}
  Cnt(double x) : v(x) {}
  Cnt(int x) : v(x) {}
  explicit operator double() const { return v; }
  Cnt operator-() const { return Cnt(-v); }
  Cnt& operator+=(Cnt o) { ++g_c.add; v += o.v; return *this; }
  Cnt& operator-=(Cnt o) { ++g_c.add; v -= o.v; return *this; }
  Cnt& operator*=(Cnt o) { ++g_c.mul; v *= o.v; return *this; }
  Cnt& operator/=(Cnt o) { ++g_c.div; v /= o.v; return *this; }
};
inline Cnt operator+(Cnt a, Cnt b) { ++g_c.add; return Cnt(a.v + b.v); }
inline Cnt operator-(Cnt a, Cnt b) { ++g_c.add; return Cnt(a.v - b.v); }
inline Cnt operator*(Cnt a, Cnt b) { ++g_c.mul; return Cnt(a.v * b.v); }
inline Cnt operator/(Cnt a, Cnt b) { ++g_c.div; return Cnt(a.v / b.v); }
#define WBCO_CMP(OP) inline bool operator OP(Cnt a, Cnt b) { ++g_c.cmp; return a.v OP b.v; }
WBCO_CMP(<) WBCO_CMP(>) WBCO_CMP(<=) WBCO_CMP(>=) WBCO_CMP(==) WBCO_CMP(!=)
#undef WBCO_CMP
inline Cnt sqrt(Cnt a) { ++g_c.sqrt_; return Cnt(std::sqrt(a.v)); }
inline Cnt sin(Cnt a) { ++g_c.trig; return Cnt(std::sin(a.v)); }
inline Cnt cos(Cnt a) { ++g_c.trig; return Cnt(std::cos(a.v)); }
inline Cnt fabs(Cnt a) { return Cnt(std::fabs(a.v)); }
inline Cnt fmax(Cnt a, Cnt b) { ++g_c.cmp; return Cnt(std::fmax(a.v, b.v)); }
}  // namespace wbco_count

namespace std {
template <> struct numeric_limits<wbco_count::Cnt> {
  static wbco_count::Cnt epsilon() { return numeric_limits<double>::epsilon(); }
  static wbco_count::Cnt infinity() { return numeric_limits<double>::infinity(); }
};
}  // namespace std

#include "wbc_oracle.hpp"

using namespace wbco;
using wbco_count::Cnt;
using wbco_count::Counters;
using wbco_count::g_c;

extern "C" {

struct wbco_params_oc {  // same layout as wbco_params in wbc_oracle_capi.cpp
  double S[6];
  double alpha, fn_min, fn_max, mu_scale, dt;
  int observer_order, max_iter;
  double qp_tol;
  double K1[MAXV], K2[MAXV];
};

// Runs one oracle step for one state with the counting scalar.  counts[stage][6] = {add, mul, div, sqrt, trig, cmp}
// for stage 0 = dynamics (a1-a5: FK, RNEA, CRBA, Jacobians, momentum/C^T v), 1 = observer update (a6),
// 2 = QP assembly (a7), 3 = QP solve (a8), 4 = torque map (a9).  Returns the QP iteration count; tau_out/f_out are
// the step's results (so the caller can check them against the double instantiation).
int wbco_op_count(int nb, const int* parent, const double* Rt, const double* rt, const double* axis,
                  const double* mass, const double* com, const double* Ic, int nf, const int* foot_body,
                  const double* foot_off, const double* gravity, const wbco_params_oc* pp, const double* q,
                  const double* v, const double* w_des, const double* vdot_des, const double* normals,
                  const double* mu, int mask, const double* tau_prev, const double* f_prev, double* obs_integ,
                  double* obs_r, long long* counts, double* tau_out, double* f_out) {
  if (nb < 1 || nb > MAXB || nf < 0 || nf > MAXF) return -1;
  static Model<Cnt> m;  // large; not on the stack
  model_from_flat(m, nb, parent, Rt, rt, axis, mass, com, Ic, nf, foot_body, foot_off, gravity);
  Params P;
  for (int i = 0; i < 6; ++i) P.S[i] = pp->S[i];
  P.alpha = pp->alpha; P.fn_min = pp->fn_min; P.fn_max = pp->fn_max; P.mu_scale = pp->mu_scale; P.dt = pp->dt;
  P.observer_order = pp->observer_order; P.max_iter = pp->max_iter; P.qp_tol = pp->qp_tol;
  for (int i = 0; i < MAXV; ++i) { P.K1[i] = pp->K1[i]; P.K2[i] = pp->K2[i]; }
  const int nv = m.nv(), nq = nv + 1, nj = m.nj();
  Cnt cq[MAXV + 1], cv[MAXV], cw[6], cvd[MAXV], cn[3 * MAXF], cmu[MAXF], ctp[MAXV], cfp[3 * MAXF], ci[MAXV], cr[MAXV];
  for (int i = 0; i < nq; ++i) cq[i] = q[i];
  for (int i = 0; i < nv; ++i) { cv[i] = v[i]; cvd[i] = vdot_des[i]; ci[i] = obs_integ ? obs_integ[i] : 0.0; cr[i] = obs_r ? obs_r[i] : 0.0; }
  for (int i = 0; i < 6; ++i) cw[i] = w_des[i];
  for (int i = 0; i < 3 * nf; ++i) { cn[i] = normals[i]; cfp[i] = f_prev ? f_prev[i] : 0.0; }
  for (int i = 0; i < nf; ++i) cmu[i] = mu[i];
  for (int i = 0; i < nj; ++i) ctp[i] = tau_prev ? tau_prev[i] : 0.0;

  auto snap = [&](int stage) {
    long long* c = counts + 6 * stage;
    c[0] = g_c.add; c[1] = g_c.mul; c[2] = g_c.div; c[3] = g_c.sqrt_; c[4] = g_c.trig; c[5] = g_c.cmp;
    g_c = Counters{};
  };
  // the same sequence as wbco::step(), stage by stage
  g_c = Counters{};
  static DynOut<Cnt> d;
  dynamics(m, cq, cv, d);
  snap(0);
  Cnt rhat[MAXV];
  for (int i = 0; i < nv; ++i) rhat[i] = 0;
  if (P.observer_order > 0) {
    observer_update(nv, m.nf, P, d.p, d.beta, d.Jc, ctp, cfp, ci, cr);
    for (int i = 0; i < nv; ++i) rhat[i] = cr[i];
  }
  snap(1);
  Cnt b[6];
  for (int i = 0; i < 6; ++i) b[i] = cw[i] - rhat[i];
  V3<Cnt> pf[MAXF];
  for (int f = 0; f < m.nf; ++f) pf[f] = V3<Cnt>(d.pf[3 * f], d.pf[3 * f + 1], d.pf[3 * f + 2]);
  static QP<Cnt> qp;
  qp_assemble(P, m.nf, (unsigned)mask, V3<Cnt>(cq[0], cq[1], cq[2]), pf, cn, cmu, b, qp);
  snap(2);
  Cnt x[QPN], lam[QPM];
  int status = 0;
  const int iters = qp_solve_gi(qp, P.max_iter, (Cnt)P.qp_tol, x, lam, &status);
  snap(3);
  Cnt fo[3 * MAXF];
  for (int e = 0; e < 3 * m.nf; ++e) fo[e] = 0;
  for (int i = 0; i < qp.n; ++i) fo[3 * qp.foot_of_var[i] + (i % 3)] = x[i];
  for (int j = 0; j < nj; ++j) {
    const int row = 6 + j;
    Cnt t = d.h[row] - rhat[row];
    for (int c = 0; c < nv; ++c) t += d.M[midx(nv, row, c)] * cvd[c];
    for (int e = 0; e < 3 * m.nf; ++e) t -= d.Jc[e * nv + row] * fo[e];
    tau_out[j] = t.v;
  }
  snap(4);
  for (int e = 0; e < 3 * m.nf; ++e) f_out[e] = fo[e].v;
  if (obs_integ) for (int i = 0; i < nv; ++i) { obs_integ[i] = ci[i].v; obs_r[i] = cr[i].v; }
  return iters;
}

// Operation count of a whole ROLLOUT of one state (configs[4]: `horizon` dependent ticks of {control step, forward dynamics,
// integration}; warm != 0: QPs after the first tick start from the previous tick's active set, as the HIP path's rollouts do).
// counts[6] = {add, mul, div, sqrt, trig, cmp} summed over the horizon; returns the sum of the QP iteration counts.
int wbco_op_count_rollout(int nb, const int* parent, const double* Rt, const double* rt, const double* axis, const double* mass,
                          const double* com, const double* Ic, int nf, const int* foot_body, const double* foot_off,
                          const double* gravity, const wbco_params_oc* pp, int horizon, int warm, const double* q, const double* v,
                          const double* w_des, const double* vdot_des, const double* normals, const double* mu, int mask,
                          const double* tau_ext, const double* obs_integ, long long* counts) {
  if (nb < 1 || nb > MAXB || nf < 0 || nf > MAXF) return -1;
  static Model<Cnt> m;
  model_from_flat(m, nb, parent, Rt, rt, axis, mass, com, Ic, nf, foot_body, foot_off, gravity);
  Params P;
  for (int i = 0; i < 6; ++i) P.S[i] = pp->S[i];
  P.alpha = pp->alpha; P.fn_min = pp->fn_min; P.fn_max = pp->fn_max; P.mu_scale = pp->mu_scale; P.dt = pp->dt;
  P.observer_order = pp->observer_order; P.max_iter = pp->max_iter; P.qp_tol = pp->qp_tol;
  for (int i = 0; i < MAXV; ++i) { P.K1[i] = pp->K1[i]; P.K2[i] = pp->K2[i]; }
  const int nv = m.nv(), nq = nv + 1;
  Cnt cq[MAXV + 1], cv[MAXV], cw[6], cvd[MAXV], cn[3 * MAXF], cmu[MAXF], cte[MAXV], ctp[MAXV], cfp[3 * MAXF], ci[MAXV], cr[MAXV];
  for (int i = 0; i < nq; ++i) cq[i] = q[i];
  for (int i = 0; i < nv; ++i) { cv[i] = v[i]; cvd[i] = vdot_des[i]; cte[i] = tau_ext ? tau_ext[i] : 0.0; ci[i] = obs_integ ? obs_integ[i] : 0.0; cr[i] = 0.0; ctp[i] = 0.0; }
  for (int i = 0; i < 6; ++i) cw[i] = w_des[i];
  for (int i = 0; i < 3 * nf; ++i) { cn[i] = normals[i]; cfp[i] = 0.0; }
  for (int i = 0; i < nf; ++i) cmu[i] = mu[i];
  int st = 0, its = 0;
  g_c = Counters{};
  rollout(m, P, horizon, cq, cv, cw, cvd, cn, cmu, (unsigned)mask, tau_ext ? cte : (const Cnt*)nullptr, ctp, cfp, ci, cr, (Cnt*)nullptr, &st,
          warm != 0, &its);
  counts[0] = g_c.add; counts[1] = g_c.mul; counts[2] = g_c.div; counts[3] = g_c.sqrt_; counts[4] = g_c.trig; counts[5] = g_c.cmp;
  return its;
}

}  // extern "C"

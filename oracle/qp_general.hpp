// TEST INFRASTRUCTURE (CPU oracle, never linked into the product): the Goldfarb-Idnani dual active-set method of
// wbc_oracle.hpp (qp_solve_gi, written from Goldfarb & Idnani, Math. Prog. 27 (1983)) for run-time sizes and with equality rows:
//
//     min 1/2 x^T H x + g^T x     s.t.   C_i x  = d_i  (i <  meq),     C_i x >= d_i  (meq <= i < m)
//
// H[n*n], C[m*n] row-major, H symmetric positive definite.  It is what the product's general one-QP-per-wavefront kernel
// (csrc/qp_general.hip.hpp) is checked against -- SURVEY.md section 7 asks for a QP "parametric in nvar <= 36, ncon <= 48"
// because the reference's variable set is unknown (README.md:11 says only "optimization problem based on the modulation of
// ground reaction forces"; the controller's source is an absent submodule, .gitmodules:4-6).  PARITY UNPINNED like the rest
// of the oracle; pinned instead by (i) bit-equality with qp_solve_gi on the 12-variable GRF QPs and (ii) scipy on random ones
// (tests/test_qp_general_oracle.py).
//
// Equality rows (the paper's section 4 remark): they enter the active set first, in order, with the sign of the normal chosen so
// that the row is "violated" in the >= sense; their multipliers are unrestricted, so they take no part in the ratio test and are
// never dropped.  A dependent equality row is skipped when it already holds to `tol` and makes the problem infeasible otherwise.
// Reported multipliers: lambda_i >= 0 for inequalities, any sign for equalities (Lagrangian  f(x) - lambda^T (C x - d)).
// status: 0 optimal, 1 iteration limit, 2 infeasible, 3 H not positive definite, -1 bad sizes.  Returns the number of step-2 passes.
#pragma once
#include <cmath>
#include <limits>
#include <vector>

namespace wbco {

constexpr int QPG_MAXN = 36, QPG_MAXM = 64;   // (a constraint per lane: 64 is the wavefront)

template <class T>
int qp_solve_gi_general(int n, int m, int meq, const T* H, const T* g, const T* C, const T* d, int max_iter, T tol, T* x,
                        T* lambda, int* status) {
  *status = 0;
  for (int i = 0; i < m; ++i) lambda[i] = 0;
  if (n < 0 || m < 0 || meq < 0 || meq > m || n > QPG_MAXN || m > QPG_MAXM) { *status = -1; return 0; }
  if (n == 0) return 0;
  const int ld = n;
  const T eps = std::numeric_limits<T>::epsilon();
  const T INF = std::numeric_limits<T>::infinity();
  std::vector<T> L(n * n, 0), J(n * n, 0), R(n * n, 0), z(n), r(n + 1), dd(n), np(n), u(m + 1, 0), sg(m + 1, 1);
  std::vector<int> A(m + 1, -1);
  std::vector<char> active(m, 0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      T s = H[i * ld + j];
      for (int k = 0; k < j; ++k) s -= L[i * ld + k] * L[j * ld + k];
      if (i == j && !(s > 0)) { *status = 3; for (int k = 0; k < n; ++k) x[k] = 0; return 0; }
      L[i * ld + j] = (i == j) ? std::sqrt(s) : s / L[j * ld + j];
    }
  for (int c = 0; c < n; ++c)
    for (int i = n - 1; i >= 0; --i) {
      T s = (i == c) ? (T)1 : (T)0;
      for (int k = i + 1; k < n; ++k) s -= L[k * ld + i] * J[k * ld + c];
      J[i * ld + c] = s / L[i * ld + i];
    }
  {
    std::vector<T> t(n);
    for (int i = 0; i < n; ++i) { T s = 0; for (int k = 0; k < n; ++k) s += J[k * ld + i] * g[k]; t[i] = s; }
    for (int i = 0; i < n; ++i) { T s = 0; for (int k = 0; k < n; ++k) s += J[i * ld + k] * t[k]; x[i] = -s; }
  }
  int iq = 0, iter = 0, neq_in = 0;   // neq_in: equality rows in the active set (they occupy slots 0 .. neq_in-1 and stay)
  T Rnorm = 1;

  auto drop = [&](int l) {
    int qq = 0;
    while (A[qq] != l) ++qq;
    active[l] = 0;
    for (int i = qq; i < iq - 1; ++i) {
      A[i] = A[i + 1]; u[i] = u[i + 1]; sg[i] = sg[i + 1];
      for (int j = 0; j < n; ++j) R[j * ld + i] = R[j * ld + i + 1];
    }
    A[iq - 1] = A[iq]; u[iq - 1] = u[iq]; sg[iq - 1] = sg[iq];
    A[iq] = -1; u[iq] = 0; sg[iq] = 1;
    --iq;
    for (int j = qq; j < iq; ++j) {
      T cc = R[j * ld + j], ss = R[(j + 1) * ld + j];
      T h = std::sqrt(cc * cc + ss * ss);
      if (h == 0) continue;
      cc /= h; ss /= h;
      R[(j + 1) * ld + j] = 0;
      R[j * ld + j] = h;
      for (int k = j + 1; k < iq; ++k) {
        T t1 = R[j * ld + k], t2 = R[(j + 1) * ld + k];
        R[j * ld + k] = cc * t1 + ss * t2;
        R[(j + 1) * ld + k] = -ss * t1 + cc * t2;
      }
      for (int k = 0; k < n; ++k) {
        T t1 = J[k * ld + j], t2 = J[k * ld + j + 1];
        J[k * ld + j] = cc * t1 + ss * t2;
        J[k * ld + j + 1] = -ss * t1 + cc * t2;
      }
    }
  };
  auto slack = [&](int i) { T s = -d[i]; for (int k = 0; k < n; ++k) s += C[i * ld + k] * x[k]; return s; };

  int next_eq = 0;
  while (true) {
    // step 1: the next equality row, then the most violated inactive inequality
    int ip = -1;
    T sip = 0, sign = 1;
    bool is_eq = false;
    if (next_eq < meq) {   // (a row that already holds is added all the same, by a step of length zero: later steps must keep it)
      const T s = slack(next_eq);
      ip = next_eq++; is_eq = true; sign = s > 0 ? (T)-1 : (T)1; sip = -std::fabs(s);
    }
    if (ip < 0) {
      T smin = -tol;
      for (int i = meq; i < m; ++i) {
        if (active[i]) continue;
        const T s = slack(i);
        if (s < smin) { smin = s; ip = i; }
      }
      if (ip < 0) break;
      sip = smin;
    }
    for (int k = 0; k < n; ++k) np[k] = sign * C[ip * ld + k];
    u[iq] = 0; A[iq] = ip; sg[iq] = sign;
    bool skipped = false;
    while (true) {
      if (++iter > max_iter) { *status = 1; goto done; }
      for (int i = 0; i < n; ++i) { T s = 0; for (int k = 0; k < n; ++k) s += J[k * ld + i] * np[k]; dd[i] = s; }
      for (int i = 0; i < n; ++i) { T s = 0; for (int j = iq; j < n; ++j) s += J[i * ld + j] * dd[j]; z[i] = s; }
      for (int i = iq - 1; i >= 0; --i) {
        T s = dd[i];
        for (int j = i + 1; j < iq; ++j) s -= R[i * ld + j] * r[j];
        r[i] = s / R[i * ld + i];
      }
      int l = -1;
      T t1 = INF, t2 = INF;
      for (int k = neq_in; k < iq; ++k)
        if (r[k] > 0 && u[k] / r[k] < t1) { t1 = u[k] / r[k]; l = A[k]; }
      T dn2 = 0, znp = 0;
      for (int j = iq; j < n; ++j) dn2 += dd[j] * dd[j];
      for (int k = 0; k < n; ++k) znp += z[k] * np[k];
      if (dn2 > (eps * Rnorm) * (eps * Rnorm) && znp > 0) t2 = -sip / znp;
      if (t1 == INF && t2 == INF) {
        if (is_eq && -sip <= tol) { skipped = true; break; }   // dependent equality row that already holds
        *status = 2; goto done;
      }
      if (t2 == INF) {
        for (int k = 0; k < iq; ++k) u[k] -= t1 * r[k];
        u[iq] += t1;
        drop(l);
        continue;
      }
      const bool full = !(t1 < t2);
      const T t = full ? t2 : t1;
      for (int k = 0; k < n; ++k) x[k] += t * z[k];
      for (int k = 0; k < iq; ++k) u[k] -= t * r[k];
      u[iq] += t;
      if (!full) {
        drop(l);
        sip = sign * slack(ip);   // (never reached for an equality row: nothing can block while only equalities are active)
        continue;
      }
      for (int j = n - 1; j >= iq + 1; --j) {
        T cc = dd[j - 1], ss = dd[j];
        T h = std::sqrt(cc * cc + ss * ss);
        if (h == 0) continue;
        cc /= h; ss /= h;
        dd[j - 1] = h; dd[j] = 0;
        for (int k = 0; k < n; ++k) {
          T a1 = J[k * ld + j - 1], a2 = J[k * ld + j];
          J[k * ld + j - 1] = cc * a1 + ss * a2;
          J[k * ld + j] = -ss * a1 + cc * a2;
        }
      }
      for (int i = 0; i <= iq; ++i) R[i * ld + iq] = dd[i];
      for (int i = iq + 1; i < n; ++i) R[i * ld + iq] = 0;
      Rnorm = std::fmax(Rnorm, std::fabs(dd[iq]));
      active[ip] = 1;
      if (is_eq) ++neq_in;   // equality rows sit in front of every inequality: all of them are added before the first inequality
      ++iq;
      break;
    }
    if (skipped) { A[iq] = -1; u[iq] = 0; sg[iq] = 1; active[ip] = 1; }
  }
done:
  for (int i = 0; i < iq; ++i) lambda[A[i]] = sg[i] * u[i];
  return iter;
}

}  // namespace wbco

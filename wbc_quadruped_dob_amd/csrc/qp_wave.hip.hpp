// GRF QP + torque map kernel for gfx950: SURVEY.md 8(a) units a7 (assembly), a8 (solve), a9 (torque map).
//
// Mapping: ONE QP PER WAVEFRONT, all factors resident in LDS (north_star).  n = 3*n_stance <= 12
// variables, m = 6*n_stance <= 24 friction-pyramid / normal-force rows.  Lane roles:
//   lane i < n   : variable i      (row i of J when multiplying, x_i, z_i)
//   lane c < n   : column c of J^T n_p (d_c), column c of R
//   lane j < m   : inequality j    (its 3-sparse row lives in that lane's registers)
// Control flow is wave-uniform (every decision is a wave reduction), so there is no divergence
// inside a QP; divergence between QPs is between wavefronts, which the CU scheduler absorbs.
//
// Algorithm: Goldfarb & Idnani (1983) dual active set, as the CPU oracle, with two wave-friendly
// changes that leave the iterates mathematically identical:
//   * adding a constraint uses ONE Householder reflection of the columns iq..n-1 of J (built from
//     z = J2 d2, which the step already computed) instead of a chain of n-iq-1 Givens rotations:
//     O(1) dependent steps instead of O(n);
//   * r = R^-1 d is a column-oriented back-substitution (iq dependent steps, lanes update rows).
// Dropping a constraint (rare) keeps the Givens re-triangularisation.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace wbc {

#define WBC_DEV __device__ __forceinline__
// LDS ops of one wave execute in issue order; the fence stops the compiler from moving LDS
// accesses across the point and drains lgkmcnt so cross-lane hand-offs through LDS are safe.
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                     __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

template <class T> struct QpLds {
  static constexpr int LD = 13;  // odd leading dimension: row and column walks are both conflict-free
  T J[12 * LD];
  T R[12 * LD];                  // Cholesky factor L during set-up, then R of the active set
  T x[12], d[12], w[12], t[12], g[12], u[13], rdinv[12], linv[12];
  T cn[24 * 3], rhs[24];
  T in[WS_WORDS];
  T nrm[12], mu[4];
  int A[13];
};

template <class T> WBC_DEV T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// argmin over the wave with smallest-index tie break; every lane returns the same pair
template <class T> WBC_DEV void wave_argmin(T& v, int& idx) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const T ov = __shfl_xor(v, o);
    const int oi = __shfl_xor(idx, o);
    const bool take = (ov < v) || (ov == v && oi < idx);
    v = take ? ov : v;
    idx = take ? oi : idx;
  }
}
template <class T> WBC_DEV T bcast(T v, int lane) { return __shfl(v, lane); }
WBC_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <class T> struct Lim;
template <> struct Lim<double> { static constexpr double eps = 2.220446049250313e-16; static constexpr double inf = __builtin_huge_val(); };
template <> struct Lim<float> { static constexpr float eps = 1.1920929e-07f; static constexpr float inf = __builtin_huge_valf(); };

WBC_DEV double sqrt_t(double x) { return sqrt(x); }
WBC_DEV float sqrt_t(float x) { return sqrtf(x); }
WBC_DEV double fabs_t(double x) { return fabs(x); }
WBC_DEV float fabs_t(float x) { return fabsf(x); }

// element (r,a) of [d]x
template <class T> WBC_DEV T skew_el(const T* d, int r, int a) {
  // [[0,-dz,dy],[dz,0,-dx],[-dy,dx,0]]
  if (r == a) return (T)0;
  const int k = 3 - r - a;                    // the remaining axis
  const T sgn = ((a - r + 3) % 3 == 1) ? (T)-1 : (T)1;  // (r,a)=(0,1),(1,2),(2,0) -> -
  return sgn * d[k];
}

struct QpJidx { int j[12]; };  // caller's joint index of leg-major joint 3l+k

template <class T>
__global__ __launch_bounds__(256) void qp_wave_kernel(DevParams<T> prm, QpArgs<T> a, QpJidx jmap) {
  using L = QpLds<T>;
  constexpr int LD = L::LD;
  __shared__ L lds_all[4];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const size_t s = (size_t)blockIdx.x * 4 + wv;
  const size_t N = a.N;
  if (s >= N) return;  // whole wave leaves; only wave-level synchronisation below
  L& S = lds_all[wv];
  const T INF = Lim<T>::inf, EPS = Lim<T>::eps;

  // ------------------------------------------------------------------ inputs -> LDS
  S.in[lane] = a.ws[(size_t)lane * N + s];
  if (lane < WS_WORDS - 64) S.in[64 + lane] = a.ws[(size_t)(64 + lane) * N + s];
  if (lane < 12) S.nrm[lane] = a.normals[(size_t)lane * N + s];
  if (lane < 4) S.mu[lane] = a.mu[(size_t)lane * N + s];
  const int mask = uni(a.mask[s]) & 0xF;
  int st[4];
  int ns = 0;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const bool on = (mask >> f) & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (on && ns == q) st[q] = f;
    ns += on ? 1 : 0;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) if (q >= ns) st[q] = 0;
  const int n = 3 * ns, m = 6 * ns;
  WSYNC();

  int status = 0, iter = 0;
  if (n > 0) {
    // ---------------------------------------------------------------- assemble H (into S.R) and g
    for (int e = lane; e < 144; e += 64) {
      const int i = e / 12, j = e - 12 * i;
      if (i < n && j < n) {
        const int si = i / 3, ai = i - 3 * si, sj = j / 3, aj = j - 3 * sj;
        const int fi = st[0] * (si == 0) + st[1] * (si == 1) + st[2] * (si == 2) + st[3] * (si == 3);
        const int fj = st[0] * (sj == 0) + st[1] * (sj == 1) + st[2] * (sj == 2) + st[3] * (sj == 3);
        const T* di = &S.in[WS_D + 3 * fi];
        const T* dj = &S.in[WS_D + 3 * fj];
        T h = (ai == aj) ? prm.S[ai] : (T)0;
        if (i == j) h += prm.alpha;
#pragma unroll
        for (int r = 0; r < 3; ++r) h += prm.S[3 + r] * skew_el(di, r, ai) * skew_el(dj, r, aj);
        S.R[i * LD + j] = h;
      }
    }
    if (lane < n) {
      const int si = lane / 3, ai = lane - 3 * si;
      const int fi = st[0] * (si == 0) + st[1] * (si == 1) + st[2] * (si == 2) + st[3] * (si == 3);
      const T* di = &S.in[WS_D + 3 * fi];
      T gg = prm.S[ai] * S.in[WS_B + ai];
#pragma unroll
      for (int r = 0; r < 3; ++r) gg += skew_el(di, r, ai) * prm.S[3 + r] * S.in[WS_B + 3 + r];
      S.g[lane] = -gg;
    }
    // ---------------------------------------------------------------- constraint rows (registers + LDS)
    T c0 = 0, c1 = 0, c2 = 0, crhs = 0;
    int cslot = 0;
    if (lane < m) {
      cslot = lane / 6;
      const int tt = lane - 6 * cslot;
      const int f = st[0] * (cslot == 0) + st[1] * (cslot == 1) + st[2] * (cslot == 2) + st[3] * (cslot == 3);
      T nx = S.nrm[3 * f], ny = S.nrm[3 * f + 1], nz = S.nrm[3 * f + 2];
      const T il = (T)1 / sqrt_t(nx * nx + ny * ny + nz * nz);
      nx *= il; ny *= il; nz *= il;
      // tangent basis: ref = ex unless the normal is nearly along x
      const bool usex = fabs_t(nx) < (T)0.9;
      const T rx = usex ? (T)1 : (T)0, ry = usex ? (T)0 : (T)1;
      const T rd = rx * nx + ry * ny;
      T t1x = rx - nx * rd, t1y = ry - ny * rd, t1z = -nz * rd;
      const T it = (T)1 / sqrt_t(t1x * t1x + t1y * t1y + t1z * t1z);
      t1x *= it; t1y *= it; t1z *= it;
      const T t2x = ny * t1z - nz * t1y, t2y = nz * t1x - nx * t1z, t2z = nx * t1y - ny * t1x;
      const T mt = S.mu[f] * prm.mu_scale;
      const T tx = (tt < 2) ? t1x : t2x, ty = (tt < 2) ? t1y : t2y, tz = (tt < 2) ? t1z : t2z;
      const T sg = (tt & 1) ? (T)1 : (T)-1;
      if (tt < 4) { c0 = mt * nx + sg * tx; c1 = mt * ny + sg * ty; c2 = mt * nz + sg * tz; crhs = 0; }
      else if (tt == 4) { c0 = nx; c1 = ny; c2 = nz; crhs = prm.fn_min; }
      else { c0 = -nx; c1 = -ny; c2 = -nz; crhs = -prm.fn_max; }
      S.cn[3 * lane] = c0; S.cn[3 * lane + 1] = c1; S.cn[3 * lane + 2] = c2;
      S.rhs[lane] = crhs;
    }
    WSYNC();

    // ---------------------------------------------------------------- Cholesky H = L L^T (right-looking, in S.R)
    for (int j = 0; j < n; ++j) {
      const T piv = S.R[j * LD + j];
      const T inv = (T)1 / sqrt_t(piv);
      WSYNC();
      if (lane >= j && lane < n) S.R[lane * LD + j] = S.R[lane * LD + j] * inv;
      if (lane == 0) S.linv[j] = inv;
      WSYNC();
      for (int e = lane; e < 144; e += 64) {
        const int i = e / 12, k = e - 12 * i;
        if (k > j && i >= k && i < n) S.R[i * LD + k] -= S.R[i * LD + j] * S.R[k * LD + j];
      }
      WSYNC();
    }
    // ---------------------------------------------------------------- J = L^-T : lane c solves L^T x = e_c
    {
      T xr[12];
#pragma unroll
      for (int i = 11; i >= 0; --i) {
        xr[i] = 0;
        if (i < n) {
          T acc = (i == lane) ? (T)1 : (T)0;
#pragma unroll
          for (int k = i + 1; k < 12; ++k)
            if (k < n) acc -= S.R[k * LD + i] * xr[k];
          xr[i] = acc * S.linv[i];
        }
      }
      // t = J^T g (own column), then J to LDS
      T tc = 0;
#pragma unroll
      for (int i = 0; i < 12; ++i) if (i < n) tc += xr[i] * S.g[i];
      if (lane < n) {
        S.t[lane] = tc;
#pragma unroll
        for (int i = 0; i < 12; ++i) if (i < n) S.J[i * LD + lane] = xr[i];
      }
    }
    WSYNC();
    // x = -J t
    if (lane < n) {
      T acc = 0;
      for (int c = 0; c < n; ++c) acc += S.J[lane * LD + c] * S.t[c];
      S.x[lane] = -acc;
    }
    if (lane < 13) { S.u[lane] = 0; S.A[lane] = -1; }
    WSYNC();

    // ---------------------------------------------------------------- dual active-set iterations
    int iq = 0;
    bool active = false;   // per constraint lane
    T Rnorm = 1;
    bool done = false;
    while (!done) {
      // step 1: most violated inactive constraint
      T sj = INF;
      if (lane < m) {
        const T v = c0 * S.x[3 * cslot] + c1 * S.x[3 * cslot + 1] + c2 * S.x[3 * cslot + 2] - crhs;
        if (!active && v < -prm.qp_tol) sj = v;
      }
      T smin = sj;
      int ip = lane;
      wave_argmin(smin, ip);
      ip = uni(ip);
      if (!(smin < INF)) break;
      T sip = smin;
      const T np0 = S.cn[3 * ip], np1 = S.cn[3 * ip + 1], np2 = S.cn[3 * ip + 2];
      const int o = 3 * (ip / 6);
      if (lane == 0) { S.u[iq] = 0; S.A[iq] = ip; }
      WSYNC();
      // step 2
      while (true) {
        if (++iter > prm.max_iter) { status = 1; done = true; break; }
        // d = J^T np
        T dc = 0;
        if (lane < n) dc = S.J[o * LD + lane] * np0 + S.J[(o + 1) * LD + lane] * np1 + S.J[(o + 2) * LD + lane] * np2;
        if (lane < n) S.d[lane] = dc;
        WSYNC();
        // z = J2 d2
        T zi = 0;
        if (lane < n)
          for (int c = iq; c < n; ++c) zi += S.J[lane * LD + c] * S.d[c];
        // r = R^-1 d1
        T rk = (lane < iq) ? dc : (T)0;
        for (int k = iq - 1; k >= 0; --k) {
          const T rkk = bcast(rk, k) * S.rdinv[k];
          if (lane < k) rk -= S.R[lane * LD + k] * rkk;
          if (lane == k) rk = rkk;
        }
        // step lengths
        T ratio = INF;
        if (lane < iq && rk > 0) ratio = S.u[lane] / rk;
        T t1 = ratio;
        int kmin = lane;
        wave_argmin(t1, kmin);
        kmin = uni(kmin);
        const T dn2 = wave_sum((lane >= iq && lane < n) ? dc * dc : (T)0);
        const T znp = bcast(zi, o) * np0 + bcast(zi, o + 1) * np1 + bcast(zi, o + 2) * np2;
        T t2 = INF;
        if (dn2 > (EPS * Rnorm) * (EPS * Rnorm) && znp > 0) t2 = -sip / znp;
        if (!(t1 < INF) && !(t2 < INF)) { status = 2; done = true; break; }
        const bool dual_only = !(t2 < INF);
        const bool full = !dual_only && !(t1 < t2);
        const T t = full ? t2 : t1;
        if (!dual_only && lane < n) S.x[lane] += t * zi;
        if (lane < iq) S.u[lane] -= t * rk;
        if (lane == 0) S.u[iq] += t;
        WSYNC();
        if (!full) {
          // ---- drop the blocking constraint at position kmin
          const int lcon = S.A[kmin];
          if (lane == lcon) active = false;
          for (int i = kmin; i < iq - 1; ++i)
            if (lane < n) S.R[lane * LD + i] = S.R[lane * LD + i + 1];
          int tA = 0; T tu = 0;
          if (lane >= kmin && lane < iq) { tA = S.A[lane + 1]; tu = S.u[lane + 1]; }
          WSYNC();
          if (lane >= kmin && lane < iq) { S.A[lane] = tA; S.u[lane] = tu; }
          if (lane == 0) { S.A[iq] = -1; S.u[iq] = 0; }
          --iq;
          WSYNC();
          for (int j = kmin; j < iq; ++j) {
            const T cc0 = S.R[j * LD + j], ss0 = S.R[(j + 1) * LD + j];
            const T hh = sqrt_t(cc0 * cc0 + ss0 * ss0);
            if (hh == 0) continue;
            const T ih = (T)1 / hh;
            const T cc = cc0 * ih, ss = ss0 * ih;
            WSYNC();
            if (lane >= j && lane < iq) {
              const T a1 = S.R[j * LD + lane], a2 = S.R[(j + 1) * LD + lane];
              S.R[j * LD + lane] = (lane == j) ? hh : cc * a1 + ss * a2;
              S.R[(j + 1) * LD + lane] = (lane == j) ? (T)0 : -ss * a1 + cc * a2;
            }
            if (lane < n) {
              const T a1 = S.J[lane * LD + j], a2 = S.J[lane * LD + j + 1];
              S.J[lane * LD + j] = cc * a1 + ss * a2;
              S.J[lane * LD + j + 1] = -ss * a1 + cc * a2;
            }
            if (lane == 0) S.rdinv[j] = ih;
            WSYNC();
          }
          if (!dual_only) {  // partial step: re-evaluate the candidate's slack at the new x
            T v = 0;
            if (lane == ip) v = c0 * S.x[3 * cslot] + c1 * S.x[3 * cslot + 1] + c2 * S.x[3 * cslot + 2] - crhs;
            sip = bcast(v, ip);
          }
          continue;
        }
        // ---- full step: add constraint ip with one Householder reflection of J[:, iq..n)
        {
          const T a0 = bcast(dc, iq);
          const T nr = sqrt_t(dn2);
          const T sg = (a0 >= 0) ? (T)1 : (T)-1;
          const T beta = (T)1 / (nr * (nr + fabs_t(a0)));
          if (lane >= iq && lane < n) S.w[lane] = (lane == iq) ? a0 + sg * nr : dc;
          if (lane < iq) S.R[lane * LD + iq] = dc;
          if (lane == iq) { S.R[iq * LD + iq] = -sg * nr; S.rdinv[iq] = (T)-1 / (sg * nr); }
          T yi = 0;
          if (lane < n) yi = (zi + sg * nr * S.J[lane * LD + iq]) * beta;
          WSYNC();
          if (lane < n)
            for (int c = iq; c < n; ++c) S.J[lane * LD + c] -= yi * S.w[c];
          Rnorm = (nr > Rnorm) ? nr : Rnorm;
          if (lane == ip) active = true;
          ++iq;
          WSYNC();
        }
        break;  // back to step 1
      }
    }
  }
  WSYNC();

  // ------------------------------------------------------------------ outputs: f, tau (a9), status
  T fe = 0;
  if (lane < 12) {
    const int foot = lane / 3, comp = lane - 3 * foot;
    if ((mask >> foot) & 1) {
      const int slot = __popc(mask & ((1 << foot) - 1));
      fe = S.x[3 * slot + comp];
    }
    a.f[(size_t)lane * N + s] = fe;
    S.t[lane] = fe;
  }
  WSYNC();
  if (lane < 12) {
    const int leg = lane / 3, k = lane - 3 * leg;
    T tq = S.in[WS_TAUP + lane];
#pragma unroll
    for (int mm = 0; mm < 3; ++mm) tq -= S.in[WS_JCL + 9 * leg + 3 * mm + k] * S.t[3 * leg + mm];
    a.tau[(size_t)jmap.j[lane] * N + s] = tq;
  }
  if (lane == 0) {
    a.status[s] = status;
    if (a.iters) a.iters[s] = iter;
  }
}

}  // namespace wbc

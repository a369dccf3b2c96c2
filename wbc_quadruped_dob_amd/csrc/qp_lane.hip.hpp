// GRF QP + torque map, ONE STATE PER LANE (round 2): SURVEY.md 8(a) units a7-a9 for large batches.
//
// qp_group16.hip.hpp solves the QP with a dense dual active-set method whose 12x12 factor J is spread over a 16-lane row
// (four QPs per wavefront; 12 of 16 lanes carry data, a third of the instructions are predication of four independent
// rows): about 510 vector instructions per QP.  The structure that section 4.2 of docs/DESIGN_R04.md uses for the initial factor goes
// further: with e = B f - beta (the residual wrench, 6 numbers) the problem
//     min 1/2 alpha |f|^2 + 1/2 |B f - beta|^2   s.t.  f_k in K_k  (friction pyramid and normal-force box of stance foot k)
// separates per foot once e is known,  f_k(e) = Proj_{K_k}(-B_k^T e / alpha)  (Euclidean projection of a 3-vector onto a
// pyramid frustum: closed form in the foot's contact frame), and e solves the piecewise-linear, strongly monotone equation
//     F(e) = e + beta - sum_k B_k Proj_{K_k}(-B_k^T e / alpha) = 0,        F = grad psi,  psi convex.
// Semismooth Newton on F = primal-dual active set on the QP: with the faces the projections sit on (P_k = projector onto
// the face's tangent space, f_k^p = its offset) the Newton iterate solves the 6x6 SPD system
//     (alpha I + sum_k B_k P_k B_k^T) e+ = alpha (sum_k B_k f_k^p - beta),
// i.e. the same matrix family as G of section 4.2.  Every lane runs this for ITS state -- no cross-lane traffic,
// 64 QPs per wavefront, loads and stores coalesced.  Undamped the iteration can cycle (under-determined stances): a lane that
// has not converged when its wavefront stops iterating (policy below) hands its state to the dense active-set kernel through a
// list (a few per cent of the bench data; NaN inputs end up there too and get their status from that kernel).  At
// convergence the faces are consistent with the multipliers, so the result is the QP's unique solution: same f, tau as the
// oracle to rounding (1e-12 in fp64).  status = 0, iters = Newton iterations (<= QPL_MAX_NEWTON) for the states solved here.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_types.hpp"
#include "qp_group16.hip.hpp"   // rsqrt_nr / rcp_nr, Lim, WBC_DEV

namespace wbc {

// Iteration policy.  Every wavefront runs QPL_MIN_NEWTON iterations (fewer if all its lanes are done) and goes on, up to
// QPL_MAX_NEWTON, only while at least QPL_MORE_LANES of its 64 lanes are still unconverged AND the last iteration finished
// at least 40 % of the lanes it started with: one more iteration costs the whole wavefront ~4 ns (of the device's time), a
// state handed to the dense kernel ~2.8 ns.  On 4-contact data about half of the stragglers converge in the next iteration
// (unconverged lanes after iteration 2 / 3 / 4 / 5: 28 / 9.4 / 3.9 / 3.9 %); stances with swing feet leave ~20 % of the
// lanes in a DIVERGING face cycle (alpha = 1e-3 against O(1) entries of B^T B: a face set with few free directions makes the
// full step huge) that no further full step ends (25 / 19.8 / 18.9 / 18.8 %) -- the dense kernel solves those in 2-5 of its
// iterations.  Full steps, no line search: measured on MI355X (per-lane kernel + dense kernel over the
// hand-over list, us, N = 262 144, fp64 configs[1] / fp64 observer-on / fp32 configs[3] data; dense kernel alone 347 / 225 / 144):
//   fixed 3 iterations, 2 line-search steps   116 + 66 | 110 + 29 |  93 + 28
//   fixed 3, 1 step                           107 + 68 | 101 + 32 |  86 + 30
//   fixed 3, none                             110 + 71 | 100 + 35 |  85 + 34
//   fixed 4, 2 steps                          132 + 33 | 124 + 25 | 109 + 26
//   fixed 4, none                             109 + 40 | 113 + 34 |  96 + 34
//   3 ... 5 while >= 4 lanes, 1 step          121 + 36 | 109 + 31 |  93 + 29
//   3 ... 5 while >= 4 lanes, none            109 + 42 | 105 + 34 |  90 + 33
//   3 ... 5 while >= 6 lanes, none            104 + 54 | 100 + 35 |  84 + 33
// A bracketing line search on phi'(t) = F(e + t dir) . dir rescues a few per cent of the stragglers and costs every wavefront
// an evaluation per step: it does not pay.  (An undamped semismooth Newton iteration can cycle on under-determined stances;
// such a lane simply ends up in the hand-over list.)
constexpr int QPL_MAX_NEWTON = 5;
constexpr int QPL_MIN_NEWTON = 3;
constexpr int QPL_MORE_LANES = 5;

constexpr int QPL_F64_WAVES = 2;
constexpr int QPL_F32_WAVES = 2;
constexpr int QPL_UNROLL_EVAL = 1;
constexpr int QPL_UNROLL_NEWTON = 1;
// threads per workgroup: measured 64 / 128 / 256 -- equal at 262 144 states (100-105 us); at 65 536 the 256-thread workgroups land one
// wavefront on every SIMD (36 us), the smaller ones double up on some CUs and leave others idle (44-45 us)
constexpr int QPL_WG = 256;   // threads per workgroup (LDS below: 72 kB fp64, 40 kB fp32)
constexpr int QPL_FREE = 1 | (1 << 2) | (1 << 4);   // no face active

// The contact frames live in LDS, one slot per lane ([component][foot][lane]: conflict-free), and the loops over the feet are
// NOT unrolled: kept in registers the frames (52 values) plus the temporaries of four interleaved feet need ~430 registers --
// one wavefront per SIMD, nothing to hide a dependent fp64 chain behind.  From LDS, one foot at a time, the kernel fits 256
// registers and two wavefronts share a SIMD.  Stored per foot: unit normal, lever arm (zero for a swing foot), mu, the
// normalisation of the first tangent (the tangent is rebuilt from the normal: 6 flops) and fp32 seeds of 1/(1+mu^2), 1/(1+2mu^2)
// (two Newton steps make them exact doubles again).
template <class T> struct QplLds {
  T n[3][4][QPL_WG], d[3][4][QPL_WG], m[4][QPL_WG], it[4][QPL_WG];
  float i12[2][4][QPL_WG];
};
template <class T> struct QplFrame { T n[3], t1[3], t2[3], d[3], m, i1, i2; };

template <class T> WBC_DEV void qpl_cross(const T* a, const T* b, T* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
WBC_DEV double qpl_refine(float seed, double den) {
  double y = (double)seed;
  double r = fma(-den, y, 1.0); y = fma(y, r, y);
  r = fma(-den, y, 1.0); y = fma(y, r, y);
  return y;
}
WBC_DEV float qpl_refine(float seed, float) { return seed; }

// first tangent of the contact frame from the unit normal (the oracle's convention: e_x, or e_y for a normal along x, made
// orthogonal to n); `it` = 1 / its length before normalisation
template <class T> WBC_DEV void qpl_tangent(const T* n, T it, T* t1) {
  const bool usex = fabs_t(n[0]) < (T)0.9;
  const T rx = usex ? (T)1 : (T)0, ry = usex ? (T)0 : (T)1;
  const T rd = rx * n[0] + ry * n[1];
  t1[0] = (rx - n[0] * rd) * it; t1[1] = (ry - n[1] * rd) * it; t1[2] = (-n[2] * rd) * it;
}

template <class T> WBC_DEV void qpl_frame(const QplLds<T>& L, int k, unsigned tid, QplFrame<T>& q) {
#pragma unroll
  for (int c = 0; c < 3; ++c) { q.n[c] = L.n[c][k][tid]; q.d[c] = L.d[c][k][tid]; }
  q.m = L.m[k][tid];
  qpl_tangent(q.n, L.it[k][tid], q.t1);
  qpl_cross(q.n, q.t1, q.t2);
  q.i1 = qpl_refine(L.i12[0][k][tid], (T)1 + q.m * q.m);
  q.i2 = qpl_refine(L.i12[1][k][tid], (T)1 + (T)2 * q.m * q.m);
}

// projection of (a0, b0, c0) onto {|a| <= m c, |b| <= m c, fmin <= c <= fmax}; face code = (sa+1) | (sb+1) << 2 | (sc+1) << 4
WBC_DEV double qpl_max(double a, double b) { return __builtin_fmax(a, b); }
WBC_DEV float qpl_max(float a, float b) { return __builtin_fmaxf(a, b); }
WBC_DEV double qpl_min(double a, double b) { return __builtin_fmin(a, b); }
WBC_DEV float qpl_min(float a, float b) { return __builtin_fminf(a, b); }
// (v_max / v_min: one instruction where a compare-and-select on a double takes three.  They drop a NaN operand instead of
// passing it on -- harmless here: a NaN input reaches F through e and beta as well, and F decides convergence.)
template <class T> WBC_DEV int qpl_project(const QplFrame<T>& k, T a0, T b0, T c0, T fmin, T fmax, T* loc) {
  const T A = fabs_t(a0), B = fabs_t(b0);
  const T hi = qpl_max(A, B), lo = qpl_min(A, B);
  const T c_one = (c0 + k.m * hi) * k.i1, c_two = (c0 + k.m * (hi + lo)) * k.i2;
  const T cu = (k.m * c0 >= hi) ? c0 : ((k.m * c_one >= lo) ? c_one : c_two);
  const T c = qpl_min(qpl_max(cu, fmin), fmax);
  const int sc = cu > fmax ? 1 : (cu < fmin ? -1 : 0);
  const T lim = k.m * c;
  const bool apex = !(lim > (T)0);
  const T a = qpl_max(qpl_min(a0, lim), -lim), b = qpl_max(qpl_min(b0, lim), -lim);
  const int sa = apex ? 1 : (A > lim ? (a0 > 0 ? 1 : -1) : 0), sb = apex ? 1 : (B > lim ? (b0 > 0 ? 1 : -1) : 0);
  loc[0] = a; loc[1] = b; loc[2] = c;
  return (sa + 1) | ((sb + 1) << 2) | ((sc + 1) << 4);
}

// F(e) and the faces of the projections behind it (6 bits per foot in `codes`).  FW: also leaves the forces (world frame)
// in the LDS slots of the normals -- the frame of a foot is dead once its force is known -- for the torque map.
template <class T, bool FW>
WBC_DEV void qpl_eval(QplLds<T>& L, unsigned tid, int mask, const T* sS, T ralpha, T fmin, T fmax, const T* beta, const T* e, int& codes, T* F) {
  T sf[3] = {0, 0, 0}, sm[3] = {0, 0, 0};   // sum of forces, sum of d x f
  const T ef[3] = {sS[0] * e[0], sS[1] * e[1], sS[2] * e[2]}, em[3] = {sS[3] * e[3], sS[4] * e[4], sS[5] * e[5]};
  int cds = 0;
#pragma unroll QPL_UNROLL_EVAL
  for (int k = 0; k < 4; ++k) {
    QplFrame<T> q;
    qpl_frame(L, k, tid, q);
    T cx[3];
    qpl_cross(em, q.d, cx);                               // B_k^T e = s_f e_f + (s_m e_m) x d_k
    const T y0 = -(ef[0] + cx[0]) * ralpha, y1 = -(ef[1] + cx[1]) * ralpha, y2 = -(ef[2] + cx[2]) * ralpha;
    const T a0 = q.t1[0] * y0 + q.t1[1] * y1 + q.t1[2] * y2, b0 = q.t2[0] * y0 + q.t2[1] * y1 + q.t2[2] * y2, c0 = q.n[0] * y0 + q.n[1] * y1 + q.n[2] * y2;
    T loc[3];
    const int cd = qpl_project(q, a0, b0, c0, fmin, fmax, loc);
    const bool on = (mask >> k) & 1;
    cds |= (on ? cd : QPL_FREE) << (6 * k);               // a swing foot has no faces
    T fw[3], dxf[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { fw[c] = loc[0] * q.t1[c] + loc[1] * q.t2[c] + loc[2] * q.n[c]; fw[c] = on ? fw[c] : (T)0; }
    qpl_cross(q.d, fw, dxf);
#pragma unroll
    for (int c = 0; c < 3; ++c) { sf[c] += fw[c]; sm[c] += dxf[c]; if (FW) L.n[c][k][tid] = fw[c]; }
  }
  codes = cds;
#pragma unroll
  for (int c = 0; c < 3; ++c) { F[c] = e[c] + beta[c] - sS[c] * sf[c]; F[3 + c] = e[3 + c] + beta[3 + c] - sS[3 + c] * sm[c]; }
}

// Newton iterate for the faces in `codes`: solves (alpha I + sum B_k P_k B_k^T) eN = alpha (sum B_k f_k^p - beta)
template <class T>
WBC_DEV void qpl_newton(const QplLds<T>& L, unsigned tid, int mask, const T* sS, T alpha, T fmin, T fmax, const T* beta, int codes, T* eN) {
  // lower triangle of the 6x6 matrix, force rows 0..2, moment rows 3..5 (unscaled; S^(1/2) is applied at the end)
  T Pff[6] = {0, 0, 0, 0, 0, 0};       // sum P                (xx xy xz yy yz zz)
  T X[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // sum D P  (row-major: moment row i, force column j)
  T Y[6] = {0, 0, 0, 0, 0, 0};         // sum D P D^T
  T pf[3] = {0, 0, 0}, pm[3] = {0, 0, 0};   // sum f^p, sum d x f^p
#pragma unroll QPL_UNROLL_NEWTON
  for (int k = 0; k < 4; ++k) {
    QplFrame<T> q;
    qpl_frame(L, k, tid, q);
    const int cd = (codes >> (6 * k)) & 63;
    const int sa = (cd & 3) - 1, sb = ((cd >> 2) & 3) - 1, sc = (cd >> 4) - 1;
    const bool Aa = sa != 0, Ba = sb != 0, Ca = sc != 0;
    // tangent space of the face set: [not A] t1, [not B] t2, [not C] u = (sa m, sb m, 1) / |.| (zeros where the face is not active)
    const T ua = Aa ? (T)sa * q.m : (T)0, ub = Ba ? (T)sb * q.m : (T)0;
    const bool on = (mask >> k) & 1;
    const T iu2 = (Aa && Ba) ? q.i2 : ((Aa || Ba) ? q.i1 : (T)1);     // 1 / |u|^2: |u|^2 is 1, 1 + m^2 or 1 + 2 m^2
    const T wA = (on && !Aa) ? (T)1 : (T)0, wB = (on && !Ba) ? (T)1 : (T)0, wC = (on && !Ca) ? iu2 : (T)0;
    T uw[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) uw[c] = ua * q.t1[c] + ub * q.t2[c] + q.n[c];
    T P[6];   // xx xy xz yy yz zz
    {
      int o = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) P[o++] = wA * q.t1[i] * q.t1[j] + wB * q.t2[i] * q.t2[j] + wC * uw[i] * uw[j];
    }
    const T Pc[3][3] = {{P[0], P[1], P[2]}, {P[1], P[3], P[4]}, {P[2], P[4], P[5]}};   // columns (= rows) of P
    T Xk[3][3];   // Xk[.][j] = d x (column j of P):  D P
#pragma unroll
    for (int j = 0; j < 3; ++j) { T c_[3]; qpl_cross(q.d, Pc[j], c_); Xk[0][j] = c_[0]; Xk[1][j] = c_[1]; Xk[2][j] = c_[2]; }
#pragma unroll
    for (int i = 0; i < 6; ++i) Pff[i] += P[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) X[3 * i + j] += Xk[i][j];
    // D P D^T: row i = d x (row i of D P)
    {
      T r0[3], r1[3], r2[3];
      qpl_cross(q.d, Xk[0], r0); qpl_cross(q.d, Xk[1], r1); qpl_cross(q.d, Xk[2], r2);
      Y[0] += r0[0]; Y[1] += r0[1]; Y[2] += r0[2]; Y[3] += r1[1]; Y[4] += r1[2]; Y[5] += r2[2];
    }
    // offset of the face set: only a fixed normal force contributes (friction faces pass through the origin)
    const T cbar = (Ca && on) ? (sc > 0 ? fmax : fmin) : (T)0;
    const T fa = Aa ? (T)sa * q.m * cbar : (T)0, fb = Ba ? (T)sb * q.m * cbar : (T)0;
    T fpw[3], dxf[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) fpw[c] = fa * q.t1[c] + fb * q.t2[c] + cbar * q.n[c];
    qpl_cross(q.d, fpw, dxf);
#pragma unroll
    for (int c = 0; c < 3; ++c) { pf[c] += fpw[c]; pm[c] += dxf[c]; }
  }
  // G (lower triangle, row-major packed: 00 | 10 11 | 20 21 22 | 30 .. 33 | 40 .. 44 | 50 .. 55)
  T G[21];
  G[0] = alpha + sS[0] * sS[0] * Pff[0];
  G[1] = sS[1] * sS[0] * Pff[1]; G[2] = alpha + sS[1] * sS[1] * Pff[3];
  G[3] = sS[2] * sS[0] * Pff[2]; G[4] = sS[2] * sS[1] * Pff[4]; G[5] = alpha + sS[2] * sS[2] * Pff[5];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int r = 3 + i, o = r * (r + 1) / 2;
#pragma unroll
    for (int j = 0; j < 3; ++j) G[o + j] = sS[3 + i] * sS[j] * X[3 * i + j];
  }
  G[6 + 3] = alpha + sS[3] * sS[3] * Y[0];
  G[10 + 3] = sS[4] * sS[3] * Y[1]; G[10 + 4] = alpha + sS[4] * sS[4] * Y[3];
  G[15 + 3] = sS[5] * sS[3] * Y[2]; G[15 + 4] = sS[5] * sS[4] * Y[4]; G[15 + 5] = alpha + sS[5] * sS[5] * Y[5];
  T rhs[6];
#pragma unroll
  for (int c = 0; c < 3; ++c) { rhs[c] = alpha * (sS[c] * pf[c] - beta[c]); rhs[3 + c] = alpha * (sS[3 + c] * pm[c] - beta[3 + c]); }
  // dense 6x6 Cholesky in place (L overwrites G), reciprocal pivots in il
  T il[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int oj = j * (j + 1) / 2;
    T piv = G[oj + j];
#pragma unroll
    for (int k = 0; k < j; ++k) piv -= G[oj + k] * G[oj + k];
    il[j] = rsqrt_nr(piv);
    G[oj + j] = piv * il[j];
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      const int oi = i * (i + 1) / 2;
      T v = G[oi + j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= G[oi + k] * G[oj + k];
      G[oi + j] = v * il[j];
    }
  }
  T w[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int oi = i * (i + 1) / 2;
    T v = rhs[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= G[oi + k] * w[k];
    w[i] = v * il[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    T v = w[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) v -= G[k * (k + 1) / 2 + i] * eN[k];
    eN[i] = v * il[i];
  }
}

// The faces of the four projections <-> the public active-set word (include/wbc_hip.h, wbc_step_batch_warm): per foot k
//   sa = +1 (a clipped at +mu c: (mu n - t1) . f = 0)  <->  bit 4k + 0        sa = -1 ((mu n + t1) . f = 0)  <->  bit 16 + 4k + 0
//   sb = +1                                             <->  bit 4k + 1        sb = -1                         <->  bit 16 + 4k + 1
//   sc = -1 (n . f = fn_min)                            <->  bit 4k + 2        sc = +1 (n . f = fn_max)        <->  bit 4k + 3
WBC_DEV int qpl_codes_from_aset(int aset, int mask) {
  int codes = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int A = (aset >> (4 * k)) & 0xF, B = (aset >> (16 + 4 * k)) & 0x3;
    const int sa = (A & 1) ? 1 : ((B & 1) ? -1 : 0), sb = (A & 2) ? 1 : ((B & 2) ? -1 : 0), sc = (A & 8) ? 1 : ((A & 4) ? -1 : 0);
    const int cd = (sa + 1) | ((sb + 1) << 2) | ((sc + 1) << 4);
    codes |= (((mask >> k) & 1) ? cd : QPL_FREE) << (6 * k);
  }
  return codes;
}
WBC_DEV int qpl_aset_from_codes(int codes, int mask) {
  int aset = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cd = (codes >> (6 * k)) & 63;
    const int sa = (cd & 3) - 1, sb = ((cd >> 2) & 3) - 1, sc = (cd >> 4) - 1;
    int A = (sa > 0 ? 1 : 0) | (sb > 0 ? 2 : 0) | (sc < 0 ? 4 : 0) | (sc > 0 ? 8 : 0);
    int B = (sa < 0 ? 1 : 0) | (sb < 0 ? 2 : 0);
    if (!((mask >> k) & 1)) { A = 0; B = 0; }
    aset |= (A << (4 * k)) | (B << (16 + 4 * k));
  }
  return aset;
}

// todo[0] = number of states handed to the dense kernel (zeroed by the front-half kernel of the same tick: SweepArgs::qp_todo),
// todo[2] = that number of the last tick (diagnostics), todo[4 ...] = their indices
constexpr int QPL_WARM_MIN_NEWTON = 1;
// WARM (wbc_step_batch_warm at large batches: dependent ticks): the Newton iteration starts from the faces of a.aset_in instead of from "all free".
// Any face set is a valid starting iterate of the semismooth Newton method -- a wrong or stale guess costs iterations (or sends the state to the
// hand-over list), never the solution -- and the right one is confirmed by ONE Newton step that leaves the faces unchanged; a wavefront then
// goes on only while >= QPL_MORE_LANES of its lanes are unfinished (no minimum of three iterations: most lanes are done at once; a minimum
// of two was measured: a third fewer states handed over, the kernel 2-3 us longer, the tick no faster).
template <class T, bool RHAT, bool WARM = false>
__global__ __launch_bounds__(QPL_WG, (sizeof(T) == 4 ? QPL_F32_WAVES : QPL_F64_WAVES)) void qp_lane_kernel(DevParams<T> prm, QpArgs<T> a, QpJidx jmap, int* __restrict__ todo) {
  __shared__ QplLds<T> L;
  const unsigned tid = threadIdx.x;
  const size_t N = a.N;
  const unsigned N32 = (unsigned)N;
  const size_t s_raw = (size_t)blockIdx.x * QPL_WG + tid;
  const bool live = s_raw < N;
  const unsigned s32 = (unsigned)(live ? s_raw : N - 1);
#define LLD(ptr, comp) (*(const T*)((const char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))))
#define LST(ptr, comp, val) (*(T*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))) = (val))
  const int mask = a.mask[s32] & 0xF;
  const bool geom_jc = a.Jc != nullptr;
  // every input of the solve is requested before the first one is used: a load that waits exposes its whole latency
  T in_d[12], in_n[12], in_mu[4], in_b[6], in_r[6];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (geom_jc) { in_d[3 * k] = LLD(a.Jc, (3 * k + 1) * 18 + 5); in_d[3 * k + 1] = LLD(a.Jc, (3 * k + 2) * 18 + 3); in_d[3 * k + 2] = LLD(a.Jc, (3 * k) * 18 + 4); }
    else { in_d[3 * k] = LLD(a.ws, WS_D + 3 * k); in_d[3 * k + 1] = LLD(a.ws, WS_D + 3 * k + 1); in_d[3 * k + 2] = LLD(a.ws, WS_D + 3 * k + 2); }
  }
#pragma unroll
  for (int c = 0; c < 12; ++c) in_n[c] = LLD(a.normals, c);
#pragma unroll
  for (int k = 0; k < 4; ++k) in_mu[k] = LLD(a.mu, k);
#pragma unroll
  for (int c = 0; c < 6; ++c) { in_b[c] = a.wdes ? LLD(a.wdes, c) : LLD(a.ws, WS_B + c); in_r[c] = RHAT ? LLD(a.ws, WS_RHAT + c) : (T)0; }
  // (the empty asm statements keep the compiler from sinking a load down to its first use)
#pragma unroll
  for (int c = 0; c < 12; ++c) { asm volatile("" : "+v"(in_d[c])); asm volatile("" : "+v"(in_n[c])); }
#pragma unroll
  for (int c = 0; c < 6; ++c) { asm volatile("" : "+v"(in_b[c])); if (RHAT) asm volatile("" : "+v"(in_r[c])); }
#pragma unroll
  for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(in_mu[c]));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool on = (mask >> k) & 1;
    T n[3] = {in_n[3 * k], in_n[3 * k + 1], in_n[3 * k + 2]};
    const T iln = rsqrt_nr(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    n[0] *= iln; n[1] *= iln; n[2] *= iln;
    T t1[3];
    qpl_tangent(n, (T)1, t1);
    const T m = in_mu[k] * prm.mu_scale;
#pragma unroll
    for (int c = 0; c < 3; ++c) { L.n[c][k][tid] = n[c]; L.d[c][k][tid] = on ? in_d[3 * k + c] : (T)0; }
    L.m[k][tid] = m;
    L.it[k][tid] = rsqrt_nr(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
    L.i12[0][k][tid] = (float)rcp_nr((T)1 + m * m);
    L.i12[1][k][tid] = (float)rcp_nr((T)1 + (T)2 * m * m);
  }
  T sS[6], beta[6];
  T bmax = 0;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    sS[c] = prm.sS[c];
    beta[c] = sS[c] * (in_b[c] - in_r[c]);
    bmax = fabs_t(beta[c]) > bmax ? fabs_t(beta[c]) : bmax;
  }
  const T alpha = prm.alpha, ralpha = prm.rsqrt_alpha * prm.rsqrt_alpha, fmin = prm.fn_min, fmax = prm.fn_max;
  const T tolF = (std::is_same<T, double>::value ? (T)1e-11 : (T)2e-5) * ((T)1 + bmax);

  // start: all faces free (the unconstrained minimum) -- or the faces the previous tick ended on (WARM)
  int code = QPL_FREE | (QPL_FREE << 6) | (QPL_FREE << 12) | (QPL_FREE << 18);
  if constexpr (WARM) { if (a.aset_in) code = qpl_codes_from_aset(a.aset_in[s32], mask); }
  T e[6], F[6];
  qpl_newton(L, tid, mask, sS, alpha, fmin, fmax, beta, code, e);
  const int code0 = code;
  qpl_eval<T, false>(L, tid, mask, sS, ralpha, fmin, fmax, beta, e, code, F);
  int iters = 0, prev_cnt = 64;
  bool conv = false;
  constexpr int MIN_NEWTON = WARM ? QPL_WARM_MIN_NEWTON : QPL_MIN_NEWTON;
  auto fnorm = [](const T* v) __attribute__((always_inline)) -> T {
    T m = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) m = qpl_max(m, fabs_t(v[c]));   // (drops NaN components: nan6 below catches those)
    return m;
  };
  auto nan6 = [](const T* v) __attribute__((always_inline)) -> bool {   // a NaN (or an Inf - Inf) among the components
    const T t = ((v[0] + v[1]) + (v[2] + v[3])) + (v[4] + v[5]);
    return !(t - t == (T)0); };
  if constexpr (WARM) conv = code == code0 && !nan6(F);   // the Newton step from the given faces left them unchanged: it IS the solution
  // Live across an iteration: the current point e, F(e) and its faces (no forces: they are recomputed once from the final e).
#pragma unroll 1
  for (int itn = 0; itn < QPL_MAX_NEWTON; ++itn) {
    conv = conv || (fnorm(F) <= tolF && !nan6(F));
    const bool act = live && !conv;
    const unsigned long long todo_lanes = __ballot(act);
    const int todo_cnt = __popcll(todo_lanes);
    // beyond the minimum: only while enough lanes are left AND the last iteration finished at least 40 % of those it started with
    // (stances with swing feet leave ~20 % of the lanes in a diverging face cycle that no further full step ends)
    if (todo_cnt == 0 || (itn >= MIN_NEWTON && (todo_cnt < QPL_MORE_LANES || 5 * todo_cnt > 3 * prev_cnt))) break;
    prev_cnt = todo_cnt;
    iters += act ? 1 : 0;
    T eN[6], Ft[6];
    int codet;
    qpl_newton(L, tid, mask, sS, alpha, fmin, fmax, beta, code, eN);
    qpl_eval<T, false>(L, tid, mask, sS, ralpha, fmin, fmax, beta, eN, codet, Ft);
    conv = conv || (act && codet == code && !nan6(Ft));     // faces unchanged by a full Newton step: it IS the solution for those faces
#pragma unroll
    for (int c = 0; c < 6; ++c) { e[c] = act ? eN[c] : e[c]; F[c] = act ? Ft[c] : F[c]; }
    code = act ? codet : code;
  }
  conv = conv || (fnorm(F) <= tolF && !nan6(F));
  // ---- outputs of the states solved here; the others go to the dense active-set kernel
  if (live && conv) {
    T jl[36], tp[12], tr[12];   // own-leg Jacobian columns and tau_partial: all requested before the first store (same reason as above)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int jm = jmap.j[3 * k + j];
#pragma unroll
        for (int c = 0; c < 3; ++c) jl[9 * k + 3 * j + c] = geom_jc ? LLD(a.Jc, (3 * k + c) * 18 + 6 + jm) : LLD(a.ws, WS_JCL + 9 * k + 3 * c + j);
        tp[3 * k + j] = LLD(a.ws, WS_TAUP + 3 * k + j);
        tr[3 * k + j] = RHAT ? LLD(a.ws, WS_RHAT + 6 + 3 * k + j) : (T)0;
      }
    // the forces at the solution go to LDS while those loads are in flight (the frames live in LDS: there are registers to spare)
    qpl_eval<T, true>(L, tid, mask, sS, ralpha, fmin, fmax, beta, e, code, F);
#pragma unroll
    for (int c = 0; c < 36; ++c) asm volatile("" : "+v"(jl[c]));
#pragma unroll
    for (int c = 0; c < 12; ++c) { asm volatile("" : "+v"(tp[c])); if (RHAT) asm volatile("" : "+v"(tr[c])); }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const T fw[3] = {L.n[0][k][tid], L.n[1][k][tid], L.n[2][k][tid]};
#pragma unroll
      for (int c = 0; c < 3; ++c) LST(a.f, 3 * k + c, fw[c]);
#pragma unroll
      for (int j = 0; j < 3; ++j)   // tau of joint j of leg k = tau_partial - (own-leg Jacobian column) . f
        LST(a.tau, jmap.j[3 * k + j], (tp[3 * k + j] - tr[3 * k + j]) - (jl[9 * k + 3 * j] * fw[0] + jl[9 * k + 3 * j + 1] * fw[1] + jl[9 * k + 3 * j + 2] * fw[2]));
    }
    a.status[s32] = 0;
    if (a.iters) a.iters[s32] = iters;
    if (a.aset_out) a.aset_out[s32] = qpl_aset_from_codes(code, mask);
  } else if (live) {
    const int slot = atomicAdd(&todo[0], 1);
    if (slot < (int)a.N) todo[4 + slot] = (int)s32;   // (always, while the count starts a tick at zero: the guard keeps a stale count from writing past the list)
    else { a.status[s32] = 1; if (a.iters) a.iters[s32] = iters; }   // list overflow (the count did not start at zero): reported, never silent -- tau, f of this state are NOT this tick's
  }
#undef LLD
#undef LST
}

}  // namespace wbc

// GRF QP + torque map, round 3: the SAME dual active-set method (Goldfarb-Idnani, same pivoting rules, tolerances and status
// codes as qp_group16.hip.hpp and the oracle) with all its linear algebra done on the 6-dimensional WRENCH space instead of on
// 12 x 12 factors.  fp64 only (the inverse below is kept by Sherman-Morrison updates; fp32 keeps the orthogonal-factor form).
//
//     min 1/2 alpha |f|^2 + 1/2 |B f - beta|^2,   B = S^(1/2) [I ; [d_k]x] per stance foot,   s.t. per-foot pyramid + box
//
// H = alpha I + B^T B has rank-6 structure and every constraint touches ONE foot (3 variables).  With N_k the active normals of
// foot k (at most 3, linearly independent), P_k the projector onto their null space and
//     G_A = alpha I + sum_k B_k P_k B_k^T                                   (6 x 6, SPD)
// the quantities of a dual active-set iteration for the candidate normal n+ on foot kp are
//     v = P_kp n+,   b = B_kp v,   y = G_A^-1 b,   w_k = B_k^T y
//     primal step direction   z_k = (delta_{k,kp} v - P_k w_k) / alpha,        z . n+ = (|v|^2 - b . y) / alpha     ( = |d2|^2 of the dense method)
//     dual step direction     r_k = N_k^+ (delta_{k,kp} n+ - w_k)              (rate at which the active multipliers of foot k fall)
//     adding n+:   P_kp -= v v^T / |v|^2,   N_kp^+ gains the row v^T / |v|^2 (the others lose their n+ component),
//                  G_A^-1 += y y^T / (alpha z . n+)           -- Sherman-Morrison with what the step already computed
//     dropping a constraint of foot k:  P_k, N_k^+ rebuilt in closed form from the <= 2 remaining normals (cross products),
//                  G_A^-1 -= (G_A^-1 bh)(G_A^-1 bh)^T / (1 + bh . G_A^-1 bh),   bh = B_k what,  what what^T = P_k(new) - P_k(old)
// (derivation and a numpy restatement checked against the oracle iteration by iteration: tools/structured_gi.py).
//
// Why: the dense form spends ~440 vector instructions and ~3 200 cycles of dependent latency per iteration on 12 x 12 products,
// a Householder update of two copies of J and a back-substitution (round 2 stamps); this form needs one 6 x 6 matrix-vector
// product, a handful of 3-vector operations per lane and no factor update beyond a rank-one FMA per row.
//
// Mapping (same as qp_group16: one QP per 16-lane DPP row, four per wavefront; lane 4k + c of a row belongs to foot k):
//   lane 4k + c, c < 3: variable (foot k, axis c); row c of P_k; slot c of foot k's active list (constraint id, multiplier,
//                       row c of N_k^+); constraints 2 (4k + c) and 2 (4k + c) + 1 as in qp_group16
//   lane i < 6 (and i + 6, i + 12): row i of G_A^-1   (y_i = row . b, then six row broadcasts from static lanes)
//   LDS per QP: the 32 constraint normals (by id), P_k (6 words per foot) and d_k for the run-time foot index kp -- every lane
//   fetches n+, P_kp, d_kp and forms v, b redundantly, so the step needs NO cross-lane reduction at all.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_types.hpp"
#include "qp_group16.hip.hpp"

namespace wbc {

// per wave: four QPs x {32 constraint normals, P of the four feet (xx xy xz yy yz zz), lever arms of the four feet, scratch what}
template <class T> struct S16Lds { T C[4][32 * 3]; T P[4][4 * 16]; T D[4][4 * 3]; T W[4][4]; T R[4][6]; };   // P: row c of foot k at 16 k + 4 c (row 3: the spare lane's dummy); R: SPEC's r_prev (base rows)

// row gi of (alpha I + B B^T)^-1 -- the inverse of the EMPTY active set -- from the lever arms of the stance feet (cold start; SPEC: also where a
// wavefront starts over).  A function of explicit arguments, not a capturing lambda: called from two places, a closure object would sit on the stack.
template <class T>
WBC_DEV void s16_cold_inverse_row(int mask, T d_me, T alpha_l, T s0, T s1, T s2, T s3, T s4, T s5, int gi, T (&Gr)[6]) {
  T a01, a02, a10, a12, a20, a21, b10, b20, b21;
  T il[6];
  {
    T Dx[4], Dy[4], Dz[4], Of[4];
    sfor<0, 4>([&](auto fc) __attribute__((always_inline)) {
      constexpr int fj = decltype(fc)::value;
      const bool onj = (mask >> fj) & 1;
      Of[fj] = onj ? (T)1 : (T)0;
      Dx[fj] = onj ? gbc<3 * fj>(d_me) : (T)0; Dy[fj] = onj ? gbc<3 * fj + 1>(d_me) : (T)0; Dz[fj] = onj ? gbc<3 * fj + 2>(d_me) : (T)0;
    });
    const T nc = (Of[0] + Of[1]) + (Of[2] + Of[3]);
    const T sx = (Dx[0] + Dx[1]) + (Dx[2] + Dx[3]), sy = (Dy[0] + Dy[1]) + (Dy[2] + Dy[3]), sz = (Dz[0] + Dz[1]) + (Dz[2] + Dz[3]);
    T Pxx = 0, Pxy = 0, Pxz = 0, Pyy = 0, Pyz = 0, Pzz = 0;
    sfor<0, 4>([&](auto fc) __attribute__((always_inline)) {
      constexpr int fj = decltype(fc)::value;
      Pxx += Dx[fj] * Dx[fj]; Pxy += Dx[fj] * Dy[fj]; Pxz += Dx[fj] * Dz[fj];
      Pyy += Dy[fj] * Dy[fj]; Pyz += Dy[fj] * Dz[fj]; Pzz += Dz[fj] * Dz[fj];
    });
    const T g00 = alpha_l + s0 * s0 * nc, g11 = alpha_l + s1 * s1 * nc, g22 = alpha_l + s2 * s2 * nc;
    const T gm01 = -(s3 * s1) * sz, gm02 = (s3 * s2) * sy, gm10 = (s4 * s0) * sz, gm12 = -(s4 * s2) * sx, gm20 = -(s5 * s0) * sy, gm21 = (s5 * s1) * sx;
    const T m00 = alpha_l + (s3 * s3) * (Pyy + Pzz), m11 = alpha_l + (s4 * s4) * (Pxx + Pzz), m22 = alpha_l + (s5 * s5) * (Pxx + Pyy);
    const T m10 = -(s4 * s3) * Pxy, m20 = -(s5 * s3) * Pxz, m21 = -(s5 * s4) * Pyz;
    il[0] = rsqrt_nr(g00); il[1] = rsqrt_nr(g11); il[2] = rsqrt_nr(g22);
    a01 = gm01 * il[1]; a02 = gm02 * il[2]; a10 = gm10 * il[0]; a12 = gm12 * il[2]; a20 = gm20 * il[0]; a21 = gm21 * il[1];
    const T c00 = m00 - a01 * a01 - a02 * a02, c11 = m11 - a10 * a10 - a12 * a12, c22 = m22 - a20 * a20 - a21 * a21;
    const T c10 = m10 - a12 * a02, c20 = m20 - a21 * a01, c21 = m21 - a20 * a10;
    il[3] = rsqrt_nr(c00);
    b10 = c10 * il[3]; b20 = c20 * il[3];
    const T t11 = c11 - b10 * b10;
    il[4] = rsqrt_nr(t11);
    b21 = (c21 - b20 * b10) * il[4];
    const T t22 = c22 - b20 * b20 - b21 * b21;
    il[5] = rsqrt_nr(t22);
  }
  // ------------------------------------------------------------------ my row of G^-1 (row gi = l16 mod 6): G x = e_gi by the factor
  {
    const T e0 = gi == 0 ? (T)1 : (T)0, e1 = gi == 1 ? (T)1 : (T)0, e2 = gi == 2 ? (T)1 : (T)0, e3 = gi == 3 ? (T)1 : (T)0, e4 = gi == 4 ? (T)1 : (T)0,
            e5 = gi == 5 ? (T)1 : (T)0;
    T w[6];
    w[0] = e0 * il[0]; w[1] = e1 * il[1]; w[2] = e2 * il[2];
    w[3] = (e3 - a01 * w[1] - a02 * w[2]) * il[3];
    w[4] = (e4 - a10 * w[0] - a12 * w[2] - b10 * w[3]) * il[4];
    w[5] = (e5 - a20 * w[0] - a21 * w[1] - b20 * w[3] - b21 * w[4]) * il[5];
    Gr[5] = w[5] * il[5];
    Gr[4] = (w[4] - b21 * Gr[5]) * il[4];
    Gr[3] = (w[3] - b10 * Gr[4] - b20 * Gr[5]) * il[3];
    Gr[2] = (w[2] - a02 * Gr[3] - a12 * Gr[4]) * il[2];
    Gr[1] = (w[1] - a01 * Gr[3] - a21 * Gr[5]) * il[1];
    Gr[0] = (w[0] - a10 * Gr[4] - a20 * Gr[5]) * il[0];
  }
  }

// TS = the solver's scalar type (what the batch arrays and the LDS workspace hold); the arithmetic is ALWAYS double: the inverse
// is kept by rank-one updates (cond(G_A) ~ 1e4-1e5 is too much for fp32), and on gfx950 a dependent v_fma_f64 costs a lone
// wavefront what a dependent v_fma_f32 costs (13.5 vs 13.1 cycles, tools/issue_probe.hip) -- the QP is latency-, not byte-bound.
// PRE (tiles only): G^-1 and the unconstrained minimum x0 of this state were computed by the tile's predictor, one state per lane, and wait
// in LDS (who.pre: 36 + 12 doubles) -- the factorisation, the unit solves and the x0 solve below (~350 of the ~750 set-up instructions a
// wavefront spends per four states) are skipped.
// WARM (dependent ticks: rollouts, closed loops): the iteration starts from a GIVEN active set instead of from the unconstrained minimum.
//   1: the set of each state comes from a.aset_in (null: cold), 2: from *carry, an LDS word per state that the persistent rollout kernel keeps
//   across its ticks; the final set always goes back to *carry (2) and to a.aset_out (when given, every instantiation).
// Encoding (include/wbc_hip.h): bit l16 = constraint A of lane l16 (lane 4k + j: mu~ n - t1, mu~ n - t2, n, -n of foot k), bit 16 + l16 =
// constraint B of lane l16 (j < 2: mu~ n + t1, mu~ n + t2) -- what the per-lane flags actA / actB are, read off two ballots.
// The block set-up (derivation and numpy restatement: tools/structured_gi.py, warm_setup): per foot the <= 3 given normals give P_k and
// N_k^+ in closed form (cross products, ONE reciprocal); every lane assembles G_A = alpha I + sum_k B_k P_k B_k^T (21 entries) from the
// four P_k in LDS, factors it (general 6 x 6 Cholesky) and solves for its row of G_A^-1; the minimiser ON the set and its multipliers are
//     f_k = f_k^p - P_k B_k^T y,   f_k^p = N_k^+T rhs_k,   G_A y = sum_k B_k f_k^p - S^(1/2) b,   u_k = alpha N_k^+ (f_k + B_k^T y).
// A set with more than three rows on a foot, dependent rows or a negative multiplier is no S-pair of the dual method: that row of the
// wavefront repeats the set-up with the empty set (= the cold start; the loop below runs at most twice).  The QP is strictly convex, so
// the start changes the iteration count, never the solution.
template <class TS, bool WSLDS, bool RHAT = false, int SPW = 16, bool TILED = false, int WPB = (WSLDS || TILED) ? 4 : 1, class Idle = QpNoIdle, bool PRE = false, int WARM = 0, int STG = 0>
WBC_DEV void qp_struct16_body(const DevParams<TS>& prm, const QpArgs<TS>& a, const QpJidx& jmap, const TS* wsl, const QpSync* sync = nullptr,
                              const QpWho who = QpWho{0, false}, Idle idle = Idle(), int* carry = nullptr) {
  using T = double;
  static_assert(!(WSLDS && TILED), "tiles are dealt by the stand-alone kernel only");
  static_assert(STG == 0 || (TILED && WARM == 0 && !RHAT), "staged tiles: cold, rhat folded into the image");
  // SPEC (fused / rollout kernels with the observer on, cold start): the target wrench b = w_des - rhat_base is the last input to arrive -- the observer
  // role's base rows end at about +6.2 us, the QP's factor is done at +3.6.  The iteration therefore STARTS on b~ = w_des - r_prev (the observer state as
  // the tick finds it: an input, one filter step away from rhat; QpArgs::rprev, null = opt out) and, when rhat is there, moves the solution to b on the
// active set it has reached: with
  // beta -> beta + d the minimiser on a fixed set moves by  dy = -G_A^-1 d,  df_k = -P_k B_k^T dy,  du_k = alpha N_k^+ (df_k + B_k^T dy)  -- one product with
  // the inverse the loop maintains.  Multipliers still >= 0: an S-pair for b, the loop goes on from it (usually it has nothing left to do).  A negative
  // multiplier (a row that the last filter step releases): that row of the wavefront starts over from the empty set with b itself.  The QP is
  // strictly convex, so f, tau, status do not depend on b~ at all -- a torn or already updated read of r_prev only makes the guess better or worse.
  constexpr bool SPEC = WSLDS && RHAT && WARM == 0 && !PRE && !TILED;
  static_assert(!(WARM != 0 && PRE), "warm starts set their blocks up themselves");
  unsigned tx = threadIdx.x;
  asm volatile("" : "+v"(tx));   // lane-derived predicates stay inside this call (see WBC_LAUNDERED_TID, dyn_split.hip.hpp)
  const int lane = tx & 63;
  const int l16 = lane & 15;
  const int rowbase = lane & 48;
  const int grp = lane >> 4;
  const int f = l16 >> 2, c3 = l16 & 3;
  const bool isvar = c3 < 3;
  const int v = 3 * f + (isvar ? c3 : 0);
  S16Lds<T>* Lp;
  if constexpr (STG != 0) Lp = (S16Lds<T>*)who.tab;     // (staged tiles: this wavefront's tables live in the kernel's dynamic LDS, beside the image)
  else { __shared__ S16Lds<T> lds_all[WPB]; Lp = &lds_all[tx >> 6]; }
  S16Lds<T>& L = *Lp;
  T* Cl = L.C[grp];
  T* Pl = L.P[grp];
  T* Dl = L.D[grp];
  T* Wl = L.W[grp];
  const size_t N = a.N;
  const unsigned N32 = (unsigned)N;
  size_t wg = blockIdx.x;
  if (WPB == 1 && (gridDim.x & 31) == 0) wg = (wg & ~(size_t)31) + ((wg & 7) << 2) + ((wg >> 3) & 3);   // XCD-aware (qp_group16.hip.hpp)
  static_assert(SPW == 16 || WSLDS, "fewer states per workgroup only inside the fused kernels");
  const size_t qp_raw = TILED ? who.state : WSLDS ? (size_t)blockIdx.x * SPW + (tx >> 4) : (wg * blockDim.x + tx) >> 4;
  bool live = TILED ? who.live : (qp_raw < N && (SPW == 16 || (int)(tx >> 4) < SPW));
  unsigned s32 = (unsigned)(live ? qp_raw : N - 1);
  const T INF = Lim<T>::inf, EPS = Lim<T>::eps;
#define GLD(ptr, comp) ((T)(*(const TS*)((const char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(TS)))))
#define WSLD(comp) (WSLDS ? (T)wsl[(comp) * 16 + (int)(tx >> 4)] : GLD(a.ws, comp))
#define BLD(c) (WSLDS ? (T)wsl[(WS_B + (c)) * 16 + (int)(tx >> 4)] : (a.wdes ? GLD(a.wdes, c) : GLD(a.ws, WS_B + (c))))
#define GST(ptr, comp, val) (*(TS*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(TS))) = (TS)(val))
  const int stg_slot = (STG != 0 && live) ? who.slot : 0;
#define IMG(comp) (((TS*)who.img)[(comp) * who.stride + stg_slot])

#ifdef WBC_QP_STAMP
  const long long st_t0 = __builtin_readcyclecounter();
#endif
  // ------------------------------------------------------------------ inputs
  int mask;
  T n_ld, mu_f;
  if constexpr (STG != 0) { mask = who.iimg[stg_slot] & 0xF; n_ld = isvar ? (T)IMG(ST_N + v) : (T)0; mu_f = (T)IMG(ST_MU + f); }
  else { mask = a.mask[s32] & 0xF; n_ld = isvar ? GLD(a.normals, v) : (T)0; mu_f = GLD(a.mu, f); }
  bool on = (mask >> f) & 1;
  const bool geom_jc = STG == 0 && !WSLDS && a.Jc != nullptr;
  T rprev_in = 0;
  if constexpr (SPEC) { if (a.rprev && l16 < 6) rprev_in = GLD(a.rprev, l16); }   // (parked in LDS once b~ is formed: read back behind the iteration)
  int aset = 0;
  if constexpr (WARM == 1) { if (a.aset_in) aset = a.aset_in[s32]; }
  if constexpr (WARM == 2) aset = *carry;
  WBC_QSTAMP(1);
  idle();
  // ------------------------------------------------------------------ my constraints (friction pyramid, force box): as in qp_group16_body.
  // They need the normals and mu only: the kernels that wait for the lever arms (WSLDS: one-launch tick, rollouts) form them in front of that wait
  // (13.3 -> 13.2 us at 4 096 states; -DWBC_QP_CONS_EARLY=0: behind the factor, as the stand-alone kernels keep it -- there the early form costs 0.8 %)
  constexpr bool CONS_EARLY = WSLDS && WARM == 0;   // (the warm observer-on tick sits at 255 registers: the rows held across the wait spill there)
  T cAx, cAy, cAz, rA, cBx = 0, cBy = 0, cBz = 0;
  const bool hasB = c3 < 2;
#define WBC_QP_FORM_ROWS do { \
    T nx = dppx<0x00>(n_ld), ny = dppx<0x55>(n_ld), nz = dppx<0xAA>(n_ld); \
    const T iln = rsqrt_nr(nx * nx + ny * ny + nz * nz); \
    nx *= iln; ny *= iln; nz *= iln; \
    const bool usex = fabs_t(nx) < (T)0.9; \
    const T rx = usex ? (T)1 : (T)0, ry = usex ? (T)0 : (T)1; \
    const T rd = rx * nx + ry * ny; \
    T t1x = rx - nx * rd, t1y = ry - ny * rd, t1z = -nz * rd; \
    const T it = rsqrt_nr(t1x * t1x + t1y * t1y + t1z * t1z); \
    t1x *= it; t1y *= it; t1z *= it; \
    const T t2x = ny * t1z - nz * t1y, t2y = nz * t1x - nx * t1z, t2z = nx * t1y - ny * t1x; \
    const T mt = mu_f * prm.mu_scale; \
    const T ttx = (c3 == 0) ? t1x : t2x, tty = (c3 == 0) ? t1y : t2y, ttz = (c3 == 0) ? t1z : t2z; \
    if (hasB) { \
      cAx = mt * nx - ttx; cAy = mt * ny - tty; cAz = mt * nz - ttz; rA = 0; \
      cBx = mt * nx + ttx; cBy = mt * ny + tty; cBz = mt * nz + ttz; \
    } else if (c3 == 2) { cAx = nx; cAy = ny; cAz = nz; rA = prm.fn_min; } \
    else { cAx = -nx; cAy = -ny; cAz = -nz; rA = -prm.fn_max; } \
  } while (0)
  if constexpr (CONS_EARLY) WBC_QP_FORM_ROWS;
  if constexpr (WSLDS) { if (sync) qp_wait(sync->geom, sync->need_geom); }
  WBC_QSTAMP(2);
  T d_me = 0;
  if constexpr (STG != 0) { if (isvar) d_me = (T)IMG(ST_D + v); }
  else if (geom_jc) {
    const int comp = c3 == 0 ? (3 * f + 1) * 18 + 5 : (c3 == 1 ? (3 * f + 2) * 18 + 3 : (3 * f) * 18 + 4);
    if (isvar) d_me = GLD(a.Jc, comp);
  } else if (isvar) d_me = WSLD(WS_D + v);

  // ------------------------------------------------------------------ G = alpha I + B B^T and its factor (as in qp_group16_body)
  const T onf = on ? (T)1 : (T)0;
  const T dqx = onf * dppx<0x00>(d_me), dqy = onf * dppx<0x55>(d_me), dqz = onf * dppx<0xAA>(d_me);   // my foot's lever arm, zero for a swing foot
  // (kernel-uniform weights pass through an empty asm so that products of two of them are not hoisted out of the tile / horizon loop and held
  //  in registers for the whole kernel.  Tiles launder the raw values as SCALARS, before any conversion: sixteen vector registers less)
  TS r0 = prm.sS[0], r1 = prm.sS[1], r2 = prm.sS[2], r3 = prm.sS[3], r4 = prm.sS[4], r5 = prm.sS[5], ra = prm.alpha, rq = prm.rsqrt_alpha;
  if constexpr (TILED) asm volatile("" : "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3), "+s"(r4), "+s"(r5), "+s"(ra), "+s"(rq));
  T s0 = r0, s1 = r1, s2 = r2, s3 = r3, s4 = r4, s5 = r5;
  T alpha_l = ra, ralpha = (T)rq * (T)rq;
  if constexpr (WSLDS) asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(alpha_l), "+v"(ralpha));
  const int gi = l16 < 6 ? l16 : (l16 < 12 ? l16 - 6 : l16 - 12);
  T Gr[6];
  if constexpr (WARM != 0) {
    sfor<0, 6>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; Gr[j] = 0; });   // (set by the block set-up below)
  } else if constexpr (PRE) {
    sfor<0, 6>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; Gr[j] = who.pre[6 * gi + j]; });
  }
  if constexpr (WARM == 0 && !PRE) s16_cold_inverse_row<T>(mask, d_me, alpha_l, s0, s1, s2, s3, s4, s5, gi, Gr);
  // y = G_A^-1 b for a row-uniform b: my component from my row, then six broadcasts from the static lanes 0..5
  T y[6];
  auto ginv_mul = [&](const T* bb, T& yi) __attribute__((always_inline)) {
    yi = ((Gr[0] * bb[0] + Gr[1] * bb[1]) + (Gr[2] * bb[2] + Gr[3] * bb[3])) + (Gr[4] * bb[4] + Gr[5] * bb[5]);
    sfor<0, 6>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; y[j] = dppx<0x150 + j>(yi); });
  };
  // w_k = B_k^T y for my own foot (zero for a swing foot: dq* and onf are zero there)
  auto bt_y = [&](T& w0, T& w1, T& w2) __attribute__((always_inline)) {
    const T m0 = s3 * y[3], m1 = s4 * y[4], m2 = s5 * y[5];
    w0 = onf * (s0 * y[0]) + (m1 * dqz - m2 * dqy);
    w1 = onf * (s1 * y[1]) + (m2 * dqx - m0 * dqz);
    w2 = onf * (s2 * y[2]) + (m0 * dqy - m1 * dqx);
  };

  if constexpr (!CONS_EARLY) WBC_QP_FORM_ROWS;
#undef WBC_QP_FORM_ROWS
  // per-lane state of the active set: row c3 of P_k, slot c3 of foot k (N^+ row, multiplier, constraint id), |active set of foot k|
  T Pr0 = c3 == 0 ? (T)1 : (T)0, Pr1 = c3 == 1 ? (T)1 : (T)0, Pr2 = c3 == 2 ? (T)1 : (T)0;
  T Np0 = 0, Np1 = 0, Np2 = 0, u_s = 0;
  int id_s = -1, qk = 0;
  T* const prow = Pl + 16 * f + 4 * c3;   // my row of foot f's P in the LDS table (the spare lane owns a dummy row: no predicate on the write)
  {
    T* c = Cl + 3 * (2 * l16);
    c[0] = cAx; c[1] = cAy; c[2] = cAz; c[3] = cBx; c[4] = cBy; c[5] = cBz;
    // tables addressed by the run-time foot index kp: P_k = I, lever arm d_k (zero for a swing foot)
    { T* pr = Pl + 16 * f + 4 * c3; pr[0] = Pr0; pr[1] = Pr1; pr[2] = Pr2; }
    if (isvar) Dl[3 * f + c3] = c3 == 0 ? dqx : (c3 == 1 ? dqy : dqz);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  }
  // ------------------------------------------------------------------ unconstrained minimum x0 = B^T G^-1 S^(1/2) b   (b = w_des - rhat_base)
  T x_me = 0;
  bool actA = false, actB = false;
  if constexpr (WARM != 0) {
    // -------------------------------------------------------------- block set-up from the given active set (see the comment above the template)
    {   // swing feet hold no rows; bits that name no constraint are dropped
      int ok_bits = 0;
      sfor<0, 4>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; ok_bits |= ((mask >> k) & 1) ? ((0xF << (4 * k)) | (0x3 << (16 + 4 * k))) : 0; });
      aset &= ok_bits;
    }
    const T ec0 = c3 == 0 ? (T)1 : (T)0, ec1 = c3 == 1 ? (T)1 : (T)0, ec2 = c3 == 2 ? (T)1 : (T)0;
#pragma clang loop unroll(disable)
    for (int pass = 0; pass < 3; ++pass) {   // (rolled: the later passes are the rare fall-backs, not more copies of the set-up in the instruction stream)
      // (a) the slots of my foot: candidates in the order A0 A1 A2 A3 B0 B1
      const int a4 = (aset >> (4 * f)) & 0xF, b2 = (aset >> (16 + 4 * f)) & 0x3;
      int m6 = a4 | (b2 << 4);
      int q = __builtin_popcount((unsigned)m6);
      bool bad = q > 3 || (a4 & 0xC) == 0xC;          // more than three rows, or both bounds of the normal force
      m6 = bad ? 0 : m6;
      q = bad ? 0 : q;
      const int m6b = m6 & (m6 - 1), m6c = m6b & (m6b - 1);
      const int p0 = m6 ? __builtin_ctz((unsigned)m6) : 0, p1 = m6b ? __builtin_ctz((unsigned)m6b) : 0, p2 = m6c ? __builtin_ctz((unsigned)m6c) : 0;
      const int i0 = p0 < 4 ? 2 * (4 * f + p0) : 2 * (4 * f + p0 - 4) + 1;
      const int i1 = p1 < 4 ? 2 * (4 * f + p1) : 2 * (4 * f + p1 - 4) + 1;
      const int i2 = p2 < 4 ? 2 * (4 * f + p2) : 2 * (4 * f + p2 - 4) + 1;
      const T n00 = Cl[3 * i0], n01 = Cl[3 * i0 + 1], n02 = Cl[3 * i0 + 2];
      const T n10 = Cl[3 * i1], n11 = Cl[3 * i1 + 1], n12 = Cl[3 * i1 + 2];
      const T n20 = Cl[3 * i2], n21 = Cl[3 * i2 + 1], n22 = Cl[3 * i2 + 2];
      // (b) P_k and N_k^+ in closed form
      const T nn0 = n00 * n00 + n01 * n01 + n02 * n02;
      const T wx = n01 * n12 - n02 * n11, wy = n02 * n10 - n00 * n12, wz = n00 * n11 - n01 * n10;   // n0 x n1
      const T ww = wx * wx + wy * wy + wz * wz;
      const T cx = n11 * n22 - n12 * n21, cy = n12 * n20 - n10 * n22, cz = n10 * n21 - n11 * n20;   // n1 x n2
      const T det = n00 * cx + n01 * cy + n02 * cz;
      bad = bad || (q == 2 && !(ww > (T)1e-12)) || (q == 3 && !(det * det > (T)1e-12));
      const T inv = rcp_nr(q == 1 ? nn0 : (q == 2 ? ww : (q == 3 ? det : (T)1)));
      const T n0c = c3 == 0 ? n00 : (c3 == 1 ? n01 : n02), wc = c3 == 0 ? wx : (c3 == 1 ? wy : wz);
      const T g1 = n0c * inv, g2 = wc * inv;
      Pr0 = q == 0 ? ec0 : (q == 1 ? ec0 - g1 * n00 : (q == 2 ? g2 * wx : (T)0));
      Pr1 = q == 0 ? ec1 : (q == 1 ? ec1 - g1 * n01 : (q == 2 ? g2 * wy : (T)0));
      Pr2 = q == 0 ? ec2 : (q == 1 ? ec2 - g1 * n02 : (q == 2 ? g2 * wz : (T)0));
      {
        // slot 0: n0 | n1 x w | n1 x n2;   slot 1: w x n0 | n2 x n0;   slot 2: w      (q = 1 | 2 | 3), all times inv
        const T ax = n11 * wz - n12 * wy, ay = n12 * wx - n10 * wz, az = n10 * wy - n11 * wx;       // n1 x w
        const T bx = wy * n02 - wz * n01, by_ = wz * n00 - wx * n02, bz = wx * n01 - wy * n00;      // w x n0
        const T ex = n21 * n02 - n22 * n01, ey = n22 * n00 - n20 * n02, ez = n20 * n01 - n21 * n00; // n2 x n0
        const T r0x = q == 1 ? n00 : (q == 2 ? ax : cx), r0y = q == 1 ? n01 : (q == 2 ? ay : cy), r0z = q == 1 ? n02 : (q == 2 ? az : cz);
        const T r1x = q == 2 ? bx : ex, r1y = q == 2 ? by_ : ey, r1z = q == 2 ? bz : ez;
        const T k_ = (isvar && c3 < q) ? inv : (T)0;
        Np0 = k_ * (c3 == 0 ? r0x : (c3 == 1 ? r1x : wx));
        Np1 = k_ * (c3 == 0 ? r0y : (c3 == 1 ? r1y : wy));
        Np2 = k_ * (c3 == 0 ? r0z : (c3 == 1 ? r1z : wz));
      }
      qk = q;
      const bool slot_on = isvar && c3 < q;
      const int p_me = c3 == 0 ? p0 : (c3 == 1 ? p1 : p2);
      id_s = slot_on ? (c3 == 0 ? i0 : (c3 == 1 ? i1 : i2)) : -1;
      const T rs = slot_on ? (p_me == 2 ? (T)prm.fn_min : (p_me == 3 ? -(T)prm.fn_max : (T)0)) : (T)0;
      // (c) the four P_k through LDS
      prow[0] = Pr0; prow[1] = Pr1; prow[2] = Pr2;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      // (d) G_A = alpha I + S^(1/2) [sum P, sum X^T; sum X, sum Y] S^(1/2),  X = [d]x P,  Y = [d]x P [d]x^T   (lower triangle, 21 entries)
      T Gm[6][6];
      {
        T Sp[6] = {0, 0, 0, 0, 0, 0}, Sx[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Sy[6] = {0, 0, 0, 0, 0, 0};
#pragma clang loop unroll(disable)
        for (int k = 0; k < 4; ++k) {   // (rolled: one foot's P and lever arm live at a time)
          const T of = ((mask >> k) & 1) ? (T)1 : (T)0;
          const T* pk = Pl + 16 * k;
          const T pxx = of * pk[0], pxy = of * pk[1], pxz = of * pk[2], pyy = of * pk[5], pyz = of * pk[6], pzz = of * pk[10];
          const T dx = Dl[3 * k], dy = Dl[3 * k + 1], dz = Dl[3 * k + 2];
          const T x00 = dy * pxz - dz * pxy, x01 = dy * pyz - dz * pyy, x02 = dy * pzz - dz * pyz;
          const T x10 = dz * pxx - dx * pxz, x11 = dz * pxy - dx * pyz, x12 = dz * pxz - dx * pzz;
          const T x20 = dx * pxy - dy * pxx, x21 = dx * pyy - dy * pxy, x22 = dx * pyz - dy * pxz;
          Sp[0] += pxx; Sp[1] += pxy; Sp[2] += pxz; Sp[3] += pyy; Sp[4] += pyz; Sp[5] += pzz;
          Sx[0] += x00; Sx[1] += x01; Sx[2] += x02; Sx[3] += x10; Sx[4] += x11; Sx[5] += x12; Sx[6] += x20; Sx[7] += x21; Sx[8] += x22;
          Sy[0] += dy * x02 - dz * x01; Sy[1] += dz * x00 - dx * x02; Sy[2] += dx * x01 - dy * x00;      // Y row 0 = d x X_0
          Sy[3] += dz * x10 - dx * x12; Sy[4] += dx * x11 - dy * x10; Sy[5] += dx * x21 - dy * x20;      // Y11, Y12, Y22
        }
        Gm[0][0] = alpha_l + (s0 * s0) * Sp[0]; Gm[1][0] = (s1 * s0) * Sp[1]; Gm[1][1] = alpha_l + (s1 * s1) * Sp[3];
        Gm[2][0] = (s2 * s0) * Sp[2]; Gm[2][1] = (s2 * s1) * Sp[4]; Gm[2][2] = alpha_l + (s2 * s2) * Sp[5];
        Gm[3][0] = (s3 * s0) * Sx[0]; Gm[3][1] = (s3 * s1) * Sx[1]; Gm[3][2] = (s3 * s2) * Sx[2];
        Gm[4][0] = (s4 * s0) * Sx[3]; Gm[4][1] = (s4 * s1) * Sx[4]; Gm[4][2] = (s4 * s2) * Sx[5];
        Gm[5][0] = (s5 * s0) * Sx[6]; Gm[5][1] = (s5 * s1) * Sx[7]; Gm[5][2] = (s5 * s2) * Sx[8];
        Gm[3][3] = alpha_l + (s3 * s3) * Sy[0]; Gm[4][3] = (s4 * s3) * Sy[1]; Gm[5][3] = (s5 * s3) * Sy[2];
        Gm[4][4] = alpha_l + (s4 * s4) * Sy[3]; Gm[5][4] = (s5 * s4) * Sy[4]; Gm[5][5] = alpha_l + (s5 * s5) * Sy[5];
      }
      // (e) G_A = L L^T (in place, il = 1 / diagonal)  (f) my row gi of G_A^-1: L w = e_gi, L^T g = w
      {
        T il[6];
        sfor<0, 6>([&](auto jc) __attribute__((always_inline)) {
          constexpr int j = decltype(jc)::value;
          T dj = Gm[j][j];
          sfor<0, j>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; dj -= Gm[j][k] * Gm[j][k]; });
          il[j] = rsqrt_nr(dj);
          sfor<j + 1, 6>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            T sij = Gm[i][j];
            sfor<0, j>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; sij -= Gm[i][k] * Gm[j][k]; });
            Gm[i][j] = sij * il[j];
          });
        });
        T w[6];
        sfor<0, 6>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          T si = gi == i ? (T)1 : (T)0;
          sfor<0, i>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; si -= Gm[i][k] * w[k]; });
          w[i] = si * il[i];
        });
        sfor_down<0, 6>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          T si = w[i];
          sfor<i + 1, 6>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; si -= Gm[k][i] * Gr[k]; });
          Gr[i] = si * il[i];
        });
      }
      // (g) the target wrench (the producers may still be at it: waited for HERE, behind the factorisation, and read again in the rare
      // second pass rather than kept in twelve registers across the set-up), f_k^p and the right-hand side
      if (pass == 0) {
        WBC_QSTAMP(3);
        if constexpr (WSLDS) { if (sync) qp_wait(sync->geom, sync->need_b); }
        if constexpr (WSLDS && RHAT) { if (sync) qp_wait(sync->rhat, sync->need_rhat); }
        WBC_QSTAMP(4);
      }
      T bb[6];
      {
        const T b_ld = (l16 < 6) ? BLD(l16) - (RHAT ? WSLD(WS_RHAT + l16) : (T)0) : (T)0;
        bb[0] = s0 * dppx<0x150 + 0>(b_ld); bb[1] = s1 * dppx<0x150 + 1>(b_ld); bb[2] = s2 * dppx<0x150 + 2>(b_ld);
        bb[3] = s3 * dppx<0x150 + 3>(b_ld); bb[4] = s4 * dppx<0x150 + 4>(b_ld); bb[5] = s5 * dppx<0x150 + 5>(b_ld);
      }
      const T cp0 = Np0 * rs, cp1 = Np1 * rs, cp2 = Np2 * rs;
      const T fp0 = (dppx<0x00>(cp0) + dppx<0x55>(cp0)) + dppx<0xAA>(cp0), fp1 = (dppx<0x00>(cp1) + dppx<0x55>(cp1)) + dppx<0xAA>(cp1),
              fp2 = (dppx<0x00>(cp2) + dppx<0x55>(cp2)) + dppx<0xAA>(cp2);
      T rr[6];
      {   // sum over the four feet of B_k f_k^p: my foot's share, then the quads added up (row_ror 4, 8); nothing to add when no box row is active
        T t6[6] = {onf * (s0 * fp0), onf * (s1 * fp1), onf * (s2 * fp2), s3 * (dqy * fp2 - dqz * fp1), s4 * (dqz * fp0 - dqx * fp2), s5 * (dqx * fp1 - dqy * fp0)};
        sfor<0, 6>([&](auto jc) __attribute__((always_inline)) {
          constexpr int j = decltype(jc)::value;
          T t = t6[j];
          t += dppx<0x124>(t);
          t += dppx<0x128>(t);
          rr[j] = t - bb[j];
        });
      }
      // (h) y = G_A^-1 rr,  f_k = f_k^p - P_k B_k^T y   (i) u = alpha N^+ (f_k + B_k^T y)
      T yi;
      ginv_mul(rr, yi);
      T w0, w1, w2;
      bt_y(w0, w1, w2);
      x_me = (c3 == 0 ? fp0 : (c3 == 1 ? fp1 : fp2)) - (Pr0 * w0 + Pr1 * w1 + Pr2 * w2);
      x_me = isvar ? x_me : (T)0;
      const T fx = dppx<0x00>(x_me), fy = dppx<0x55>(x_me), fz = dppx<0xAA>(x_me);
      u_s = slot_on ? alpha_l * (Np0 * (fx + w0) + Np1 * (fy + w1) + Np2 * (fz + w2)) : (T)0;
      // (j) an S-pair?  A set that is structurally impossible starts over from the empty set.  Rows with NEGATIVE multipliers -- constraints
      // the state has just left, the common way a carried set goes stale -- are dropped and the set-up repeated on the smaller set (its
      // minimiser has a lower objective and, nearly always, multipliers >= 0: an S-pair one or no iteration from the solution, where the cold
      // start pays one iteration per active row); a second failure starts cold, and the empty set cannot fail.
      const bool neg = slot_on && !(u_s >= 0);
      const unsigned long long fbad = __ballot(bad), fneg = __ballot(neg);
      const bool row_bad = (unsigned)((fbad >> rowbase) & 0xFFFFull) != 0u, row_neg = (unsigned)((fneg >> rowbase) & 0xFFFFull) != 0u;
      int drop = neg ? (1 << (p_me < 4 ? 4 * f + p_me : 16 + 4 * f + (p_me - 4))) : 0;
      drop |= dppx<0x121>(drop); drop |= dppx<0x122>(drop); drop |= dppx<0x124>(drop); drop |= dppx<0x128>(drop);   // OR over the 16 lanes of my row (row_ror 1, 2, 4, 8)
      aset = (row_bad || (row_neg && pass > 0)) ? 0 : (aset & ~drop);
      if ((fbad | fneg) == 0ull) break;
    }
    actA = ((aset >> l16) & 1) != 0;
    actB = hasB && ((aset >> (16 + l16)) & 1) != 0;
  } else
  {
    WBC_QSTAMP(3);
    if constexpr (WSLDS) { if (sync) qp_wait(sync->geom, sync->need_b); }
    if constexpr (WSLDS && RHAT && !SPEC) { if (sync) qp_wait(sync->rhat, sync->need_rhat); }
    if constexpr (SPEC) {   // (a.rprev == null: the caller opts out -- wait for rhat and start on b itself; the move below is then by zero)
      if (!a.rprev) { if (sync) qp_wait(sync->rhat, sync->need_rhat); rprev_in = (l16 < 6) ? WSLD(WS_RHAT + l16) : (T)0; }
    }
    WBC_QSTAMP(4);
    if constexpr (PRE) {
      x_me = isvar ? who.pre[36 + v] : (T)0;
    } else {
    T b_ld;
    if constexpr (STG != 0) b_ld = (l16 < 6) ? (T)IMG(ST_B + l16) : (T)0;
    else b_ld = (l16 < 6) ? BLD(l16) - (SPEC ? rprev_in : (RHAT ? WSLD(WS_RHAT + l16) : (T)0)) : (T)0;   // (SPEC: b~, see the top)
    if constexpr (SPEC) {
      if (l16 < 6) L.R[grp][l16] = rprev_in;
      if (sync && sync->rp_ack) {   // r_prev is in my registers: the observer role may overwrite it now (QpSync::rp_ack)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(sync->rp_ack, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    T bb[6];
    bb[0] = s0 * dppx<0x150 + 0>(b_ld); bb[1] = s1 * dppx<0x150 + 1>(b_ld); bb[2] = s2 * dppx<0x150 + 2>(b_ld);
    bb[3] = s3 * dppx<0x150 + 3>(b_ld); bb[4] = s4 * dppx<0x150 + 4>(b_ld); bb[5] = s5 * dppx<0x150 + 5>(b_ld);
    T yi;
    ginv_mul(bb, yi);
    T w0, w1, w2;
    bt_y(w0, w1, w2);
    x_me = c3 == 0 ? w0 : (c3 == 1 ? w1 : w2);
    }
  }

  // ------------------------------------------------------------------ dual active-set iterations (a8)
  // What a lone wavefront pays (tools/issue_probe.hip on MI355X): 13.5 cycles per DEPENDENT fp64 operation, ~5.5 per independent one,
  // ~45 cycles per divergent `if` region (saveexec / branch / restore), 35 per step of a row argmin, 135 per LDS round trip.  So the
  // loop below is written as ONE basic block of selects (no `if` with side effects outside the rare drop path), the two
  // independent chains of a trip sit side by side -- the ratio test over the active multipliers, and the full step with the
  // search for the NEXT candidate at its end point, which is speculative: it is used when the step turns out to be full
  // (almost always) -- and per-row decisions are committed by selects at the end.
  int ip = -1, status = 0, iter = 0;
  int iter0 = 0;   // (SPEC) trips of the phase on b~: the second phase has the whole of max_iter (it used to get what phase 0 left)
  bool done = !live;
  T sip = 0, Rn2 = 1, u_c = 0;   // Rn2 = max(1, largest z . n+ of an added constraint) = Rnorm^2 of the dense method (its new R diagonal is |d2|)
  const T ntol = -prm.qp_tol;

#ifdef WBC_QP_STAMP
  const long long st_t1 = __builtin_readcyclecounter();
  long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long st_last = st_t1;
#define SEG(i) do { const long long now_ = __builtin_readcyclecounter(); seg[i] += now_ - st_last; st_last = now_; } while (0)
#else
#define SEG(i) do {} while (0)
#endif
#pragma clang loop unroll(disable)
  for (int ph = 0; ph < (SPEC ? 2 : 1); ++ph) {   // (SPEC: the iteration on b~, then -- from the point moved to b -- whatever is left of it; rolled: one copy of the loop)
  // (the two helpers below are defined INSIDE the phase loop: defined in front of it, the closure of one of them stayed behind as a dead 36-byte
  //  stack object in one fused instantiation -- no scratch instruction, but a private segment to set up)
  // most violated inactive constraint of the row at the point whose foot components are (xq0, xq1, xq2) in my quad
  auto most_violated = [&](T xq0, T xq1, T xq2, bool aA, bool aB, T& val, int& id) __attribute__((always_inline)) -> bool {
    const T sA = cAx * xq0 + cAy * xq1 + cAz * xq2 - rA;
    const T sB = cBx * xq0 + cBy * xq1 + cBz * xq2;
    const bool vA = on && !aA && sA < ntol;
    const bool vB = on && hasB && !aB && sB < ntol && (!vA || sB < sA);
    val = vB ? sB : (vA ? sA : KeyT<T>::BIG);
    id = 2 * l16 + (vB ? 1 : 0);
    return gargmin(val, id);
  };
  // closed forms for <= 2 active normals n0 (, n1): my row of P, my slot's row of N^+   (drop path only)
  auto rebuild = [&](int q, T n00, T n01, T n02, T n10, T n11, T n12) __attribute__((always_inline)) {
    const T i1 = rcp_nr(q >= 1 ? (n00 * n00 + n01 * n01 + n02 * n02) : (T)1);
    const T wx = n01 * n12 - n02 * n11, wy = n02 * n10 - n00 * n12, wz = n00 * n11 - n01 * n10;    // n0 x n1
    const T i2 = rcp_nr(q >= 2 ? (wx * wx + wy * wy + wz * wz) : (T)1);
    const T ec0 = c3 == 0 ? (T)1 : (T)0, ec1 = c3 == 1 ? (T)1 : (T)0, ec2 = c3 == 2 ? (T)1 : (T)0;
    const T n0c = c3 == 0 ? n00 : (c3 == 1 ? n01 : n02), wc = c3 == 0 ? wx : (c3 == 1 ? wy : wz);
    if (q == 0) { Pr0 = ec0; Pr1 = ec1; Pr2 = ec2; Np0 = 0; Np1 = 0; Np2 = 0; }
    else if (q == 1) {
      const T g = n0c * i1;
      Pr0 = ec0 - g * n00; Pr1 = ec1 - g * n01; Pr2 = ec2 - g * n02;
      const T k0 = c3 == 0 ? i1 : (T)0;
      Np0 = n00 * k0; Np1 = n01 * k0; Np2 = n02 * k0;
    } else {
      const T g = wc * i2;
      Pr0 = g * wx; Pr1 = g * wy; Pr2 = g * wz;
      const T r0x = n11 * wz - n12 * wy, r0y = n12 * wx - n10 * wz, r0z = n10 * wy - n11 * wx;   // (n1 x w) / |w|^2
      const T r1x = wy * n02 - wz * n01, r1y = wz * n00 - wx * n02, r1z = wx * n01 - wy * n00;   // (w x n0) / |w|^2
      const T k0 = c3 == 0 ? i2 : (T)0, k1 = c3 == 1 ? i2 : (T)0;
      Np0 = r0x * k0 + r1x * k1; Np1 = r0y * k0 + r1y * k1; Np2 = r0z * k0 + r1z * k1;
    }
  };
  {  // first candidate, at x0
    T val; int id;
    const bool found = most_violated(dppx<0x00>(x_me), dppx<0x55>(x_me), dppx<0xAA>(x_me), actA, actB, val, id);
    ip = (!done && found) ? id : -1;
    sip = val;
    done = done || !found;
  }
  while (__ballot(!done && ip >= 0) != 0ull) {   // (every trip counts against max_iter in every row that is still working)
    bool go = !done && ip >= 0;
    iter += go ? 1 : 0;
    const bool over = go && iter - iter0 > prm.max_iter;
    status = over ? 1 : status;
    done = done || over;
    go = go && !over;
    const int ipc = ip < 0 ? 0 : ip;
    const int lp = (ipc >> 1) & 15, kp = lp >> 2;
    // ---- candidate normal, P and lever arm of its foot (LDS, run-time indices); v, b formed by every lane
    const T np0 = Cl[3 * ipc], np1 = Cl[3 * ipc + 1], np2 = Cl[3 * ipc + 2];
    const T* pk_ = Pl + 16 * kp;
    const T pxx = pk_[0], pxy = pk_[1], pxz = pk_[2], pyy = pk_[5], pyz = pk_[6], pzz = pk_[10];
    const T dkx = Dl[3 * kp], dky = Dl[3 * kp + 1], dkz = Dl[3 * kp + 2];
    const T v0 = pxx * np0 + pxy * np1 + pxz * np2, v1 = pxy * np0 + pyy * np1 + pyz * np2, v2 = pxz * np0 + pyz * np1 + pzz * np2;
    const T vv = v0 * v0 + v1 * v1 + v2 * v2;
    T bb[6];
    bb[0] = s0 * v0; bb[1] = s1 * v1; bb[2] = s2 * v2;
    bb[3] = s3 * (dky * v2 - dkz * v1); bb[4] = s4 * (dkz * v0 - dkx * v2); bb[5] = s5 * (dkx * v1 - dky * v0);
    T yi;
    ginv_mul(bb, yi);
    const T by = ((bb[0] * y[0] + bb[1] * y[1]) + (bb[2] * y[2] + bb[3] * y[3])) + (bb[4] * y[4] + bb[5] * y[5]);
    const T znA = vv - by;                 // alpha (z . n+)
    const T zn = znA * ralpha;
    SEG(0);
    // ---- step directions of my variable and my slot
    T w0, w1, w2;
    bt_y(w0, w1, w2);
    const bool mine = f == kp;
    const T vc = c3 == 0 ? v0 : (c3 == 1 ? v1 : v2);
    const T z_me = ((mine ? vc : (T)0) - (Pr0 * w0 + Pr1 * w1 + Pr2 * w2)) * ralpha;
    const T a_s = Np0 * np0 + Np1 * np1 + Np2 * np2;               // my slot's coefficient of n+ (foot kp only)
    const T r_me = (mine ? a_s : (T)0) - (Np0 * w0 + Np1 * w1 + Np2 * w2);
    const bool slot_act = isvar && c3 < qk;
    // ---- the full step and, at its end point, the next candidate (speculative: valid when the step is full).
    // Linearly dependent on the active normals of its foot = no primal step: the dense method's test |d2| <= eps Rnorm, plus a
    // purely local one (v = P n+ is rounding noise of n+: P is kept by rank-one downdates, exact only to ~1e-16; the constraint
    // normals have |n|^2 between 1 and 1 + mu^2)
    const bool indep = zn > (EPS * EPS) * Rn2 && vv > (T)8e-28;
    const T rz = rcp_nr(indep ? zn : (T)1);
    const T t2 = indep ? -sip * rz : INF;
    const T x_full = x_me + (indep ? t2 : (T)0) * z_me;
    const bool own = l16 == lp;
    const bool sA_act = actA || (own && !(ipc & 1)), sB_act = actB || (own && (ipc & 1));   // the candidate is active at that point
    T nval; int nid;
    const bool nfound = most_violated(dppx<0x00>(x_full), dppx<0x55>(x_full), dppx<0xAA>(x_full), sA_act, sB_act, nval, nid);
    SEG(1);
    // ---- is the step full?  It is unless some active multiplier reaches zero first: u_j / r_j < t2 for a slot with r_j > 0 --
    // tested as u_j < t2 r_j, no division, and "does any lane of my row say so" read off the wavefront's ballot.  The ratio
    // test proper (reciprocal, row argmin for the blocking slot) is only needed when a row is NOT full: it lives in the rare
    // branch below, off the path of the 98 % of the trips that add their candidate.
    const bool pos = slot_act && r_me > 0;
    const unsigned long long blk = __ballot(pos && (!indep || u_s < t2 * r_me));
    const bool row_blk = (unsigned)((blk >> rowbase) & 0xFFFFull) != 0u;
    const bool full = indep && !row_blk;
    const bool dual_only = !indep;
    const bool addg = go && full, slowg = go && !full;
    x_me = addg ? x_full : x_me;
    u_s = (addg && slot_act) ? u_s - t2 * r_me : u_s;
    u_c = addg ? u_c + t2 : u_c;
    // full step: the candidate joins the active set of its foot
    {
      const T gy = addg ? yi * (rz * ralpha) : (T)0;                 // y_i / (alpha z . n+)
      sfor<0, 6>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; Gr[j] += gy * y[j]; });
      const T ivv = rcp_nr1(vv > 0 ? vv : (T)1);
      const T e0 = v0 * ivv, e1 = v1 * ivv, e2 = v2 * ivv;       // v / |v|^2: the new row of N^+
      const bool upd = addg && mine;
      const bool shrink = upd && slot_act, fresh = upd && isvar && c3 == qk;
      const T as_ = shrink ? a_s : (T)0;
      Np0 = fresh ? e0 : Np0 - as_ * e0;
      Np1 = fresh ? e1 : Np1 - as_ * e1;
      Np2 = fresh ? e2 : Np2 - as_ * e2;
      u_s = fresh ? u_c : u_s;
      id_s = fresh ? ipc : id_s;
      const T vcu = upd ? vc : (T)0;
      Pr0 -= vcu * e0; Pr1 -= vcu * e1; Pr2 -= vcu * e2;
      qk += upd ? 1 : 0;
      prow[0] = Pr0; prow[1] = Pr1; prow[2] = Pr2;                 // every lane, every trip: its (possibly unchanged) row
      Rn2 = (addg && zn > Rn2) ? zn : Rn2;
      actA = actA || (addg && own && !(ipc & 1));
      actB = actB || (addg && own && (ipc & 1));
      // the next candidate of the rows that added: the speculative search was made at their new point
      ip = addg ? (nfound ? nid : -1) : ip;
      sip = addg ? nval : sip;
      u_c = addg ? (T)0 : u_c;
      done = done || (addg && !nfound);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (LDS is in order within a wavefront: the table reads of the next trip follow these writes)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    SEG(4);
    // ---- partial / dual-only step: the blocking constraint leaves its foot's active set (rare: wave-uniform branch)
    if (__ballot(slowg) != 0ull) {
      // ratio test over the active multipliers of the rows that could not take their full step
      T t1k = pos ? u_s * rcp_nr(r_me) : KeyT<T>::BIG;
      int kmin = l16;
      const bool t1found = gargmin(t1k, kmin);
      const bool infeas = slowg && !t1found;           // (not full and nothing blocks: the step has no primal part and no bound)
      status = infeas ? 2 : status;
      done = done || infeas;
      const bool dropg = slowg && t1found;
      const T t1 = t1k;
      x_me = (dropg && !dual_only) ? x_me + t1 * z_me : x_me;
      u_s = (dropg && slot_act) ? u_s - t1 * r_me : u_s;
      u_c = dropg ? u_c + t1 : u_c;
      const int kq = dropg ? kmin : 0;                 // lane (within the row) of the blocking slot
      const int kd = kq >> 2, sd = kq & 3;             // its foot and slot
      const int cid = gread(id_s, kq, rowbase);        // the constraint that leaves
      if (dropg && l16 == ((cid >> 1) & 15)) { if (cid & 1) actB = false; else actA = false; }
      const bool inq = dropg && f == kd;
      const T Po0 = Pr0, Po1 = Pr1, Po2 = Pr2;
      {  // close the hole in the slot list of foot kd
        const T un = dppx<0xF9>(u_s);                  // quad_perm [1,2,3,3]: the next slot's values
        const int idn = dppx<0xF9>(id_s);
        if (inq && c3 >= sd && c3 < qk - 1) { u_s = un; id_s = idn; }
        if (inq && c3 == qk - 1) { u_s = 0; id_s = -1; }
        if (inq) --qk;
      }
      // remaining normals of foot kd (ids of slots 0, 1 after the shift)
      const int id0 = dppx<0x00>(id_s), id1 = dppx<0x55>(id_s);
      const int i0 = (inq && qk >= 1) ? id0 : 0, i1 = (inq && qk >= 2) ? id1 : 0;
      const T n00 = Cl[3 * i0], n01 = Cl[3 * i0 + 1], n02 = Cl[3 * i0 + 2];
      const T n10 = Cl[3 * i1], n11 = Cl[3 * i1 + 1], n12 = Cl[3 * i1 + 2];
      if (inq) rebuild(qk, n00, n01, n02, n10, n11, n12);
      prow[0] = Pr0; prow[1] = Pr1; prow[2] = Pr2;
      // what: the direction P_k gained, what what^T = P_k(new) - P_k(old): row c3 of that difference is what_c what; the lane of
      // foot kd with the largest |what_c| normalises its row and publishes what through LDS
      {
        const T D0 = Pr0 - Po0, D1 = Pr1 - Po1, D2 = Pr2 - Po2;
        const T dcc = c3 == 0 ? D0 : (c3 == 1 ? D1 : D2);                 // what_c^2
        T key = (inq && isvar) ? -dcc : KeyT<T>::BIG;
        int who_ = l16;
        gargmin(key, who_);
        if (inq && l16 == who_) {
          const T sc = rsqrt_nr(dcc > 0 ? dcc : (T)1);
          Wl[0] = D0 * sc; Wl[1] = D1 * sc; Wl[2] = D2 * sc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        // (rows that do not drop read whatever the scratch holds: zeroed, so that their G_A^-1 sees an update by exactly nothing)
        const T h0 = dropg ? Wl[0] : (T)0, h1 = dropg ? Wl[1] : (T)0, h2 = dropg ? Wl[2] : (T)0;
        const int kdc = dropg ? kd : 0;
        const T ex = Dl[3 * kdc], ey = Dl[3 * kdc + 1], ez = Dl[3 * kdc + 2];
        T bh[6];
        bh[0] = s0 * h0; bh[1] = s1 * h1; bh[2] = s2 * h2;
        bh[3] = s3 * (ey * h2 - ez * h1); bh[4] = s4 * (ez * h0 - ex * h2); bh[5] = s5 * (ex * h1 - ey * h0);
        T ygi;
        ginv_mul(bh, ygi);                                                // y = G_A^-1 bh (overwrites y: the step is over)
        const T den = (T)1 + (((bh[0] * y[0] + bh[1] * y[1]) + (bh[2] * y[2] + bh[3] * y[3])) + (bh[4] * y[4] + bh[5] * y[5]));
        const T gd = dropg ? ygi * rcp_nr(den) : (T)0;
        sfor<0, 6>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; Gr[j] -= gd * y[j]; });
      }
      {  // a partial step moved x: refresh the candidate's slack (cross-lane ops stay unconditional)
        const T xq0 = dppx<0x00>(x_me), xq1 = dppx<0x55>(x_me), xq2 = dppx<0xAA>(x_me);
        const T sA = cAx * xq0 + cAy * xq1 + cAz * xq2 - rA, sB = cBx * xq0 + cBy * xq1 + cBz * xq2;
        const T sv = gread((ipc & 1) ? sB : sA, lp, rowbase);
        if (dropg && !dual_only) sip = sv;
      }
    }
    SEG(5);
  }
  if constexpr (SPEC) {
    if (ph == 0) {
      // ---- rhat is needed now: move the minimiser on the active set reached with b~ to b = b~ - (rhat - r_prev)
      if (sync) qp_wait(sync->rhat, sync->need_rhat);
      const T db_ld = (l16 < 6) ? L.R[grp][l16] - WSLD(WS_RHAT + l16) : (T)0;   // (my own LDS word: program order of one lane)
      T rr[6];
      rr[0] = -(s0 * dppx<0x150 + 0>(db_ld)); rr[1] = -(s1 * dppx<0x150 + 1>(db_ld)); rr[2] = -(s2 * dppx<0x150 + 2>(db_ld));
      rr[3] = -(s3 * dppx<0x150 + 3>(db_ld)); rr[4] = -(s4 * dppx<0x150 + 4>(db_ld)); rr[5] = -(s5 * dppx<0x150 + 5>(db_ld));
      T yi;
      ginv_mul(rr, yi);                       // dy = -G_A^-1 d
      T w0, w1, w2;
      bt_y(w0, w1, w2);                       // B_k^T dy of my foot
      const T dx = isvar ? -(Pr0 * w0 + Pr1 * w1 + Pr2 * w2) : (T)0;     // df_k = -P_k B_k^T dy, my component
      x_me += dx;
      const T dfx = dppx<0x00>(dx), dfy = dppx<0x55>(dx), dfz = dppx<0xAA>(dx);
      const bool slot_now = isvar && c3 < qk;
      u_s += slot_now ? alpha_l * (Np0 * (dfx + w0) + Np1 * (dfy + w1) + Np2 * (dfz + w2)) : (T)0;
      // ---- still an S-pair?  A row with a negative multiplier starts over from the empty set, with b itself.  Per ROW: what a state computes
      // must not depend on the three states that happen to share its wavefront (shards of a batch group the states differently, and their
      // results are compared bit for bit with the unsharded run)
      // (round 5, ADVICE r4: a row whose iteration on b~ ENDED with a status -- iteration limit, "infeasible" -- starts over as well, with a fresh
      // iteration budget: what the solver reports for b must not be a verdict about b~)
      const unsigned long long fneg = __ballot(live && (status != 0 || (slot_now && !(u_s >= 0))));
      if (fneg != 0ull) {   // (wave-uniform branch, rare; inside it every change is selected per row)
        const bool row_over = (unsigned)((fneg >> rowbase) & 0xFFFFull) != 0u;
        status = row_over ? 0 : status;
        iter = row_over ? 0 : iter;
        u_c = row_over ? (T)0 : u_c;      // (a row stopped by the iteration limit may sit between a partial step and its full step)
        T GrK[6];
        sfor<0, 6>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; GrK[j] = Gr[j]; });
        s16_cold_inverse_row<T>(mask, d_me, alpha_l, s0, s1, s2, s3, s4, s5, gi, Gr);
        sfor<0, 6>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; Gr[j] = row_over ? Gr[j] : GrK[j]; });
        Pr0 = row_over ? (c3 == 0 ? (T)1 : (T)0) : Pr0; Pr1 = row_over ? (c3 == 1 ? (T)1 : (T)0) : Pr1; Pr2 = row_over ? (c3 == 2 ? (T)1 : (T)0) : Pr2;
        Np0 = row_over ? (T)0 : Np0; Np1 = row_over ? (T)0 : Np1; Np2 = row_over ? (T)0 : Np2;
        u_s = row_over ? (T)0 : u_s;
        id_s = row_over ? -1 : id_s;
        qk = row_over ? 0 : qk;
        actA = row_over ? false : actA; actB = row_over ? false : actB;
        Rn2 = row_over ? (T)1 : Rn2;
        prow[0] = Pr0; prow[1] = Pr1; prow[2] = Pr2;      // (every lane rewrites its row of the LDS table: unchanged for the rows that go on)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const T bt_ld = (l16 < 6) ? BLD(l16) - WSLD(WS_RHAT + l16) : (T)0;
        T bb[6];
        bb[0] = s0 * dppx<0x150 + 0>(bt_ld); bb[1] = s1 * dppx<0x150 + 1>(bt_ld); bb[2] = s2 * dppx<0x150 + 2>(bt_ld);
        bb[3] = s3 * dppx<0x150 + 3>(bt_ld); bb[4] = s4 * dppx<0x150 + 4>(bt_ld); bb[5] = s5 * dppx<0x150 + 5>(bt_ld);
        T y2;
        ginv_mul(bb, y2);                     // (rows that go on compute a product they do not use)
        T v0, v1, v2;
        bt_y(v0, v1, v2);
        const T x0_me = c3 == 0 ? v0 : (c3 == 1 ? v1 : v2);
        x_me = row_over ? x0_me : x_me;
      }
      iter0 = iter;                           // (a row that started over: both zero)
      done = !live || status != 0;            // every row looks for violated rows again at its new point
    }
  }
  }   // ph

  // ------------------------------------------------------------------ outputs: f, tau (a9), status
  WBC_QSTAMP(5);
  WBC_QSTAMP3(10);
  if constexpr (WSLDS) { if (sync) qp_wait(sync->fin, sync->need_fin); }
  WBC_QSTAMP(6);
  {   // the active set at the solution: the per-lane flags, read off two ballots (row-uniform)
    const unsigned long long ba = __ballot(actA), bbm = __ballot(actB);
    const int aset_fin = (int)(((unsigned)(ba >> rowbase) & 0xFFFFu) | (((unsigned)(bbm >> rowbase) & 0xFFFFu) << 16));
    if constexpr (WARM == 2) { if (l16 == 0) *carry = aset_fin; }   // (LDS: read back by this row in the next tick -- program order of one wavefront)
    if constexpr (STG != 0) { if (live && l16 == 0) who.iimg[3 * who.tile + stg_slot] = aset_fin; }
    else if (a.aset_out && live && l16 == 0) a.aset_out[s32] = aset_fin;
  }
  bool to_mem = true;   // (QpSync::skip_out: wavefront-uniform)
  bool from_hand = false;   // (QpSync::hand)
  const TS* hand_img = nullptr;
  if constexpr (WSLDS) {
    if (sync) {
      to_mem = !sync->skip_out;
      if (sync->hand) { qp_wait(sync->hand_flag, sync->need_hand); from_hand = true; hand_img = (const TS*)sync->hand; }
    }
  }
  if (live) {
    T taup = 0, jl0 = 0, jl1 = 0, jl2 = 0;
    int jm = 0;
    jm = (int)((unsigned)(a.jpack >> (4 * (v & 15))) & 15u);
    if constexpr (STG != 0) {
      if (isvar) { taup = (T)IMG(ST_TAUP + v); jl0 = (T)IMG(ST_JCL + 9 * f + 0 + c3); jl1 = (T)IMG(ST_JCL + 9 * f + 3 + c3); jl2 = (T)IMG(ST_JCL + 9 * f + 6 + c3); }
    } else if (isvar) {
      taup = WSLD(WS_TAUP + v) - (RHAT ? WSLD(WS_RHAT + 6 + v) : (T)0);
      if (from_hand) {
        const int hslot = 16 * f + (int)((tx >> 4) & 15);   // (my state's slot in the workgroup)
        jl0 = (T)hand_img[(24 + c3) * 64 + hslot]; jl1 = (T)hand_img[(27 + c3) * 64 + hslot]; jl2 = (T)hand_img[(30 + c3) * 64 + hslot];
      } else if (geom_jc) {
        jl0 = GLD(a.Jc, (3 * f + 0) * 18 + 6 + jm); jl1 = GLD(a.Jc, (3 * f + 1) * 18 + 6 + jm); jl2 = GLD(a.Jc, (3 * f + 2) * 18 + 6 + jm);
      } else {
        jl0 = WSLD(WS_JCL + 9 * f + 0 + c3); jl1 = WSLD(WS_JCL + 9 * f + 3 + c3); jl2 = WSLD(WS_JCL + 9 * f + 6 + c3);
      }
    }
    const T xq0 = dppx<0x00>(x_me), xq1 = dppx<0x55>(x_me), xq2 = dppx<0xAA>(x_me);
    if (isvar) {
      const T fx = on ? xq0 : (T)0, fy = on ? xq1 : (T)0, fz = on ? xq2 : (T)0;
      if constexpr (STG != 0) {
        IMG(ST_F + v) = (TS)(on ? x_me : (T)0);
        IMG(ST_TAU + jm) = (TS)(taup - (jl0 * fx + jl1 * fy + jl2 * fz));
      } else if (to_mem) {
        GST(a.f, v, on ? x_me : (T)0);
        GST(a.tau, jm, taup - (jl0 * fx + jl1 * fy + jl2 * fz));
      }
      if constexpr (WSLDS) {
        if (sync && sync->res) {   // (persistent rollout: the integrator reads this tick's tau, f from LDS -- QpSync::res)
          TS* rs = (TS*)sync->res + (int)(tx >> 4);
          rs[(RES_F + v) * 16] = (TS)(on ? x_me : (T)0);
          rs[(RES_TAU + jm) * 16] = (TS)(taup - (jl0 * fx + jl1 * fy + jl2 * fz));
        }
      }
    }
    if constexpr (STG != 0) {
      if (l16 == 0) { who.iimg[who.tile + stg_slot] = status; who.iimg[2 * who.tile + stg_slot] = iter; }
    } else if (l16 == 0 && to_mem) {
      a.status[s32] = status;
#ifdef WBC_QP_STAMP
      {
        const long long vals[8] = {st_t1 - st_t0, seg[0], seg[1], seg[2], seg[3], seg[4], seg[5], seg[6]};
        long long lo = vals[0], hi = vals[1];
        if (grp == 1) { lo = vals[2]; hi = vals[3]; } else if (grp == 2) { lo = vals[4]; hi = vals[5]; } else if (grp == 3) { lo = vals[6]; hi = vals[7]; }
        lo >>= 4; hi >>= 4;
        if (lo > 0xFFFF) lo = 0xFFFF;
        if (hi > 0x7FFF) hi = 0x7FFF;
        if (a.iters) a.iters[s32] = (int)(lo | (hi << 16));
      }
#else
      if (a.iters) a.iters[s32] = iter;
#endif
    }
  }
  WBC_QSTAMP(11);
#undef SEG
#undef IMG
#undef GST
#undef WSLD
#undef BLD
#undef GLD
}

// the QP body of a scalar type.  WBC_QP_STRUCT = 2 (default): the structured form for both scalar types (fp32 solvers: fp32 arrays,
// fp64 arithmetic); 1: structured for fp64, the orthogonal-factor form (qp_group16.hip.hpp) for fp32; 0: always the latter
template <class T, bool WSLDS, bool RHAT = false, int SPW = 16, bool TILED = false, int WPB = (WSLDS || TILED) ? 4 : 1, class Idle = QpNoIdle, bool PRE = false, int WARM = 0, int STG = 0>
WBC_DEV void qp_body(const DevParams<T>& prm, const QpArgs<T>& a, const QpJidx& jmap, const T* wsl, const QpSync* sync = nullptr,
                     const QpWho who = QpWho{0, false}, Idle idle = Idle(), int* carry = nullptr) {
  static_assert(STG == 0 || true, "staged tiles run the structured body");
  if constexpr (WARM != 0 || ((true || std::is_same<T, double>::value)))
    qp_struct16_body<T, WSLDS, RHAT, SPW, TILED, WPB, Idle, PRE, WARM, STG>(prm, a, jmap, wsl, sync, who, idle, carry);
  else qp_group16_body<T, WSLDS, RHAT, SPW, TILED, WPB, Idle>(prm, a, jmap, wsl, sync, who, idle);
}

}  // namespace wbc

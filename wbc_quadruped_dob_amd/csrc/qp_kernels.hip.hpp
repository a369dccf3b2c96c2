// The stand-alone GRF-QP kernels: one-wavefront workgroups (qp_group16_kernel), tiles dealt by predicted work -- inputs gathered per state (qp_tile_kernel) or
// staged through LDS (qp_stile_kernel / qp_stile_body, round 6: also the second half of tile_tick_kernel) --, the dense solver over a device-side list (qp_list_kernel).  The per-QP body is qp_body (qp_struct16.hip.hpp): the structured
// wrench-space form in fp64, the orthogonal-factor form (qp_group16.hip.hpp) in fp32.
#pragma once
#include "qp_struct16.hip.hpp"

namespace wbc {

// RHAT: rhat comes from the separate observer kernel through the HBM workspace (large observer-on batches)
// WARM: every state's iteration starts from the active set in a.aset_in (wbc_step_batch_warm: dependent ticks)
template <class T, bool RHAT = false, bool WARM = false>
__global__ __launch_bounds__(64, WBC_QP_WAVES) void qp_group16_kernel(DevParams<T> prm, QpArgs<T> a, QpJidx jmap) {
  qp_body<T, false, RHAT, 16, false, 1, QpNoIdle, false, WARM ? 1 : 0>(prm, a, jmap, nullptr);
}

// ======================================================================================================================
// qp_tile_kernel: the same QPs, DEALT BY PREDICTED WORK (large batches).
// A wavefront runs until the slowest of its four rows is done, and iteration counts of neighbouring states differ a lot
// (0 ... 13, mean 2.5 on the bench data): with four consecutive states per wavefront the loop runs 5.0 trips per group
// for 2.5 iterations per QP -- half of the row-iterations idle.  Here a 256-thread workgroup owns a TILE of consecutive
// states and
//   1. predicts each state's work, one state per LANE: the unconstrained minimum x0 = B^T G^-1 S^(1/2) b from the same 6x6
//      factor the solver uses (a few hundred instructions per 64 states), the number of constraints x0 violates and by how
//      much (count alone: correlation with the iteration count 0.86-0.89 on the bench data);
//   2. sorts the tile by that key in LDS (counting sort, hardest first);
//   3. its four wavefronts pull groups of four similar states from an LDS counter until the tile is empty -- no wavefront
//      waits for another, rows of a group finish together (about 3 trips per group instead of 5.0), and the short groups
//      at the end of the queue level the tail.
// Results per state are those of qp_group16_kernel (same body, another assignment of states to rows).
// WBC_QP_PRED_FINISH: a state whose unconstrained minimum x0 violates nothing IS solved (f = x0, zero iterations) -- the predictor
// can write f, tau, status, iters itself (one state per lane) and report -1, and the row-form body, which pays its set-up per FOUR
// states before it can tell, never sees it: 28 % of the states of the standing benchmark batch, 70 % of the trot / fp32 ones.
// 1 (default): fp32 solvers only; 2: fp64 too; 0: off.  Measured on MI355X (QP stage, us, off -> on):
//   fp32 trot batch  32 768: 24.3 -> 22.5    65 536: 38.5 -> 35.8    131 072: 74.9 -> 63.8
//   fp64 standing    16 384: 22.2 -> 26.4    32 768: 35.2 -> 38.0    131 072 (tiles only): 120.6 -> 133.5    fp64 trot 32 768: 25.3 -> 23.6
// fp64 loses: the predictor runs on ONE of the tile's four wavefronts, and full-accuracy fp64 reciprocal square roots (14 per state)
// plus the finishing loads lengthen exactly that serial stretch by more than the skipped groups (set-up only, no trips) give back.
// fp32 solvers, whose row form works in fp64, finish a state here only when every slack clears a 1e-3 N margin, so a decision fp32
// rounding could flip is always left to the solver; the factor then uses Newton-refined reciprocal square roots.
// feet per round of finishing loads (4: +38 VGPRs in fp64 -> two instead of three tiles per CU)
// WBC_QP_TILE_PRE (default 1; fp64 tiles of <= 64 states): the predictor also leaves G^-1 (36 values) and x0 (12) of its state in LDS, and
// the row-form body starts from them (qp_struct16_body<..., PRE>) instead of factorising G again in all 16 lanes of the state's row.
constexpr int QP_PRE_WORDS = 48;
// The predictor's arithmetic: the 6x6 factor and z = G^-1 S^(1/2) b, then the feet of `fmask` (x0 of the foot, its six slacks: count, summed
// violation, "every slack clears the finishing threshold") and the columns of G^-1 in `cmask` (PRE only).  (The masks exist because sharing
// one state's work among the tile's four wavefronts was tried: see qp_tile_kernel.)
// A = the arithmetic type: the storage type T, except that a predictor that feeds the (fp64-arithmetic) solver works in double.
template <class T, bool PRE> struct PredA { using type = typename std::conditional<PRE, double, T>::type; };
template <class A> struct PredPart { int cnt; A mag; bool fin_ok; A zf0, zf1, zf2, zm0, zm1, zm2; };
template <class T, bool RHAT, bool PRE = false>
WBC_DEV void qp_predict_part(const DevParams<T>& prm, const QpArgs<T>& a, unsigned s32, unsigned N32, double* pre, int fmask, int cmask,
                             PredPart<typename PredA<T, PRE>::type>& out) {
  using A = typename PredA<T, PRE>::type;
#define PLD(ptr, comp) ((A)(*(const T*)((const char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T)))))
  constexpr bool FIN = (std::is_same<T, float>::value);
  const int mask = a.mask[s32] & 0xF;
  const A s0 = prm.sS[0], s1 = prm.sS[1], s2 = prm.sS[2], s3 = prm.sS[3], s4 = prm.sS[4], s5 = prm.sS[5];
  // (loops over the feet are NOT unrolled where they carry per-foot data: with Dx..Dz[4] and the four feet's normals in flight the
  //  predictor needed 220-250 VGPRs -- more than the solver it serves; the lever arms are re-read, L2-hot, where they are needed)
#define PLD_D(f_, dx_, dy_, dz_) do { \
    if (a.Jc) { dx_ = PLD(a.Jc, (3 * (f_) + 1) * 18 + 5); dy_ = PLD(a.Jc, (3 * (f_) + 2) * 18 + 3); dz_ = PLD(a.Jc, (3 * (f_)) * 18 + 4); } \
    else { dx_ = PLD(a.ws, WS_D + 3 * (f_)); dy_ = PLD(a.ws, WS_D + 3 * (f_) + 1); dz_ = PLD(a.ws, WS_D + 3 * (f_) + 2); } } while (0)
  A nc = 0, sx = 0, sy = 0, sz = 0, Pxx = 0, Pxy = 0, Pxz = 0, Pyy = 0, Pyz = 0, Pzz = 0;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const bool on = (mask >> f) & 1;
    A dx, dy, dz;
    PLD_D(f, dx, dy, dz);
    dx = on ? dx : (A)0; dy = on ? dy : (A)0; dz = on ? dz : (A)0;
    nc += on ? (A)1 : (A)0; sx += dx; sy += dy; sz += dz;
    Pxx += dx * dx; Pxy += dx * dy; Pxz += dx * dz; Pyy += dy * dy; Pyz += dy * dz; Pzz += dz * dz;
  }
  A bt[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) bt[k] = (a.wdes ? PLD(a.wdes, k) : PLD(a.ws, WS_B + k)) - (RHAT ? PLD(a.ws, WS_RHAT + k) : (A)0);
  // G = alpha I + B B^A and its factor, as in qp_group16_body (selection only: seed-accuracy reciprocal square roots)
  const A g00 = prm.alpha + s0 * s0 * nc, g11 = prm.alpha + s1 * s1 * nc, g22 = prm.alpha + s2 * s2 * nc;
  const A gm01 = -(s3 * s1) * sz, gm02 = (s3 * s2) * sy, gm10 = (s4 * s0) * sz, gm12 = -(s4 * s2) * sx, gm20 = -(s5 * s0) * sy, gm21 = (s5 * s1) * sx;
  const A m00 = prm.alpha + (s3 * s3) * (Pyy + Pzz), m11 = prm.alpha + (s4 * s4) * (Pxx + Pzz), m22 = prm.alpha + (s5 * s5) * (Pxx + Pyy);
  const A m10 = -(s4 * s3) * Pxy, m20 = -(s5 * s3) * Pxz, m21 = -(s5 * s4) * Pyz;
  auto rs = [](A x) __attribute__((always_inline)) -> A {
    if constexpr (FIN || PRE) return rsqrt_nr(x);
    else if constexpr (std::is_same<A, double>::value) return __builtin_amdgcn_rsq(x); else return __builtin_amdgcn_rsqf(x);
  };
  A il[6];
  il[0] = rs(g00); il[1] = rs(g11); il[2] = rs(g22);
  const A a01 = gm01 * il[1], a02 = gm02 * il[2], a10 = gm10 * il[0], a12 = gm12 * il[2], a20 = gm20 * il[0], a21 = gm21 * il[1];
  const A c00 = m00 - a01 * a01 - a02 * a02, c11 = m11 - a10 * a10 - a12 * a12, c22 = m22 - a20 * a20 - a21 * a21;
  const A c10 = m10 - a12 * a02, c20 = m20 - a21 * a01, c21 = m21 - a20 * a10;
  il[3] = rs(c00);
  const A b10 = c10 * il[3], b20 = c20 * il[3];
  il[4] = rs(c11 - b10 * b10);
  const A b21 = (c21 - b20 * b10) * il[4];
  il[5] = rs(c22 - b20 * b20 - b21 * b21);
  A w[6], z[6];   // z = G^-1 S^(1/2) b
  w[0] = s0 * bt[0] * il[0]; w[1] = s1 * bt[1] * il[1]; w[2] = s2 * bt[2] * il[2];
  w[3] = (s3 * bt[3] - a01 * w[1] - a02 * w[2]) * il[3];
  w[4] = (s4 * bt[4] - a10 * w[0] - a12 * w[2] - b10 * w[3]) * il[4];
  w[5] = (s5 * bt[5] - a20 * w[0] - a21 * w[1] - b20 * w[3] - b21 * w[4]) * il[5];
  z[5] = w[5] * il[5];
  z[4] = (w[4] - b21 * z[5]) * il[4];
  z[3] = (w[3] - b10 * z[4] - b20 * z[5]) * il[3];
  z[2] = (w[2] - a02 * z[3] - a12 * z[4]) * il[2];
  z[1] = (w[1] - a01 * z[3] - a21 * z[5]) * il[1];
  z[0] = (w[0] - a10 * z[4] - a20 * z[5]) * il[0];
  const A zf0 = s0 * z[0], zf1 = s1 * z[1], zf2 = s2 * z[2], zm0 = s3 * z[3], zm1 = s4 * z[4], zm2 = s5 * z[5];
  if constexpr (PRE) {   // G^-1, column by column (= row by row: symmetric): G x = e_c through the factor, zeros of e_c skipped by the compiler
    sfor<0, 6>([&](auto cc_) __attribute__((always_inline)) {
      constexpr int c = decltype(cc_)::value;
      if (!((cmask >> c) & 1)) return;      // (wavefront-uniform)
      const A e0 = c == 0 ? (A)1 : (A)0, e1 = c == 1 ? (A)1 : (A)0, e2 = c == 2 ? (A)1 : (A)0, e3 = c == 3 ? (A)1 : (A)0, e4 = c == 4 ? (A)1 : (A)0,
              e5 = c == 5 ? (A)1 : (A)0;
      A u[6], gc[6];
      u[0] = e0 * il[0]; u[1] = e1 * il[1]; u[2] = e2 * il[2];
      u[3] = (e3 - a01 * u[1] - a02 * u[2]) * il[3];
      u[4] = (e4 - a10 * u[0] - a12 * u[2] - b10 * u[3]) * il[4];
      u[5] = (e5 - a20 * u[0] - a21 * u[1] - b20 * u[3] - b21 * u[4]) * il[5];
      gc[5] = u[5] * il[5];
      gc[4] = (u[4] - b21 * gc[5]) * il[4];
      gc[3] = (u[3] - b10 * gc[4] - b20 * gc[5]) * il[3];
      gc[2] = (u[2] - a02 * gc[3] - a12 * gc[4]) * il[2];
      gc[1] = (u[1] - a01 * gc[3] - a21 * gc[5]) * il[1];
      gc[0] = (u[0] - a10 * gc[4] - a20 * gc[5]) * il[0];
#pragma unroll
      for (int j = 0; j < 6; ++j) pre[6 * c + j] = (double)gc[j];
    });
  }
  int cnt_all = 0;
  bool fin_ok = true;   // every slack of every stance foot at or above the finishing threshold (false for a NaN state: the solver reports those)
  const A fin_thr = std::is_same<T, double>::value ? (A)-prm.qp_tol : (A)1e-3;
  A mag = 0;   // summed violation of the violated constraints
#pragma unroll 4
  for (int f = 0; f < 4; ++f) {
    // x0 of foot f = on (s_f z_f + (s_m z_m) x d_f)
    if (!((fmask >> f) & 1)) continue;      // (wavefront-uniform)
    const bool on = (mask >> f) & 1;
    A dx, dy, dz;
    PLD_D(f, dx, dy, dz);
    const A x0 = on ? zf0 + (zm1 * dz - zm2 * dy) : (A)0;
    const A x1 = on ? zf1 + (zm2 * dx - zm0 * dz) : (A)0;
    const A x2 = on ? zf2 + (zm0 * dy - zm1 * dx) : (A)0;
    if constexpr (PRE) { pre[36 + 3 * f] = (double)x0; pre[36 + 3 * f + 1] = (double)x1; pre[36 + 3 * f + 2] = (double)x2; }
    A nx = PLD(a.normals, 3 * f), ny = PLD(a.normals, 3 * f + 1), nz = PLD(a.normals, 3 * f + 2);
    const A iln = rs(nx * nx + ny * ny + nz * nz);
    nx *= iln; ny *= iln; nz *= iln;
    const bool usex = fabs_t(nx) < (A)0.9;
    const A rx = usex ? (A)1 : (A)0, ry = usex ? (A)0 : (A)1;
    const A rd = rx * nx + ry * ny;
    A t1x = rx - nx * rd, t1y = ry - ny * rd, t1z = -nz * rd;
    const A it = rs(t1x * t1x + t1y * t1y + t1z * t1z);
    t1x *= it; t1y *= it; t1z *= it;
    const A t2x = ny * t1z - nz * t1y, t2y = nz * t1x - nx * t1z, t2z = nx * t1y - ny * t1x;
    const A fn = nx * x0 + ny * x1 + nz * x2, f1 = t1x * x0 + t1y * x1 + t1z * x2, f2 = t2x * x0 + t2y * x1 + t2z * x2;
    const A mf = PLD(a.mu, f) * prm.mu_scale * fn, tol = -prm.qp_tol;
    const A sl[6] = {mf - f1, mf + f1, mf - f2, mf + f2, fn - prm.fn_min, prm.fn_max - fn};
    int cnt = 0;
    A mg = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) { cnt += (sl[c] < tol) ? 1 : 0; mg += (sl[c] < tol) ? -sl[c] : (A)0; }
    bool okf = true;
#pragma unroll
    for (int c = 0; c < 6; ++c) okf = okf && (sl[c] >= fin_thr);
    fin_ok = fin_ok && (okf || !on);
    cnt_all += on ? cnt : 0;
    mag += on ? mg : (A)0;
  }
  out.cnt = cnt_all; out.mag = mag; out.fin_ok = fin_ok;
  out.zf0 = zf0; out.zf1 = zf1; out.zf2 = zf2; out.zm0 = zm0; out.zm1 = zm1; out.zm2 = zm2;
#undef PLD_D
#undef PLD
}

// a state whose x0 violates nothing: f = x0, tau = taup - rhat - Jc_leg^T f, status 0, no iterations (the epilogue of qp_struct16_body, one state per lane)
template <class T, bool RHAT, bool PRE = false>
WBC_DEV void qp_predict_finish(const DevParams<T>& prm, const QpArgs<T>& a, const QpJidx& jmap, unsigned s32, unsigned N32,
                               const PredPart<typename PredA<T, PRE>::type>& pp) {
  using A = typename PredA<T, PRE>::type;
#define PLD(ptr, comp) ((A)(*(const T*)((const char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T)))))
#define PLD_D(f_, dx_, dy_, dz_) do { \
    if (a.Jc) { dx_ = PLD(a.Jc, (3 * (f_) + 1) * 18 + 5); dy_ = PLD(a.Jc, (3 * (f_) + 2) * 18 + 3); dz_ = PLD(a.Jc, (3 * (f_)) * 18 + 4); } \
    else { dx_ = PLD(a.ws, WS_D + 3 * (f_)); dy_ = PLD(a.ws, WS_D + 3 * (f_) + 1); dz_ = PLD(a.ws, WS_D + 3 * (f_) + 2); } } while (0)
  const int mask = a.mask[s32] & 0xF;
  const A zf0 = pp.zf0, zf1 = pp.zf1, zf2 = pp.zf2, zm0 = pp.zm0, zm1 = pp.zm1, zm2 = pp.zm2;
  {
#define PST(ptr, comp, val) (*(T*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))) = (T)(val))
#pragma unroll 2
    for (int f = 0; f < 4; ++f) {
      const bool on = (mask >> f) & 1;
      A dx, dy, dz;
      PLD_D(f, dx, dy, dz);
      const A fx = on ? zf0 + (zm1 * dz - zm2 * dy) : (A)0, fy = on ? zf1 + (zm2 * dx - zm0 * dz) : (A)0, fz = on ? zf2 + (zm0 * dy - zm1 * dx) : (A)0;
      PST(a.f, 3 * f, fx); PST(a.f, 3 * f + 1, fy); PST(a.f, 3 * f + 2, fz);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int jm = jmap.j[3 * f + k];
        const A taup = PLD(a.ws, WS_TAUP + 3 * f + k) - (RHAT ? PLD(a.ws, WS_RHAT + 6 + 3 * f + k) : (A)0);
        A j0, j1, j2;
        if (a.Jc) { j0 = PLD(a.Jc, (3 * f + 0) * 18 + 6 + jm); j1 = PLD(a.Jc, (3 * f + 1) * 18 + 6 + jm); j2 = PLD(a.Jc, (3 * f + 2) * 18 + 6 + jm); }
        else { j0 = PLD(a.ws, WS_JCL + 9 * f + k); j1 = PLD(a.ws, WS_JCL + 9 * f + 3 + k); j2 = PLD(a.ws, WS_JCL + 9 * f + 6 + k); }
        PST(a.tau, jm, taup - (j0 * fx + j1 * fy + j2 * fz));
      }
    }
    a.status[s32] = 0;
    if (a.iters) a.iters[s32] = 0;
    if (a.aset_out) a.aset_out[s32] = 0;   // the unconstrained minimum: the empty active set
#undef PST
  }
#undef PLD_D
#undef PLD
}

// fitted on the bench data (least squares on the iteration count): 0.52 count + 0.70 ln(1 + summed violation); three
// buckets per predicted iteration.  Sorting by it: 2.96 trips per group (count alone 3.25, perfect knowledge 2.51).
WBC_DEV int qp_predict_keyval(int cnt_all, float mag) {
  const float kf = 1.56f * (float)cnt_all + 2.1f * __logf(1.0f + mag);
  return (kf > 0.0f) ? (int)fminf(kf, 61.0f) : 0;   // 0 ... 61; a NaN / Inf state (garbage in) sorts as "no work", never out of range
}

// the whole predictor by one lane (tiles of more than 64 states): key 0 ... 61, or -1 = finished here
template <class T, bool RHAT, bool PRE = false>
WBC_DEV int qp_predict_key(const DevParams<T>& prm, const QpArgs<T>& a, const QpJidx& jmap, unsigned s32, unsigned N32, double* pre = nullptr, long long* st_part = nullptr) {
  constexpr bool FIN = (std::is_same<T, float>::value);
  PredPart<typename PredA<T, PRE>::type> pp;
  qp_predict_part<T, RHAT, PRE>(prm, a, s32, N32, pre, 0xF, 0x3F, pp);
#ifdef WBC_TILE_STAMP
  if (st_part) { asm volatile("" :: "v"(pp.cnt), "v"(pp.mag)); *st_part = __builtin_readcyclecounter(); }
#endif
  if (FIN && pp.fin_ok) { qp_predict_finish<T, RHAT, PRE>(prm, a, jmap, s32, N32, pp); return -1; }
  return qp_predict_keyval(pp.cnt, (float)pp.mag);
}

constexpr int WBC_QP_TILE_WAVES = 2;

// DENSE (fp32 only): the orthogonal-factor body in fp32 arithmetic instead of the structured body in fp64 arithmetic.  At 117 VGPRs
// four workgroups share a CU where the structured body's 183 allow two: from ~49 152 fp32 states on (more than two tiles of 64 per
// CU) the dense body wins (65 536: 38 vs 46 us, 131 072: 72 vs 92 us), below it loses (32 768: 26.0 vs 23.4 us).  Measured, MI355X.
template <class T, bool RHAT, int TILE, bool DENSE = false>
__global__ __launch_bounds__(256, (DENSE ? 4 : WBC_QP_TILE_WAVES)) void qp_tile_kernel(DevParams<T> prm, QpArgs<T> a, QpJidx jmap) {
  static_assert(TILE % 4 == 0 && TILE <= 1024, "tile of whole four-state groups");
  constexpr bool PRE = (std::is_same<T, double>::value) && !DENSE && TILE <= 64 && (true || (false && std::is_same<T, double>::value));
  __shared__ double pre[PRE ? TILE * QP_PRE_WORDS : 1];
  __shared__ unsigned short order[TILE];
  __shared__ int hist[64];
  __shared__ int next_grp;
  const unsigned tid = threadIdx.x;
  const size_t N = a.N;
  const unsigned N32 = (unsigned)a.N;
  const size_t base = (size_t)blockIdx.x * TILE;
  if (tid < 64) hist[tid] = 0;
  if (tid == 0) next_grp = 0;
#ifdef WBC_TILE_STAMP   // diagnostic build (tools/tile_stamp.py): per-wavefront cycle stamps of a tile go out in place of its iteration counts
  const long long ts_c0 = __builtin_readcyclecounter();
  const long long ts_w0 = wall_clock64();
  long long ts_b1 = 0, ts_b3 = 0, ts_g1 = 0, ts_end = 0, ts_pp = 0;
  int ts_ng = 0;
#define TSTAMP(x) x = __builtin_readcyclecounter()
#else
#define TSTAMP(x) do {} while (0)
#endif
  __syncthreads();
  // 1. keys: bucket 0 = most predicted work ... 61 = none; 62 = finished by the predictor, or beyond the end (not dealt)
  int bucket[(TILE + 255) / 256], rank[(TILE + 255) / 256];
  // (measured and not kept: sharing one state's predictor work among the four wavefronts -- feet / columns of G^-1 per wavefront after the
  //  common 6x6 factor, partial counts combined through LDS -- 31.5 vs 31.1 us at 32 768 fp64 states, 25.2 vs 24.4 at 24 576: the kernel is
  //  bound by instructions issued, not by the stretch in which three wavefronts wait for the first, and the shared stem is issued four times)
#pragma unroll
  for (int r = 0; r < (TILE + 255) / 256; ++r) {
    const unsigned i = tid + 256u * r;
    bucket[r] = 62; rank[r] = 0;
    if (i < TILE) {
      const size_t s = base + i;
      if (s < N) {
#ifdef WBC_TILE_STAMP
        const int key = qp_predict_key<T, RHAT, PRE>(prm, a, jmap, (unsigned)s, N32, pre + (PRE ? i * QP_PRE_WORDS : 0), &ts_pp);
#else
        const int key = qp_predict_key<T, RHAT, PRE>(prm, a, jmap, (unsigned)s, N32, pre + (PRE ? i * QP_PRE_WORDS : 0));
#endif
        bucket[r] = key < 0 ? 62 : 61 - key;
      }
      rank[r] = __hip_atomic_fetch_add(&hist[bucket[r]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  TSTAMP(ts_b1);
  __syncthreads();
  // 2. counting sort: position = states in harder buckets + my arrival rank in mine
#pragma unroll
  for (int r = 0; r < (TILE + 255) / 256; ++r) {
    const unsigned i = tid + 256u * r;
    if (i < TILE) {
      int pos = rank[r];
      for (int j = 0; j < bucket[r]; ++j) pos += hist[j];
      order[pos] = (unsigned short)i;
    }
  }
  __syncthreads();
  TSTAMP(ts_b3);
  // 3. the four wavefronts pull groups of four states, hardest first
  const int row = (int)((tid & 63) >> 4);
  const int nsolve = TILE - hist[62];
  for (;;) {
    int g = 0;
    if ((tid & 63) == 0) g = __hip_atomic_fetch_add(&next_grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    g = __builtin_amdgcn_readfirstlane(g);
    if (4 * g >= nsolve) break;
    const bool live = 4 * g + row < nsolve;
    const unsigned oi = order[live ? 4 * g + row : 0];
    const size_t s = base + oi;
    if constexpr (DENSE) qp_group16_body<T, false, RHAT, 16, true>(prm, a, jmap, nullptr, nullptr, QpWho{live ? s : (size_t)0, live});
    else qp_body<T, false, RHAT, 16, true, 4, QpNoIdle, PRE>(prm, a, jmap, nullptr, nullptr, QpWho{live ? s : (size_t)0, live, pre + (PRE ? oi * QP_PRE_WORDS : 0)});
#ifdef WBC_TILE_STAMP
    if (ts_ng == 0) TSTAMP(ts_g1);
    ++ts_ng;
#endif
  }
#ifdef WBC_TILE_STAMP
  TSTAMP(ts_end);
  const long long ts_w1 = wall_clock64();
  __syncthreads();
  if (a.iters && TILE >= 64 && (tid & 63) == 0 && base + TILE <= N) {
    int* o = a.iters + base + 8 * (tid >> 6);
    o[0] = (int)(ts_b1 - ts_c0); o[1] = (int)(ts_b3 - ts_c0); o[2] = ts_ng ? (int)(ts_g1 - ts_c0) : 0; o[3] = (int)(ts_end - ts_c0);
    o[4] = ts_ng | ((int)((ts_pp ? ts_pp - ts_c0 : 0) >> 4) << 8); o[5] = (int)(ts_w0 & 0x7FFFFFFF); o[6] = (int)(ts_w1 & 0x7FFFFFFF); o[7] = nsolve;
  }
#endif
#undef TSTAMP
}

// ======================================================================================================================
// qp_stile_kernel (round 6): the tile kernel with the tile's inputs STAGED THROUGH LDS.
// What the timeline of qp_tile_kernel showed at configs[3]'s shard (32 768 fp32 states, observer on; tools/tile_stamp.py,
// profiles/r06a_tile_timeline.txt): the predictor's decision is there at +2.6 us, but the states it finishes keep its ONE wavefront busy
// until +8.4 us (per-lane loads of the Jacobian blocks, two feet per round) while three wavefronts wait at the barrier; every group of
// four states then pays two dependent rounds of global loads (inputs in front of the set-up, the torque map's Jacobian blocks behind the
// iteration), each a GATHER: the sixteen lanes of a row read sixteen different component rows of one state, a whole 64-byte sector per
// 4-byte word (19 MB fetched from Jc per launch).
// Here the workgroup first copies everything its states need -- normals, mu, lever arms, b = w_des - rhat_base, tau_partial - rhat_joint,
// the own-leg Jacobian blocks: 82 words per state -- from HBM into an LDS image, row by row, a 256-byte line per load instruction, all
// loads of a wavefront in flight together: ONE memory latency per tile and every byte fetched once.  Predictor, finisher and the
// row-form body then work out of LDS (qp_struct16_body<..., STG>), results go back into the image, and the workgroup stores f, tau, status,
// iters and the active sets row by row at the end.  The predictor runs one FOOT per thread (thread = foot x state): the 6 x 6 factor is
// computed redundantly, the foot's minimiser and slacks once; a thread finishes its own foot of a state whose four verdicts say "x0
// violates nothing" from the values it still holds.
// NW wavefronts per workgroup, tiles of up to 64 CH states (`tile`, a multiple of 4, run-time): the host sizes the tile so that the launch
// is ONE round of resident workgroups -- at 32 768 states one workgroup of 128 states per CU, whose NW wavefronts all pull groups from
// the one queue (two 64-state tiles per CU level only within each tile: the slower tile of a CU set the pace).
// Same keys, same order of the groups within a tile, same arithmetic per state as qp_tile_kernel.
constexpr int STILE_IN_ROWS = 82;   // staged input rows per state (ST_F: the results follow)
constexpr size_t stile_lds_bytes(int tile, size_t scalar, int nw) {
  return (size_t)nw * sizeof(S16Lds<double>) + (size_t)ST_WORDS * (size_t)(tile | 1) * scalar + (size_t)tile * (4 * 4 + 4 * 4 + 4 * scalar + 2) + 64 * 4 * 2 + 16;
}
// the stage as a function of the workgroup's dynamic LDS (`smem`, stile_lds_bytes bytes, 16-byte aligned) and of this wavefront's index among the NW that
// run it: the stand-alone kernel below, and the second half of tile_tick_kernel (tile_tick.hip.hpp).  blk = the tile's index.
// FIN: the predictor finishes the states whose unconstrained minimum violates nothing (default: WBC_QP_PRED_FINISH's rule -- fp32 solvers; the fp64 tile tick
// turns it on too: here the predictor is spread over all wavefronts and one foot per thread, not the serial stretch of one wavefront it was in qp_tile_kernel)
constexpr bool stile_fin_default(bool is_float) { return (is_float); }
// HAND (tile tick): tau_partial and, with RHAT, the observer's estimate are NOT in the memory workspace: the role wavefronts of this workgroup left them in LDS
// (`hand`: [HAND_ROWS][tile] words outside this stage's `smem`, dyn_sweep.hip.hpp)
template <class T, bool RHAT, int NW, int CH, bool FIN = stile_fin_default(std::is_same<T, float>::value), bool HAND = false>
WBC_DEV void qp_stile_body(const DevParams<T>& prm, const QpArgs<T>& a, const QpJidx& jmap, int tile, unsigned blk, unsigned char* smem, unsigned wave_in, const T* hand = nullptr) {
  using A = T;   // (the caller sizes the tile: 4 tile <= 64 NW, one predictor thread per foot and state)
  const int ST = tile | 1;   // row stride of the image: the sixteen lanes of a row read sixteen ROWS of one column -- an odd stride spreads them over the banks
  S16Lds<double>* const tabs = (S16Lds<double>*)smem;   // [NW] solver tables, one per wavefront
  T* const img = (T*)(tabs + NW);
  A* const pmag = (A*)(img + ST_WORDS * ST);            // [4][tile] per foot: summed violation at x0
  int* const iimg = (int*)(pmag + 4 * tile);            // [4][tile] mask | status | iters | active set
  int* const pcnt = iimg + 4 * tile;                    // [4][tile] per foot: violated rows at x0 (bit 8: some slack of the foot is below the finishing threshold)
  int* const hist = pcnt + 4 * tile;                    // [64]
  int* const next_grp = hist + 64 + 64;                 // (64 spare words: stile_lds_bytes)
  unsigned short* const order = (unsigned short*)(next_grp + 4);
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_in);
  const unsigned tid = wave * 64 + lane;
  const size_t N = a.N;
  const unsigned N32 = (unsigned)a.N;
  const size_t base = (size_t)blk * (size_t)tile;
  if (tid < 64) hist[tid] = 0;
  if (tid == 0) *next_grp = 0;
#ifdef WBC_TILE_STAMP
  const long long ts_c0 = __builtin_readcyclecounter();
  const long long ts_w0 = wall_clock64();
  long long ts_b1 = 0, ts_b3 = 0, ts_g1 = 0, ts_end = 0, ts_pp = 0;
  int ts_ng = 0;
#define TSTAMP(x) x = __builtin_readcyclecounter()
#else
#define TSTAMP(x) do {} while (0)
#endif
  // ---- 0. stage in.  Unit u = (staging row su, chunk ch of 64 columns); wavefront W takes the units u = j NW + W: every wavefront requests all its units,
  // then parks them.  One straight-line copy of the code per wavefront (rows, and so the source arrays and components, are compile-time there).
  // Staging order: b (6 rows) and tau_partial (12) first -- the rows the observer estimate is subtracted from -- then normals, mu, lever arms, Jacobian blocks.
  {
    const bool jc = a.Jc != nullptr;
    const T* gsrc = jc ? a.Jc : a.ws;   // geometry: from the Jacobian the sweep wrote, or from the workspace (ticks without M, h, Jc)
    const T* bsrc = a.wdes ? a.wdes : a.ws;
    const unsigned boff = a.wdes ? 0u : (unsigned)WS_B;
    sfor<0, NW>([&](auto w_) __attribute__((always_inline)) {
      constexpr int W = decltype(w_)::value;
      if (wave != (unsigned)W) return;
      constexpr int UNITS = STILE_IN_ROWS * CH, UPW = (UNITS - W + NW - 1) / NW;
      T val[UPW], sub[UPW];
      int mk[CH];
      sfor<0, UPW>([&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        constexpr unsigned u = (unsigned)j * NW + W, su = u / CH, ch = u % CH;
        const unsigned col = ch * 64 + lane;
        const unsigned sidx = (unsigned)(base + col < N ? base + col : N - 1);
        const T* ptr; unsigned comp;
        if constexpr (su < 6) { ptr = bsrc; comp = boff + su; }
        else if constexpr (su < 18) { ptr = a.ws; comp = (unsigned)WS_TAUP + (su - 6); }
        else if constexpr (su < 30) { ptr = a.normals; comp = su - 18; }
        else if constexpr (su < 34) { ptr = a.mu; comp = su - 30; }
        else if constexpr (su < 46) {
          constexpr unsigned v = su - 34, f = v / 3, c = v - 3 * f;
          ptr = gsrc;
          comp = jc ? (c == 0 ? (3 * f + 1) * 18 + 5 : (c == 1 ? (3 * f + 2) * 18 + 3 : (3 * f) * 18 + 4)) : (unsigned)WS_D + v;
        } else {
          constexpr unsigned r = su - 46, f = r / 9, mm = (r - 9 * f) / 3, k = r - 9 * f - 3 * mm;
          const unsigned jm = (unsigned)((a.jpack >> (4 * (3 * f + k))) & 15u);
          ptr = gsrc;
          comp = jc ? (3 * f + mm) * 18 + 6 + jm : (unsigned)WS_JCL + r;
        }
        const unsigned hcol = col < (unsigned)tile ? col : (unsigned)tile - 1u;
        if constexpr (HAND && su >= 6 && su < 18) val[j] = hand[((unsigned)HAND_TAUP + su - 6) * (unsigned)tile + hcol];
        else val[j] = *(const T*)((const char*)ptr + (size_t)((comp * N32 + sidx) * (unsigned)sizeof(T)));
        if constexpr (RHAT && su < 18) {
          if constexpr (HAND) sub[j] = hand[((unsigned)HAND_RHAT + su) * (unsigned)tile + hcol];
          else sub[j] = *(const T*)((const char*)a.ws + (size_t)((((unsigned)WS_RHAT + su) * N32 + sidx) * (unsigned)sizeof(T)));
        } else sub[j] = 0;
      });
      if constexpr (W == NW - 1) sfor<0, CH>([&](auto c_) __attribute__((always_inline)) {
        constexpr int ch = decltype(c_)::value;
        const unsigned col = ch * 64 + lane;
        mk[ch] = a.mask[base + col < N ? base + col : N - 1];
      });
      sfor<0, UPW>([&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        constexpr unsigned u = (unsigned)j * NW + W, su = u / CH, ch = u % CH;
        constexpr unsigned row = su < 18 ? (unsigned)ST_B + su : (su < 46 ? su - 18 : su);
        const unsigned col = ch * 64 + lane;
        if (col < (unsigned)tile) img[row * ST + col] = val[j] - sub[j];
      });
      if constexpr (W == NW - 1) sfor<0, CH>([&](auto c_) __attribute__((always_inline)) {
        constexpr int ch = decltype(c_)::value;
        const unsigned col = ch * 64 + lane;
        if (col < (unsigned)tile) iimg[col] = base + col < N ? mk[ch] : 0;
      });
    });
  }
  __syncthreads();
  // ---- 1. predictor: thread t = (foot t / tile, column t mod tile) for t < 4 tile; the arithmetic of qp_predict_part out of the image
  const unsigned pfoot = tid / (unsigned)tile, pcol = tid - pfoot * (unsigned)tile;
  const bool incol = pfoot < 4u;
  const bool pred = __ballot(incol) != 0ull;          // (wavefronts beyond the 4 tile tasks only solve)
  const bool valid = incol && base + pcol < N;
  const unsigned colr = incol ? pcol : 0u;
  const int fo = (int)(incol ? pfoot : 0u);
#define IMGR(row) ((A)img[(row) * ST + colr])
#define IMGW(row) img[(row) * ST + pcol]
  A x0 = 0, x1 = 0, x2 = 0;       // my foot's part of the unconstrained minimum
  A fjl[9], ftp[3];               // my foot's Jacobian block and tau_partial rows: read in front of the verdict barrier, used behind it by the threads that finish their foot
  if (pred) {
    if constexpr (FIN) {
#pragma unroll
      for (int k = 0; k < 9; ++k) fjl[k] = IMGR(ST_JCL + 9 * fo + k);
#pragma unroll
      for (int k = 0; k < 3; ++k) ftp[k] = IMGR(ST_TAUP + 3 * fo + k);
    }
    const int mask = iimg[colr] & 0xF;
    const A s0 = prm.sS[0], s1 = prm.sS[1], s2 = prm.sS[2], s3 = prm.sS[3], s4 = prm.sS[4], s5 = prm.sS[5];
    A nc = 0, sx = 0, sy = 0, sz = 0, Pxx = 0, Pxy = 0, Pxz = 0, Pyy = 0, Pyz = 0, Pzz = 0;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const bool on = (mask >> f) & 1;
      A dx = IMGR(ST_D + 3 * f), dy = IMGR(ST_D + 3 * f + 1), dz = IMGR(ST_D + 3 * f + 2);
      dx = on ? dx : (A)0; dy = on ? dy : (A)0; dz = on ? dz : (A)0;
      nc += on ? (A)1 : (A)0; sx += dx; sy += dy; sz += dz;
      Pxx += dx * dx; Pxy += dx * dy; Pxz += dx * dz; Pyy += dy * dy; Pyz += dy * dz; Pzz += dz * dz;
    }
    A bt[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) bt[k] = IMGR(ST_B + k);
    const A g00 = prm.alpha + s0 * s0 * nc, g11 = prm.alpha + s1 * s1 * nc, g22 = prm.alpha + s2 * s2 * nc;
    const A gm01 = -(s3 * s1) * sz, gm02 = (s3 * s2) * sy, gm10 = (s4 * s0) * sz, gm12 = -(s4 * s2) * sx, gm20 = -(s5 * s0) * sy, gm21 = (s5 * s1) * sx;
    const A m00 = prm.alpha + (s3 * s3) * (Pyy + Pzz), m11 = prm.alpha + (s4 * s4) * (Pxx + Pzz), m22 = prm.alpha + (s5 * s5) * (Pxx + Pyy);
    const A m10 = -(s4 * s3) * Pxy, m20 = -(s5 * s3) * Pxz, m21 = -(s5 * s4) * Pyz;
    auto rs = [](A x) __attribute__((always_inline)) -> A {
      if constexpr (FIN) return rsqrt_nr(x);
      else if constexpr (std::is_same<A, double>::value) return __builtin_amdgcn_rsq(x); else return __builtin_amdgcn_rsqf(x);
    };
    A il[6];
    il[0] = rs(g00); il[1] = rs(g11); il[2] = rs(g22);
    const A a01 = gm01 * il[1], a02 = gm02 * il[2], a10 = gm10 * il[0], a12 = gm12 * il[2], a20 = gm20 * il[0], a21 = gm21 * il[1];
    const A c00 = m00 - a01 * a01 - a02 * a02, c11 = m11 - a10 * a10 - a12 * a12, c22 = m22 - a20 * a20 - a21 * a21;
    const A c10 = m10 - a12 * a02, c20 = m20 - a21 * a01, c21 = m21 - a20 * a10;
    il[3] = rs(c00);
    const A b10 = c10 * il[3], b20 = c20 * il[3];
    il[4] = rs(c11 - b10 * b10);
    const A b21 = (c21 - b20 * b10) * il[4];
    il[5] = rs(c22 - b20 * b20 - b21 * b21);
    A w[6], z[6];   // z = G^-1 S^(1/2) b
    w[0] = s0 * bt[0] * il[0]; w[1] = s1 * bt[1] * il[1]; w[2] = s2 * bt[2] * il[2];
    w[3] = (s3 * bt[3] - a01 * w[1] - a02 * w[2]) * il[3];
    w[4] = (s4 * bt[4] - a10 * w[0] - a12 * w[2] - b10 * w[3]) * il[4];
    w[5] = (s5 * bt[5] - a20 * w[0] - a21 * w[1] - b20 * w[3] - b21 * w[4]) * il[5];
    z[5] = w[5] * il[5];
    z[4] = (w[4] - b21 * z[5]) * il[4];
    z[3] = (w[3] - b10 * z[4] - b20 * z[5]) * il[3];
    z[2] = (w[2] - a02 * z[3] - a12 * z[4]) * il[2];
    z[1] = (w[1] - a01 * z[3] - a21 * z[5]) * il[1];
    z[0] = (w[0] - a10 * z[4] - a20 * z[5]) * il[0];
    const A zf0 = s0 * z[0], zf1 = s1 * z[1], zf2 = s2 * z[2], zm0 = s3 * z[3], zm1 = s4 * z[4], zm2 = s5 * z[5];
    // my foot: x0 = on (s_f z_f + (s_m z_m) x d), its six slacks
    const bool on = (mask >> fo) & 1;
    const A dx = IMGR(ST_D + 3 * fo), dy = IMGR(ST_D + 3 * fo + 1), dz = IMGR(ST_D + 3 * fo + 2);
    x0 = on ? zf0 + (zm1 * dz - zm2 * dy) : (A)0;
    x1 = on ? zf1 + (zm2 * dx - zm0 * dz) : (A)0;
    x2 = on ? zf2 + (zm0 * dy - zm1 * dx) : (A)0;
    A nx = IMGR(ST_N + 3 * fo), ny = IMGR(ST_N + 3 * fo + 1), nz = IMGR(ST_N + 3 * fo + 2);
    const A iln = rs(nx * nx + ny * ny + nz * nz);
    nx *= iln; ny *= iln; nz *= iln;
    const bool usex = fabs_t(nx) < (A)0.9;
    const A rx = usex ? (A)1 : (A)0, ry = usex ? (A)0 : (A)1;
    const A rd = rx * nx + ry * ny;
    A t1x = rx - nx * rd, t1y = ry - ny * rd, t1z = -nz * rd;
    const A it = rs(t1x * t1x + t1y * t1y + t1z * t1z);
    t1x *= it; t1y *= it; t1z *= it;
    const A t2x = ny * t1z - nz * t1y, t2y = nz * t1x - nx * t1z, t2z = nx * t1y - ny * t1x;
    const A fn = nx * x0 + ny * x1 + nz * x2, f1 = t1x * x0 + t1y * x1 + t1z * x2, f2 = t2x * x0 + t2y * x1 + t2z * x2;
    const A mf = IMGR(ST_MU + fo) * prm.mu_scale * fn, tol = -prm.qp_tol;
    const A sl[6] = {mf - f1, mf + f1, mf - f2, mf + f2, fn - prm.fn_min, prm.fn_max - fn};
    const A fin_thr = std::is_same<T, double>::value ? (A)-prm.qp_tol : (A)1e-3;
    int cnt = 0;
    A mg = 0;
    bool okf = true;
#pragma unroll
    for (int c = 0; c < 6; ++c) { cnt += (sl[c] < tol) ? 1 : 0; mg += (sl[c] < tol) ? -sl[c] : (A)0; okf = okf && (sl[c] >= fin_thr); }
    if (incol) {
      pcnt[fo * tile + pcol] = on ? (okf ? cnt : (cnt | 0x100)) : 0;    // (a NaN state fails every comparison: never finished here)
      pmag[fo * tile + pcol] = on ? mg : (A)0;
    }
  }
  TSTAMP(ts_pp);
  __syncthreads();
  // ---- 2. the state's verdict (every thread of the column); a thread finishes its foot, the threads of foot 0 file the key
  int bucket = 62, rank = 0;
  if (pred) {
    const int c0 = pcnt[colr], c1 = pcnt[tile + colr], c2 = pcnt[2 * tile + colr], c3 = pcnt[3 * tile + colr];
    const bool fin = FIN && valid && ((c0 | c1 | c2 | c3) & 0x100) == 0;
    if (fin) {   // f = x0, tau = tau_partial - Jc_leg^T f of my foot's leg; status 0, no iterations, the empty set
      IMGW(ST_F + 3 * fo) = (T)x0; IMGW(ST_F + 3 * fo + 1) = (T)x1; IMGW(ST_F + 3 * fo + 2) = (T)x2;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int jm = (int)((a.jpack >> (4 * (3 * fo + k))) & 15u);
        IMGW(ST_TAU + jm) = (T)(ftp[k] - (fjl[k] * x0 + fjl[3 + k] * x1 + fjl[6 + k] * x2));
      }
      if (fo == 0) { iimg[tile + pcol] = 0; iimg[2 * tile + pcol] = 0; iimg[3 * tile + pcol] = 0; }
    }
    if (fo == 0 && incol) {
      if (valid && !fin) {
        const A mag = ((pmag[colr] + pmag[tile + colr]) + pmag[2 * tile + colr]) + pmag[3 * tile + colr];
        bucket = 61 - qp_predict_keyval((c0 & 0xFF) + (c1 & 0xFF) + (c2 & 0xFF) + (c3 & 0xFF), (float)mag);
      }
      rank = __hip_atomic_fetch_add(&hist[bucket], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  TSTAMP(ts_b1);
  __syncthreads();
  // ---- 3. counting sort: position = states in harder buckets + my arrival rank in mine.  Every wavefront that holds threads of foot 0 scans the 64-bin histogram
  // itself (lane j = bin j, shuffles) and each thread fetches its bucket's exclusive prefix from the lane of that bin: no table, no barrier in between
  if (__ballot(incol && fo == 0) != 0ull) {
    const int h = hist[lane];
    int incl = h;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); incl += (int)lane >= d ? t : 0; }
    const int base_of_mine = __shfl(incl - h, bucket);
    if (incol && fo == 0) order[base_of_mine + rank] = (unsigned short)pcol;
  }
  __syncthreads();
  TSTAMP(ts_b3);
  // ---- 4. the wavefronts pull groups of four states, hardest first
  const int row = (int)(lane >> 4);
  const int nsolve = tile - hist[62];
  for (;;) {
    int g = 0;
    if (lane == 0) g = __hip_atomic_fetch_add(next_grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    g = __builtin_amdgcn_readfirstlane(g);
    if (4 * g >= nsolve) break;
    const bool live = 4 * g + row < nsolve;
    const unsigned oi = order[live ? 4 * g + row : 0];
    qp_body<T, false, false, 16, true, NW, QpNoIdle, false, 0, 1>(prm, a, jmap, nullptr, nullptr, QpWho{0, live, nullptr, (void*)img, iimg, (int)oi, ST, tile, (void*)(tabs + wave)});
#ifdef WBC_TILE_STAMP
    if (ts_ng == 0) TSTAMP(ts_g1);
    ++ts_ng;
#endif
  }
  TSTAMP(ts_end);
  __syncthreads();
  // ---- 5. stage out: f, tau row by row (unit = (row, chunk), u = j NW + wave); status, iters, active sets
  {
    constexpr int OUNITS = 27 * CH, OPW = (OUNITS + NW - 1) / NW;
    sfor<0, OPW>([&](auto j_) __attribute__((always_inline)) {
      constexpr int j = decltype(j_)::value;
      const unsigned u = (unsigned)j * NW + wave;
      const unsigned r = u / CH, ch = u % CH;
      const unsigned col = ch * 64 + lane;
      if (r < 27u && col < (unsigned)tile && base + col < N) {
        const unsigned sidx = (unsigned)(base + col);
        if (r < 12) *(T*)((char*)a.f + (size_t)((r * N32 + sidx) * (unsigned)sizeof(T))) = img[(ST_F + r) * ST + col];
        else if (r < 24) *(T*)((char*)a.tau + (size_t)(((r - 12) * N32 + sidx) * (unsigned)sizeof(T))) = img[(ST_TAU + r - 12) * ST + col];
        else if (r == 24) a.status[sidx] = iimg[tile + col];
#ifdef WBC_TILE_STAMP   // (the stamp build returns its stamps through `iters`)
        else if (r == 25) {}
#else
        else if (r == 25) { if (a.iters) a.iters[sidx] = iimg[2 * tile + col]; }
#endif
        else if (r == 26) { if (a.aset_out) a.aset_out[sidx] = iimg[3 * tile + col]; }
      }
    });
  }
#ifdef WBC_TILE_STAMP
  {
    const long long ts_w1 = wall_clock64();
    __syncthreads();
    if (a.iters && tile >= 8 * NW && lane == 0 && base + tile <= N) {
      int* o = a.iters + base + 8 * wave;
      o[0] = (int)(ts_b1 - ts_c0); o[1] = (int)(ts_b3 - ts_c0); o[2] = ts_ng ? (int)(ts_g1 - ts_c0) : 0; o[3] = (int)(ts_end - ts_c0);
      o[4] = ts_ng | ((int)((ts_pp - ts_c0) >> 4) << 8); o[5] = (int)(ts_w0 & 0x7FFFFFFF); o[6] = (int)(ts_w1 & 0x7FFFFFFF); o[7] = nsolve;
    }
  }
#endif
#undef TSTAMP
#undef IMGR
#undef IMGW
}
template <class T, bool RHAT, int NW, int CH>
__global__ __launch_bounds__(64 * NW, (NW > 8 ? 3 : 2)) void qp_stile_kernel(DevParams<T> prm, QpArgs<T> a, QpJidx jmap, int tile) {
  extern __shared__ __attribute__((aligned(16))) unsigned char stile_dyn[];
  qp_stile_body<T, RHAT, NW, CH>(prm, a, jmap, tile, blockIdx.x, stile_dyn, threadIdx.x >> 6);
}

// qp_list_kernel: the dense active-set solver over a LIST of states (list[0] = how many, list[4 ...] = their indices): the
// states qp_lane_kernel (qp_lane.hip.hpp) did not finish.  One wavefront per workgroup, four listed states per wavefront,
// grid-stride over the list (the launch cannot know its length: it lives on the device).  list[2] keeps the length for
// wbc_solver_qp_handover.  The count is zeroed by one thread of the tick's front-half kernel (dyn_sweep / rnea_step, which
// run between this kernel and the next qp_lane_kernel on the stream).  What did not work:
//   * a 4-byte hipMemsetAsync did the job in eager mode, but as a memset node of a captured hipGraph it did not reliably run
//     in front of the next kernel on this stack (the list overflowed after a few replays);
//   * letting the last workgroup of this kernel reset it (one agent-scope atomic per workgroup to count them, an agent-scope
//     load of the length) serialises on that one address: 8192 workgroups took 330 us instead of 65 us (N = 262 144);
//   * a one-thread kernel of its own: correct, 4.7 us per tick.
// WARM (behind the WARM per-lane kernel): the listed states start from their carried active sets too -- the states the Newton iteration gives
// up on are mostly the same from tick to tick, and their sets are the ones this kernel reported the tick before
template <class T, bool RHAT, bool WARM = false>
__global__ __launch_bounds__(64, WBC_QP_WAVES) void qp_list_kernel(DevParams<T> prm, QpArgs<T> a, QpJidx jmap, int* __restrict__ list) {
  const int n = min(list[0], (int)a.N);
  const int ngroups = (n + 3) >> 2;
  const int row = (int)((threadIdx.x & 63) >> 4);
  if (blockIdx.x == 0 && threadIdx.x == 0) list[2] = n;
  for (int g = (int)blockIdx.x; g < ngroups; g += (int)gridDim.x) {
    const int i = 4 * g + row;
    const bool live = i < n;
    qp_body<T, false, RHAT, 16, true, 1, QpNoIdle, false, WARM ? 1 : 0>(prm, a, jmap, nullptr, nullptr, QpWho{live ? (size_t)list[4 + i] : (size_t)0, live});
  }
}

}  // namespace wbc

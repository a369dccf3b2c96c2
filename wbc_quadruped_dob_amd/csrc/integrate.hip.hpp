// Forward dynamics with the planned GRFs + semi-implicit Euler: SURVEY.md 8(f)-1, the step Gazebo performs in the
// reference loop (/root/reference/README.md:58), restated as the simplest model that closes the loop for rollouts:
//     vdot = M^-1 (S^T tau + Jc^T f + tau_ext - h),   v += dt vdot,   q <- q (+) dt v   (world-frame omega)
//
// Same lane mapping as the dynamics sweep (lane = 16*leg + state; one leg per 16-lane row).  The quadruped's mass
// matrix is an arrow: a 6x6 base block, four 6x3 base-leg blocks and four independent 3x3 leg blocks.  Each lane
// inverts ITS leg block in closed form, forms its part of the base Schur complement (21 + 6 words summed across
// the four rows with v_permlane16/32_swap), every lane solves the 6x6 base system redundantly in registers, and
// back-substitutes its own leg.  M, h, Jc come from the buffers the sweep wrote (foot lever arms = the base-angular
// columns of Jc, own-leg Jacobian blocks = its joint columns).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_types.hpp"
#include "dyn_sweep.hip.hpp"

namespace wbc {

// (also called as one wavefront of the persistent rollout kernel, fused_tick.hip.hpp: same mapping, no LDS.)
// Two phases.  Phase 1 needs only M and Jc: the leg blocks' inverses, the base Schur complement and its Cholesky factor.
// `between()` runs after it (nothing in the stand-alone kernel; the tick barrier in the rollout kernel, where phase 1
// overlaps the QP).  Phase 2 needs tau, f, h, q, v: right-hand sides, the two triangular solves, the state update.
struct IntegrateNoWait { WBC_DEV void operator()() const {} };
// `hand` (persistent rollout): M's leg / base-leg / base blocks, the own-leg Jacobian block and the lever arm come from the LDS image
// the mass_jac role left (dyn_split.hip.hpp, MJ_HAND_WORDS) instead of from the M / Jc buffers through L2 -- same numbers, so the
// results are bit-identical; what changes is when phase 1 can start and how long its operands take to arrive.
// `res` (persistent rollout, round 5): this tick's tau (rows 0 .. 11), f (12 .. 23), h (24 .. 41) in an LDS image [42][16] written by the QP wavefronts and
// the rnea role: phase 2 starts behind an LDS-only barrier instead of behind the acknowledgement of their global stores and a trip through L2.
// PHASE (persistent rollout, round 5): 0 = both phases on this wavefront.  1 = phase 1 ONLY, run by the mass_jac wavefront itself right behind its
// image (no flag, no second wavefront waiting for it), leaving the leg block's inverse (6 words) and the base factor (21) in the LDS image
// fact[27][64]; 2 = phase 2 only, on the integrator wavefront: state loads, the tick barrier, then M's blocks from `hand` and the factors from `fact`.
// The integrator wavefront used to run the observer's joint rows, THEN phase 1, and arrived last at the tick barrier (+10.3 us; profiles/r05f_*).
constexpr int INT_FACT_WORDS = 27;
// UNGUARD (persistent rollout): the state stores carry no `if (live)`.  Lanes beyond the batch or beyond the workgroup's states recompute a state of THEIR OWN
// wavefront (the workgroup's first, or the batch's last, which the last workgroup owns), in lockstep with the lane that owns it: what they store is a
// bit-identical duplicate at the same address, and every one of the 22 guards was an exec region of ~45 cycles in the one phase of a rollout tick during
// which nothing else runs (knock-out analysis: profiles/r05h_ro_knock.log).  Not for the stand-alone kernel: there a dead lane may sit in another wavefront.
// WBC_INT_SINV: 1 = phase 1 (which runs beside the QP: hidden) ends with the explicit inverse of the base Schur complement, and phase 2 multiplies by it --
// 36 independent multiply-adds instead of two triangular solves of 54 dependent operations at 13 cycles each for a lone wavefront; 0 (default) = the solves.
// measured (profiles/r05i_ab_rollout_phase2_*.log): 12.05 -> 12.22 us per tick in fp64, and the fp32 inverse loses the accuracy the
// fp32 rollout tests ask for (NaN on stiff states): not kept
// HAND / RESI: `hand` / `res` are given (template parameters, not null tests: a pointer that may be an LDS image or null turns every load behind it into a
// flat_load -- 39 of them in the round-4 rollout kernels, 30 in phase 2 -- where the image wants a ds_read).
// `after_state()` (4-state rollout workgroups): called when the new state is complete in the LDS image and before anything goes to memory -- the kernel
// can put the tick's barrier there, so that the next tick's roles start while this wavefront (the next tick's QP, idle until the lever arms are out)
// still issues its stores.  Measured (profiles/r05q_ab_rollout_early_barrier.log): 9.39-9.47 -> 9.48-9.52 us per tick at 1 024 robots, cold 15.71 -> 15.97:
// not kept (-DWBC_RO_EARLY_BARRIER=1).
template <class T, int SPW = 16, class Between = IntegrateNoWait, int PHASE = 0, bool UNGUARD = false, bool HAND = false, bool RESI = false,
          class AfterState = IntegrateNoWait, bool SIMG_ = false>
WBC_DEV void integrate_body(const DevModel<T>* __restrict__ model, const IntegrateArgs<T>& a, Between between = Between(), const T* hand_ = nullptr,
                            const T* res_ = nullptr, T* fact = nullptr, AfterState after_state = AfterState()) {
  static_assert(PHASE == 0 || PHASE == 1 || PHASE == 2, "phase");
  static_assert(PHASE == 0 || HAND, "the split phases hand M's blocks over in LDS");
  constexpr bool FASTR = PHASE != 0 && SIMG_;   // (rollout workgroups with the state image) rsqrt_fast, see dyn_sweep.hip.hpp
  const T* const hand = HAND ? hand_ : nullptr;
  const T* const res = RESI ? res_ : nullptr;
  const size_t N = a.N;
  const unsigned N32 = (unsigned)N;
  unsigned tx = threadIdx.x;
  asm volatile("" : "+v"(tx));   // see WBC_LAUNDERED_TID (dyn_split.hip.hpp)
  const int leg = (int)((tx & 63) >> 4);
  const size_t s_raw = (size_t)blockIdx.x * SPW + (tx & 15);
  const bool slot_ok = SPW == 16 || (int)(tx & 15) < SPW;   // (SPW <= 16 states per workgroup, see WBC_ADDR_MACROS)
  const bool live = slot_ok && s_raw < N;
  const unsigned s32 = (unsigned)(live ? s_raw : (slot_ok ? N - 1 : (size_t)blockIdx.x * SPW));
  const unsigned legN = (unsigned)leg * N32;
#define LDU(ptr, comp) (*(const T*)((const char*)((ptr) + (size_t)(comp) * N) + (size_t)(s32 * (unsigned)sizeof(T))))
#define LDV(ptr, comp) (*(const T*)((const char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))))
#define LDL(ptr, c0, stride) (*(const T*)((const char*)((ptr) + (size_t)(c0) * N) + (size_t)(((unsigned)(stride) * legN + s32) * (unsigned)sizeof(T))))
#define LDLX(ptr, c0, stride, xN) (*(const T*)((const char*)((ptr) + (size_t)(c0) * N) + (size_t)(((unsigned)(stride) * legN + (xN) + s32) * (unsigned)sizeof(T))))
#define STV(ptr, comp, val) do { if (UNGUARD || live) *(T*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))) = (val); } while (0)
  int jx[3];
  jidx_of_leg(model, a.jpack, leg, jx);
#ifdef WBC_FUSED_STAMP
#define ISTAMP(slot) do { if (a.istamp && SPW < 16 && (tx & 63) == 0) a.istamp[(size_t)(slot) * a.istampN + (size_t)blockIdx.x * SPW + 1] = (double)wall_clock64(); } while (0)
#else
#define ISTAMP(slot) do {} while (0)
#endif
  ISTAMP(0);

  // ================================================================== phase 1: M, Jc only
  const T* hl = HAND ? hand + (int)(tx & 63) : nullptr;
  T* const fl_ = PHASE != 0 ? fact + (int)(tx & 63) : nullptr;
  V3<T> dl;
  T jcl[3][3];   // own-leg Jacobian block: jcl[m][k] = d pf_m / d q_(leg, k)
  T Mb[6][3];    // base-leg block (6x3) of M
  T A[3][3];     // inverse of the leg block
  T L[6][6];     // Cholesky factor of the base Schur complement (L[j][j] holds 1 / L_jj)
  auto load_blocks = [&]() __attribute__((always_inline)) {
    // foot position relative to the base origin from the base-angular block of my foot's Jacobian rows, -[d]x
    dl = HAND ? mk<T>(hl[33 * 64], hl[34 * 64], hl[35 * 64])
              : mk<T>(LDL(a.Jc, 18 * 1 + 5, 54), LDL(a.Jc, 18 * 2 + 3, 54), LDL(a.Jc, 18 * 0 + 4, 54));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned jo = (unsigned)(6 + jx[k]) * N32;   // column of joint (leg, k) in rows 3*leg + m of Jc
      if constexpr (HAND) { jcl[0][k] = hl[(24 + k) * 64]; jcl[1][k] = hl[(27 + k) * 64]; jcl[2][k] = hl[(30 + k) * 64]; }
      else { jcl[0][k] = LDLX(a.Jc, 0, 54, jo); jcl[1][k] = LDLX(a.Jc, 18, 54, jo); jcl[2][k] = LDLX(a.Jc, 36, 54, jo); }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) Mb[r][k] = HAND ? hl[(6 + 3 * r + k) * 64] : LDV(a.M, midx18(r, r) + (6 + jx[k] - r));
  };
  if constexpr (PHASE != 2) {
  load_blocks();
  // leg block (symmetric 3x3) of M
  auto mi = [](int i, int j) { if (i > j) { const int t = i; i = j; j = t; } return i * 18 - i * (i - 1) / 2 + (j - i); };
  T Ml[3][3];
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
    for (int k2 = k1; k2 < 3; ++k2) {
      Ml[k1][k2] = HAND ? hl[(k1 * 3 - k1 * (k1 - 1) / 2 + (k2 - k1)) * 64] : LDV(a.M, mi(6 + jx[k1], 6 + jx[k2]));
      Ml[k2][k1] = Ml[k1][k2];
    }
  // base block of M (upper triangle): from the buffer, or rebuilt from (m, R h, R I R^T) of the hand-over image
  T Mbb[6][6];
  if constexpr (HAND) {
    const T tm = hl[36 * 64], hx = hl[37 * 64], hy = hl[38 * 64], hz = hl[39 * 64];
    const T Z = (T)0;
    Mbb[0][0] = tm; Mbb[0][1] = Z; Mbb[0][2] = Z; Mbb[0][3] = Z; Mbb[0][4] = hz; Mbb[0][5] = -hy;
    Mbb[1][1] = tm; Mbb[1][2] = Z; Mbb[1][3] = -hz; Mbb[1][4] = Z; Mbb[1][5] = hx;
    Mbb[2][2] = tm; Mbb[2][3] = hy; Mbb[2][4] = -hx; Mbb[2][5] = Z;
    Mbb[3][3] = hl[40 * 64]; Mbb[3][4] = hl[41 * 64]; Mbb[3][5] = hl[42 * 64]; Mbb[4][4] = hl[43 * 64]; Mbb[4][5] = hl[44 * 64]; Mbb[5][5] = hl[45 * 64];
  } else {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = r; c < 6; ++c) Mbb[r][c] = LDU(a.M, midx18(r, c));
  }
  ISTAMP(1);   // image read requested
  // A = Ml^-1 by cofactors (SPD 3x3)
  {
    const T c00 = Ml[1][1] * Ml[2][2] - Ml[1][2] * Ml[1][2];
    const T c01 = Ml[0][2] * Ml[1][2] - Ml[0][1] * Ml[2][2];
    const T c02 = Ml[0][1] * Ml[1][2] - Ml[0][2] * Ml[1][1];
    const T c11 = Ml[0][0] * Ml[2][2] - Ml[0][2] * Ml[0][2];
    const T c12 = Ml[0][1] * Ml[0][2] - Ml[0][0] * Ml[1][2];
    const T c22 = Ml[0][0] * Ml[1][1] - Ml[0][1] * Ml[0][1];
    const T idet = (T)1 / (Ml[0][0] * c00 + Ml[0][1] * c01 + Ml[0][2] * c02);
    A[0][0] = c00 * idet; A[0][1] = A[1][0] = c01 * idet; A[0][2] = A[2][0] = c02 * idet;
    A[1][1] = c11 * idet; A[1][2] = A[2][1] = c12 * idet; A[2][2] = c22 * idet;
  }
  ISTAMP(2);   // leg block inverted
  // base Schur complement S = Mbb - sum_legs W Mb^T with W = Mb A (6x3), one row of W at a time
  T S[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    T Wr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) Wr[k] = Mb[r][0] * A[0][k] + Mb[r][1] * A[1][k] + Mb[r][2] * A[2][k];
#pragma unroll
    for (int c = r; c < 6; ++c) {
      const T sc = Wr[0] * Mb[c][0] + Wr[1] * Mb[c][1] + Wr[2] * Mb[c][2];
      S[r][c] = Mbb[r][c] - xrow_sum(sc);
    }
  }
  ISTAMP(3);   // Schur complement summed over the legs
  // Cholesky of S (L[j][j] holds 1 / L_jj), in registers
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    T d = S[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    const T inv = rsqrt_sel<FASTR>(d);
    L[j][j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      T sij = S[j][i];
#pragma unroll
      for (int k = 0; k < j; ++k) sij -= L[i][k] * L[j][k];
      L[i][j] = sij * inv;
    }
  }

  if constexpr (PHASE == 1) {   // hand the factors to the integrator wavefront (which reads them behind the tick barrier) and return
    fl_[0 * 64] = A[0][0]; fl_[1 * 64] = A[0][1]; fl_[2 * 64] = A[0][2]; fl_[3 * 64] = A[1][1]; fl_[4 * 64] = A[1][2]; fl_[5 * 64] = A[2][2];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) fl_[(6 + i * (i + 1) / 2 + j) * 64] = L[i][j];
    return;
  }
  }   // PHASE != 2
  ISTAMP(4);   // Cholesky done
  // the state of this tick (q, v: inputs of the tick, untouched until the stores at the end) is requested BEFORE the tick barrier: the loads
  // complete while the wavefront waits there instead of after it
  // (4-state rollout workgroups, WBC_RO_MERGE: the state lives in the workgroup's LDS image -- device_types.hpp, SIMG_* -- and the new one goes to both)
  constexpr bool SIMG = PHASE == 2 && SIMG_;
  T* const si_ = SIMG ? a.simg + (int)(s32 - (unsigned)((size_t)blockIdx.x * SPW)) : nullptr;
#define STS(comp, val) do { if constexpr (SIMG) si_[(comp) * 16] = (val); } while (0)
  constexpr bool DEFER = SIMG && !std::is_same<AfterState, IntegrateNoWait>::value;   // (see after_state)
  const bool to_mem = !(SIMG && a.skip_state != 0);   // (wavefront-uniform: a kernel argument) the state in memory is read by nobody before the launch ends
  T ql[3], vl[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { ql[k] = SIMG ? si_[(7 + jx[k]) * 16] : LDV(a.q, 7 + jx[k]); vl[k] = SIMG ? si_[(SIMG_V + 6 + jx[k]) * 16] : LDV(a.v, 6 + jx[k]); }
  T vb0[6], qb[7];
#pragma unroll
  for (int c = 0; c < 6; ++c) vb0[c] = SIMG ? si_[(SIMG_V + c) * 16] : LDU(a.v, c);
#pragma unroll
  for (int c = 0; c < 7; ++c) qb[c] = SIMG ? si_[c * 16] : LDU(a.q, c);

  // (so is the unit quaternion of the state: a square root and a division of ~0.2 us for a lone wavefront, which used to sit behind the barrier)
  // Three shortenings of phase 2 (round 5), each 1 = fp32 only (default), 2 = both scalar types, 0 = off.  Measured at 1 024 rollouts
  // (profiles/r05n_ab_rollout_phase2.log): fp32 10.08 -> 9.82 us per tick with all three; fp64 11.57 -> 11.75 / 11.68 / 11.88 as they are added -- the fp64
  // rollout kernel sits at its 256 registers and each of them moves spill code INTO this phase (ISA: 11 scratch loads behind the barrier), so fp64 keeps
  // the forms of rounds 1-4.
// W rl = Mb (A rl) instead of forming W = Mb A again behind the barrier
// the quaternion increment from its power series (below)
// the state's unit quaternion (a square root and a division) in front of the barrier
  constexpr bool ARL = (sizeof(T) == 4);
  constexpr bool QUAT_SERIES = (sizeof(T) == 4);
  constexpr bool QNORM_EARLY = (sizeof(T) == 4);
  T ux, uy, uz, uw;
  if constexpr (QNORM_EARLY) {
    const T n = rsqrt_sel<FASTR>(qb[3] * qb[3] + qb[4] * qb[4] + qb[5] * qb[5] + qb[6] * qb[6]);
    ux = qb[3] * n; uy = qb[4] * n; uz = qb[5] * n; uw = qb[6] * n;
  }

  between();
  if constexpr (!QNORM_EARLY) {
    const T n = rsqrt_sel<FASTR>(qb[3] * qb[3] + qb[4] * qb[4] + qb[5] * qb[5] + qb[6] * qb[6]);
    ux = qb[3] * n; uy = qb[4] * n; uz = qb[5] * n; uw = qb[6] * n;
  }
  ISTAMP(5);   // barrier passed
  // (the image is indexed by the slot of the state a lane COMPUTES: a lane beyond the workgroup's states duplicates state s32, and with UNGUARD it stores)
  const T* rs = RESI ? res + (int)(s32 - (unsigned)((size_t)blockIdx.x * SPW)) : nullptr;
  if constexpr (PHASE == 2) {   // M's blocks from the mass_jac role's image, the factors from the image phase 1 left (both complete behind the barrier)
    load_blocks();
    A[0][0] = fl_[0 * 64]; A[0][1] = A[1][0] = fl_[1 * 64]; A[0][2] = A[2][0] = fl_[2 * 64]; A[1][1] = fl_[3 * 64]; A[1][2] = A[2][1] = fl_[4 * 64]; A[2][2] = fl_[5 * 64];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) L[i][j] = fl_[(6 + i * (i + 1) / 2 + j) * 64];
  }
  // the external torques: rows RES_WORDS .. RES_WORDS + 17 of the image hold them for the whole launch (parked once by the rollout kernel: fetched per
  // tick in front of the barrier they cost 18 registers across it -- and scratch), or the caller's buffer
  auto text = [&](int comp) __attribute__((always_inline)) -> T { if constexpr (RESI) return rs[(42 + comp) * 16]; else return a.tau_ext ? LDV(a.tau_ext, comp) : (T)0; };

  // ================================================================== phase 2: tau, f, h, q, v
  // ---- my leg: rhs_l = tau_l + JcL^T f_l + tau_ext_l - h_l
  const V3<T> fl = RESI ? mk<T>(rs[(12 + 3 * leg + 0) * 16], rs[(12 + 3 * leg + 1) * 16], rs[(12 + 3 * leg + 2) * 16])
                      : mk<T>(LDL(a.f, 0, 3), LDL(a.f, 1, 3), LDL(a.f, 2, 3));
  T rl[3], taul[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    taul[k] = RESI ? rs[jx[k] * 16] : LDV(a.tau, jx[k]);
    const T hl_k = RESI ? rs[(24 + 6 + jx[k]) * 16] : LDV(a.h, 6 + jx[k]);
    rl[k] = taul[k] + jcl[0][k] * fl.x + jcl[1][k] * fl.y + jcl[2][k] * fl.z - hl_k + text(6 + jx[k]);
  }
  ISTAMP(6);   // tau, f, h arrived: leg right-hand side
  // ---- base right-hand side rb = rhs_b - sum_legs W rl
  T rb[6];
  {
    const V3<T> mo = cross(dl, fl);
    const T own[6] = {fl.x, fl.y, fl.z, mo.x, mo.y, mo.z};
    T ul[3];   // W rl = Mb (A rl): 27 multiply-adds behind the barrier instead of the 72 of forming W = Mb A again
#pragma unroll
    for (int k = 0; k < 3; ++k) ul[k] = A[k][0] * rl[0] + A[k][1] * rl[1] + A[k][2] * rl[2];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      T part;
      if constexpr (ARL) part = own[r] - (Mb[r][0] * ul[0] + Mb[r][1] * ul[1] + Mb[r][2] * ul[2]);
      else {
        T Wr[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) Wr[k] = Mb[r][0] * A[0][k] + Mb[r][1] * A[1][k] + Mb[r][2] * A[2][k];
        part = own[r] - (Wr[0] * rl[0] + Wr[1] * rl[1] + Wr[2] * rl[2]);
      }
      rb[r] = xrow_sum(part) - (RESI ? rs[(24 + r) * 16] : LDU(a.h, r)) + text(r);
    }
  }
  ISTAMP(7);   // base right-hand side summed
  // ---- base accelerations: S vb = rb
  T vb[6];
  {
    T y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      T sy = rb[i];
#pragma unroll
      for (int k = 0; k < i; ++k) sy -= L[i][k] * y[k];
      y[i] = sy * L[i][i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      T sx = y[i];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) sx -= L[k][i] * vb[k];
      vb[i] = sx * L[i][i];
    }
  }
  ISTAMP(8);   // solves done
  // ---- leg accelerations: vdl = A (rl - Mb^T vb)
  T tl[3], vdl[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    T t = rl[k];
#pragma unroll
    for (int r = 0; r < 6; ++r) t -= Mb[r][k] * vb[r];
    tl[k] = t;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) vdl[k] = A[k][0] * tl[0] + A[k][1] * tl[1] + A[k][2] * tl[2];

  // ---- semi-implicit Euler
  const T dt = a.dt;
  T vjn[3], qjn[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    vjn[k] = vl[k] + dt * vdl[k];
    qjn[k] = ql[k] + dt * vjn[k];
    STS(SIMG_V + 6 + jx[k], vjn[k]);
    STS(7 + jx[k], qjn[k]);
    if constexpr (!DEFER) {
      if (to_mem) {
        STV(a.v, 6 + jx[k], vjn[k]);
        STV(a.q, 7 + jx[k], qjn[k]);
      }
      if (a.tau_traj) STV(a.tau_traj, jx[k], taul[k]);
    }
  }
  ISTAMP(9);   // joint rows stored
  T vbn[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) vbn[c] = vb0[c] + dt * vb[c];
  T qn[7];
#pragma unroll
  for (int c = 0; c < 3; ++c) qn[c] = qb[c] + dt * vbn[c];
  {
    // q <- dq(omega dt) * q with dq = [sin(th/2) / th * w ; cos(th/2)], w = omega dt, th = |w|.  Both factors are even power series in th: with
    // u = th^2 / 4 <= 1/16 (half a radian per step) seven terms carry them to below 1e-19 -- no square root, no division, no argument reduction, which
    // together were ~1 us of the 3 us a rollout tick spends in this phase alone (profiles/r05g_rollout_timeline_spw4.txt).  Beyond that the closed form,
    // selected per lane: what a state gets does not depend on its neighbours.
    const T wx = vbn[3] * dt, wy = vbn[4] * dt, wz = vbn[5] * dt;
    const T th2 = wx * wx + wy * wy + wz * wz;
    const T u = th2 * (T)0.25;
    T sc = (T)(1.0 / 6227020800.0), dw = (T)(1.0 / 479001600.0);
    sc = sc * u - (T)(1.0 / 39916800.0); dw = dw * u - (T)(1.0 / 3628800.0);
    sc = sc * u + (T)(1.0 / 362880.0);   dw = dw * u + (T)(1.0 / 40320.0);
    sc = sc * u - (T)(1.0 / 5040.0);     dw = dw * u - (T)(1.0 / 720.0);
    sc = sc * u + (T)(1.0 / 120.0);      dw = dw * u + (T)(1.0 / 24.0);
    sc = sc * u - (T)(1.0 / 6.0);        dw = dw * u - (T)0.5;
    sc = sc * u + (T)1;                  dw = dw * u + (T)1;
    sc = sc * (T)0.5;
    const bool big = QUAT_SERIES ? !(u <= (T)0.0625) : true;
    if (!QUAT_SERIES || __ballot(big) != 0) {   // the closed form (with its own two-term series next to th = 0, as in rounds 1-4)
      const T th = th2 > (T)0 ? th2 * rsqrt_sel<FASTR>(th2) : (T)0;
      T sn, cs;
      sincos_t(th * (T)0.5, &sn, &cs);
      const bool small = th <= (T)1e-8;
      const T sc_c = small ? (T)0.5 - th2 * (T)(1.0 / 48.0) : sn / (small ? (T)1 : th);
      const T dw_c = small ? (T)1 - th2 * (T)0.125 : cs;
      sc = big ? sc_c : sc;
      dw = big ? dw_c : dw;
    }
    const T dx = sc * wx, dy = sc * wy, dz = sc * wz;
    const T x = ux, y = uy, z = uz, w = uw;
    qn[3] = dw * x + dx * w + dy * z - dz * y;
    qn[4] = dw * y - dx * z + dy * w + dz * x;
    qn[5] = dw * z + dx * y - dy * x + dz * w;
    qn[6] = dw * w - dx * x - dy * y - dz * z;
  }
  // the 13 base words are replicated over the four leg rows: every row stores its share.  All rows have read
  // q/v base rows above (their values feed these stores), so no lane can store before every lane has loaded.
  ISTAMP(10);   // quaternion advanced
  if constexpr (SIMG) {   // (every row holds all 13 base words: row `leg` writes its share, as to memory)
    STS(SIMG_V + sel4<int>(leg, 0, 1, 2, 3), sel4<T>(leg, vbn[0], vbn[1], vbn[2], vbn[3]));
    if (leg < 2) STS(SIMG_V + 4 + leg, leg == 0 ? vbn[4] : vbn[5]);
    STS(sel4<int>(leg, 0, 1, 2, 3), sel4<T>(leg, qn[0], qn[1], qn[2], qn[3]));
    if (leg < 3) STS(4 + leg, sel4<T>(leg, qn[4], qn[5], qn[6], qn[6]));
  }
  if constexpr (DEFER) {
    after_state();   // the image is complete: memory comes behind
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (to_mem) {
        STV(a.v, 6 + jx[k], vjn[k]);
        STV(a.q, 7 + jx[k], qjn[k]);
      }
      if (a.tau_traj) STV(a.tau_traj, jx[k], taul[k]);
    }
  }
  if (to_mem) {
    STV(a.v, sel4<int>(leg, 0, 1, 2, 3), sel4<T>(leg, vbn[0], vbn[1], vbn[2], vbn[3]));
    if (leg < 2) STV(a.v, 4 + leg, leg == 0 ? vbn[4] : vbn[5]);
    STV(a.q, sel4<int>(leg, 0, 1, 2, 3), sel4<T>(leg, qn[0], qn[1], qn[2], qn[3]));
    if (leg < 3) STV(a.q, 4 + leg, sel4<T>(leg, qn[4], qn[5], qn[6], qn[6]));
  }
  ISTAMP(11);   // base rows stored
#undef ISTAMP
#undef STS
#undef STV
#undef LDLX
#undef LDL
#undef LDV
#undef LDU
}

template <class T>
__global__ __launch_bounds__(64) void integrate_kernel(const DevModel<T>* __restrict__ model, IntegrateArgs<T> a) {
  integrate_body<T>(model, a);
}

}  // namespace wbc

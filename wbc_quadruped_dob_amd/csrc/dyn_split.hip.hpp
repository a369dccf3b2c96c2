// rnea_step_kernel<RS_STEP...> is also the DEFAULT front half of a tick whose caller does not ask for M, h, Jc
// (no CRBA at all: tau_partial comes from the merged force recursion).  The two-kernel split for ticks that DO
// want M, h, Jc is an
// EXPERIMENT (WBC_SWEEP=split; NOT the default): the dynamics sweep as TWO kernels that run concurrently on two
// HIP streams.  Measured on MI355X it is 20 % slower per tick at N = 4 096 and 5 % slower at N = 262 144 than the
// fused dyn_sweep kernel: at small batch each half pays the same fixed latencies (table staging, state loads,
// sincos chains, store drain) as the fused kernel, and at large batch the duplicated kinematics and the second
// force recursion cost more than the overlap with the QP kernel returns.  Kept because it is parity-green and
// documents the negative result.
//
//   mass_jac_kernel   (a1, a3, a4)  q        -> M, Jc, pf      store-bound: 387 of the 443 words of the stage
//   rnea_step_kernel  (a1, a2, a5, a6, a7/a9 prologue)  q, v, vdot_des, w_des -> h, step workspace, observer state
//
// Why split: the GRF QP only needs the small workspace (foot positions, own-leg Jacobian blocks, target wrench,
// tau_partial), none of M / Jc.  With one fused sweep the QP waited for 3.5 kB/state of stores it never reads;
// now the QP kernel is queued behind rnea_step only, and the store-heavy mass_jac kernel overlaps with it on a
// second stream (the QP is latency-bound and leaves HBM idle).  Each half also needs far fewer registers than the
// fused sweep (no scratch, more waves per SIMD), and tau_partial = (M vdot_des + h) comes from a second force
// recursion in rnea_step instead of from the M entries, so vdot_des is consumed in the first sweep, not loaded late.
//
// Same lane-per-leg mapping (leg-major rows), permlane row reductions, LDS constant table and addressing as
// dyn_sweep.hip.hpp (the fused form, kept selectable with WBC_SWEEP=fused).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_types.hpp"
#include "dyn_sweep.hip.hpp"  // V3/M3/S3 helpers, quad_sum, sel4, midx18, sincos_t

namespace wbc {


// The work-item id passes through an empty asm at the top of every body: inside the persistent rollout kernel the bodies
// sit in the horizon loop, and without this every lane-derived predicate, LDS address and table index (hundreds of
// registers' worth) is hoisted out of that loop as "invariant" and spilled.  One no-op per call elsewhere.
#define WBC_LAUNDERED_TID(name) unsigned name = threadIdx.x; asm volatile("" : "+v"(name))
#define WBC_ADDR_MACROS                                                                                                   \
  const size_t N = a.N;                                                                                                    \
  const unsigned N32 = (unsigned)N;                                                                                        \
  /* lane = 16*leg + (state within the wave), as in dyn_sweep_kernel */                                                    \
  const int leg = (int)((tx & 63) >> 4);                                                                          \
  /* roles (EXT != 0): a workgroup owns SPW <= 16 consecutive states; lanes of the other slots repeat its first state */   \
  const size_t s_raw = EXT ? (size_t)blockIdx.x * SPW + (tx & 15)                                                 \
                           : ((size_t)blockIdx.x * (BLOCK / 64) + (tx >> 6)) * 16 + (tx & 15);           \
  const bool slot_ok = SPW == 16 || (int)(tx & 15) < SPW;                                                                  \
  const bool live = slot_ok && s_raw < N;                                                                                  \
  unsigned s32 = (unsigned)(live ? s_raw : (slot_ok ? N - 1 : (size_t)blockIdx.x * SPW));                                  \
  const unsigned legN = (unsigned)leg * N32;
// the state (q, v): from memory, or -- roles of the rollout workgroups that keep their states on chip, EXT = 3 -- from the workgroup's LDS image (device_types.hpp, SIMG_*),
// indexed by the slot of the state the lane COMPUTES (dead lanes duplicate a state of their own workgroup).  Needs jx / jxN in scope for the joint rows.
#define WBC_STATE_MACROS                                                                                                                          \
  constexpr bool SIMG = EXT == 3;   /* role of a rollout workgroup that keeps its states in LDS (EXT = 3: like 1, plus the image) */                  \
  const T* const si_ = SIMG ? a.simg + (int)(s32 - (unsigned)((size_t)blockIdx.x * SPW)) : nullptr;                                               \
  auto ldq = [&](int comp) __attribute__((always_inline)) -> T { if constexpr (SIMG) return si_[comp * 16]; else return LDU(a.q, comp); };         \
  auto ldv = [&](int comp) __attribute__((always_inline)) -> T { if constexpr (SIMG) return si_[(SIMG_V + comp) * 16]; else return LDU(a.v, comp); }; \
  (void)ldq; (void)ldv;
#define LDQJ(k_) (SIMG ? si_[(7 + jx[k_]) * 16] : LDX(a.q, 7, jxN[k_]))
#define LDVJ(k_) (SIMG ? si_[(SIMG_V + 6 + jx[k_]) * 16] : LDX(a.v, 6, jxN[k_]))
#define CS(i) cst[(i) * 4 + leg]
#define LDU(ptr, comp) (*(const T*)((const char*)((ptr) + (size_t)(comp) * N) + (size_t)(s32 * (unsigned)sizeof(T))))
#define LDV(ptr, comp) (*(const T*)((const char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))))
#define LDX(ptr, c0, xN) (*(const T*)((const char*)((ptr) + (size_t)(c0) * N) + (size_t)(((xN) + s32) * (unsigned)sizeof(T))))
// (pure-output stores keep their `if (live)` guard here: without it the roles save ~1 % in the fused tick but the persistent rollout
// kernel, which sits at 256 registers, spills -- 19.5 -> 24.0 us per tick, measured)
// -DWBC_ROLE_UNGUARD=1 (A/B): pure-output stores of the ROLES (EXT != 0) without their guard -- a dead lane of a role duplicates a state of its own wavefront
#define WBC_ROLE_LIVE ((0 && EXT != 0) || live)
#define STV(ptr, comp, val) do { if (WBC_ROLE_LIVE) *(T*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))) = (val); } while (0)
#define STVG(ptr, comp, val) do { if (live) *(T*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))) = (val); } while (0)   /* in/out state (observer): dead lanes of OTHER wavefronts would race with the live one */
#define STL(ptr, c0, stride, val) do { if (WBC_ROLE_LIVE) *(T*)((char*)((ptr) + (size_t)(c0) * N) + (size_t)(((unsigned)(stride) * legN + s32) * (unsigned)sizeof(T))) = (val); } while (0)
#define STLX(ptr, c0, stride, xN, val) do { if (WBC_ROLE_LIVE) *(T*)((char*)((ptr) + (size_t)(c0) * N) + (size_t)(((unsigned)(stride) * legN + (xN) + s32) * (unsigned)sizeof(T))) = (val); } while (0)
#define ST4(ptr, c0, v0_, c1, v1_, c2, v2_, c3, v3_) STV(ptr, sel4<int>(leg, c0, c1, c2, c3), sel4<T>(leg, v0_, v1_, v2_, v3_))
#define ST4G(ptr, c0, v0_, c1, v1_, c2, v2_, c3, v3_) STVG(ptr, sel4<int>(leg, c0, c1, c2, c3), sel4<T>(leg, v0_, v1_, v2_, v3_))
#define MAKE_R(R_, qx, qy, qz, qw) do { const T x = qx, y = qy, z = qz, w = qw; \
    R_.a[0] = 1 - 2 * (y * y + z * z); R_.a[1] = 2 * (x * y - z * w);     R_.a[2] = 2 * (x * z + y * w); \
    R_.a[3] = 2 * (x * y + z * w);     R_.a[4] = 1 - 2 * (x * x + z * z); R_.a[5] = 2 * (y * z - x * w); \
    R_.a[6] = 2 * (x * z - y * w);     R_.a[7] = 2 * (y * z + x * w);     R_.a[8] = 1 - 2 * (x * x + y * y); } while (0)

// (A/B) 1: the one-launch tick's rnea role normalises the quaternion with rsqrt_fast: lever arms ~0.2 us earlier
constexpr int WBC_MJ_WAVES = 2;
constexpr int WBC_RS_WAVES = 2;

// ======================================================================================================================
// mass_jac_kernel: M(q) by CRBA, Jc(q), pf(q).  No velocities anywhere.
// ======================================================================================================================
// EXT != 0 (fused_tick.hip.hpp): the body is ONE wavefront of a larger workgroup that owns 16 states and the constant
// tables are staged by other wavefronts of the workgroup (cst_ext, zidx_ext).  EXT = 1: they are complete on entry, the
// body contains no barrier.  EXT = 2: the body issues its state loads first and THEN joins the workgroup barrier behind
// which the tables are complete (one memory round trip instead of two at the head of the tick).
// Structural zeros / ones of M and Jc (54 + 6 packed-M entries, 39 Jc entries per foot: ~55 store instructions per wavefront of
// 16 states -- ~4 us of store issue when ONE wavefront does them, as mass_jac_body does).  Inside the fused tick the four QP
// wavefronts are idle until the lever arms arrive: each takes a quarter of these stores there (wavefronts 0..2 one row of every
// foot's Jc block, wavefront 3 the zeros of M), and the mass_jac role (ZEROS = false) is left with the data-dependent entries.
// tx = thread index within the four QP wavefronts (0..255): state slot tx & 15, leg (tx >> 4) & 3, quarter tx >> 6.
template <class T>
WBC_DEV void structural_consts_quarter(const DevModel<T>* __restrict__ model, const SweepArgs<T>& a, const int* zidx_s, unsigned tx) {
  const size_t N = a.N;
  const unsigned N32 = (unsigned)N;
  const int leg = (int)((tx >> 4) & 3), part = (int)(tx >> 6);
  const size_t s_raw = (size_t)blockIdx.x * 16 + (tx & 15);
  const bool live = s_raw < N;
  const unsigned s32 = (unsigned)(live ? s_raw : N - 1);
  const T Z = (T)0;
  int jq[3];
  jidx_of_leg(model, a.jpack, leg, jq);
  const int j0 = jq[0], j1 = jq[1], j2 = jq[2];
  if ((N & 1) == 0) {
    // 16 bytes per lane: neighbouring lanes pair up, the even one takes the even-numbered constants, the odd one the odd-numbered
    // ones, each for both states (as in dyn_sweep_kernel): half the store instructions
    struct alignas(2 * sizeof(T)) T2 { T a, b; };
    const unsigned odd = s32 & 1u, s2 = s32 & ~1u;
#define ST2C(ptr, comp, val) do { if (live) *(T2*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s2) * (unsigned)sizeof(T))) = T2{(val), (val)}; } while (0)
    if (part == 3) {
      for (int e = 2 * leg + (int)odd; e < 64; e += 8) {
        const int zi = zidx_s[e];
        if (zi >= 0) ST2C(a.M, zi, Z);
      }
    } else {
      const int mrow = part;
      const int rb = 54 * leg + 18 * mrow;
      ST2C(a.Jc, rb + (odd ? 1 : 0), ((odd ? 1 : 0) == mrow) ? (T)1 : Z);
      ST2C(a.Jc, rb + (odd ? 3 + mrow : 2), (!odd && mrow == 2) ? (T)1 : Z);
#pragma unroll
      for (int c = 6; c < 18; c += 2) {
        const int col = c + (int)odd - 6;
        if (col != j0 && col != j1 && col != j2) ST2C(a.Jc, rb + c + (int)odd, Z);
      }
    }
#undef ST2C
  } else {
#define ST1C(ptr, comp, val) do { if (live) *(T*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))) = (val); } while (0)
    if (part == 3) {
      for (int e = leg; e < 64; e += 4) {
        const int zi = zidx_s[e];
        if (zi >= 0) ST1C(a.M, zi, Z);
      }
    } else {
      const int mrow = part;
      const int rb = 54 * leg + 18 * mrow;
#pragma unroll
      for (int c = 0; c < 3; ++c) ST1C(a.Jc, rb + c, (c == mrow) ? (T)1 : Z);
      ST1C(a.Jc, rb + 3 + mrow, Z);
#pragma unroll
      for (int c = 0; c < 12; ++c)
        if (c != j0 && c != j1 && c != j2) ST1C(a.Jc, rb + 6 + c, Z);
    }
#undef ST1C
  }
}

// `hand` (persistent rollout, round 4): the entries of M and Jc the integrator's factorisation needs ALSO go to an LDS image
// hand[word][64] (lane = 16 leg + state slot) -- leg block of M (6: upper triangle (k, j)), base-leg block (18: 6 + 3 r + k), own-leg
// Jacobian block (9: 24 + 3 m + k), lever arm (3: 33..35), base block as (m, R h, R I R^T) (10: 36..45) -- so that the integrator
// wavefront neither waits for this role's global stores to drain nor reloads them through L2.
// Round 5: with `hand` the image is COMPLETE (it holds every data-dependent word this role produces), so the role computes into the image only,
// calls after_hand() -- the caller raises the integrator's flag there -- and issues the ~50 global store instructions of M, Jc, pf AFTERWARDS, from
// the image; a tick whose M / Jc nobody will read (a.skip_mats: every tick of a persistent rollout but the last) does not issue them at all.  A wave
// store instruction costs ~90 cycles to issue whatever its width (tools/issue_probe.hip): they were ~2 us of the mass_jac -> factorisation chain.
constexpr int MJ_HAND_WORDS = 46;
struct MjNoHook { WBC_DEV void operator()() const {} };
template <class T, int BLOCK, int EXT, int SPW = 16, int ZEROS = 1, class AfterHand = MjNoHook>   // ZEROS: 0 = other wavefronts write the structural constants, 1 = first, 2 = LAST (behind the data: fused_tick.hip.hpp)
WBC_DEV void mass_jac_body(const DevModel<T>* __restrict__ model, const SweepArgs<T>& a, const T* cst_ext, const int* zidx_ext, T* hand = nullptr,
                           AfterHand after_hand = AfterHand()) {
  static_assert(!EXT || BLOCK == 64 || BLOCK == 128, "one wavefront (128: one of the two role wavefronts of a fused_pair_kernel workgroup, each with its own half of the parking lot)");
  static_assert(SPW == 16 || EXT != 0, "fewer states per workgroup only for roles");
  WBC_LAUNDERED_TID(tx);
  __shared__ __attribute__((aligned(512))) T cst_own[EXT ? 1 : CST_WORDS];   // (aligned: first in LDS, see section 4.9 of docs/DESIGN_R04.md)
  __shared__ int zidx_own[EXT ? 1 : 64];
  if constexpr (EXT == 0) {
    for (int i = tx; i < CST_WORDS; i += blockDim.x) cst_own[i] = model->cst[i];
    if (tx < 64) zidx_own[tx] = model->zidx[tx];
    __syncthreads();
  }
  const T* cst = EXT ? cst_ext : cst_own;
  const int* zidx_s = EXT ? zidx_ext : zidx_own;
  WBC_ADDR_MACROS
  WBC_STATE_MACROS
  const bool direct = hand == nullptr;   // (a literal at every call site: folded)

  T qq[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) qq[c] = ldq(3 + c);
  int jx[3];
  unsigned jxN[3];
  jidx_of_leg(model, a.jpack, leg, jx);
#pragma unroll
  for (int k = 0; k < 3; ++k) jxN[k] = (unsigned)jx[k] * N32;
  T ql[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) ql[k] = LDQJ(k);
  if constexpr (EXT == 2) __syncthreads();   // tables staged by the other wavefronts while my loads are in flight

  // structural zeros / ones first: they drain while the sweeps compute (ZEROS = 0: other wavefronts write them; 2: last, see the end of the body)
  auto write_consts = [&]() __attribute__((always_inline)) {
    const T Z = (T)0;
    for (int e = leg; e < 64; e += 4) {
      const int zi = zidx_s[e];
      if (zi >= 0) STV(a.M, zi, Z);
    }
#pragma unroll
    for (int mrow = 0; mrow < 3; ++mrow) {
#pragma unroll
      for (int c = 0; c < 3; ++c) STL(a.Jc, 18 * mrow + c, 54, (c == mrow) ? (T)1 : Z);
      STL(a.Jc, 18 * mrow + 3 + mrow, 54, Z);
#pragma unroll
      for (int c = 0; c < 12; ++c)
        if (c != jx[0] && c != jx[1] && c != jx[2]) STL(a.Jc, 18 * mrow + 6 + c, 54, Z);   // own-leg columns get data below
    }
  };
  if constexpr (ZEROS == 1) { if (!a.skip_consts) write_consts(); }
  T qx, qy, qz, qw;
  {
    const T n = rsqrt_sel<SIMG>(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
    qx = qq[0] * n; qy = qq[1] * n; qz = qq[2] * n; qw = qq[3] * n;
  }
  // joint transforms: E of joints 0 and 1 wait in LDS ([word][lane]) until the return sweep reaches them
  __shared__ T park[18][BLOCK];
  const int ln = EXT ? (int)(tx & (unsigned)(BLOCK - 1)) : (int)tx;
  T* const hl = hand ? hand + (int)(tx & 63) : nullptr;
  M3<T> E2;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int o = JOINT_WORDS * k;
    T sn, cs;
    sincos_t(ql[k], &sn, &cs);
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      const T v = CS(o + e) + cs * CS(o + 9 + e) + sn * CS(o + 18 + e);
      if (k < 2) park[9 * k + e][ln] = v; else E2.a[e] = v;
    }
  }
  // return sweep: composite inertias, CRBA force columns, Jacobian columns
  T cm; V3<T> ch; S3<T> cI;
  V3<T> dft = mk<T>(CS(129), CS(130), CS(131));
  V3<T> jc[3];
  SF<T> Fp[3];
#pragma unroll
  for (int k = 2; k >= 0; --k) {
    const int o = JOINT_WORDS * k;
    const V3<T> r = mk<T>(CS(o + 27), CS(o + 28), CS(o + 29));
    const V3<T> ax = mk<T>(CS(o + 30), CS(o + 31), CS(o + 32));
    const T m = CS(o + 33);
    const V3<T> h = mk<T>(CS(o + 34), CS(o + 35), CS(o + 36));
    S3<T> Io;
    Io.xx = CS(o + 37); Io.xy = CS(o + 38); Io.xz = CS(o + 39); Io.yy = CS(o + 40); Io.yz = CS(o + 41); Io.zz = CS(o + 42);
    M3<T> E;
    if (k == 2) E = E2;
    else {
#pragma unroll
      for (int e = 0; e < 9; ++e) E.a[e] = park[9 * k + e][ln];
    }
    if (k == 2) { cm = m; ch = h; cI = Io; }
    else {
      cm += m; ch = ch + h;
      cI.xx += Io.xx; cI.xy += Io.xy; cI.xz += Io.xz; cI.yy += Io.yy; cI.yz += Io.yz; cI.zz += Io.zz;
    }
    Fp[k].n = mul(cI, ax);
    Fp[k].f = cross(ax, ch);
#pragma unroll
    for (int j = k; j < 3; ++j) {
      const T mkj = dot(ax, Fp[j].n);
      if (direct) {
        int i = 6 + jx[k], jj = 6 + jx[j];
        if (i > jj) { const int t = i; i = jj; jj = t; }
        STV(a.M, i * 18 - i * (i - 1) / 2 + (jj - i), mkj);
      } else hl[(k * 3 - k * (k - 1) / 2 + (j - k)) * 64] = mkj;
    }
    jc[k] = cross(ax, dft);
    dft = r + mul(E, dft);
#pragma unroll
    for (int j = k; j < 3; ++j) { jc[j] = mul(E, jc[j]); Fp[j] = to_parent(E, r, Fp[j]); }
    {
      const V3<T> hr = mul(E, ch);
      const S3<T> Ir = congr(E, cI);
      const V3<T> w = hr + r * (cm * (T)0.5);
      const T sc = 2 * dot(w, r);
      cI.xx = Ir.xx + sc - 2 * w.x * r.x;
      cI.yy = Ir.yy + sc - 2 * w.y * r.y;
      cI.zz = Ir.zz + sc - 2 * w.z * r.z;
      cI.xy = Ir.xy - (w.x * r.y + r.x * w.y);
      cI.xz = Ir.xz - (w.x * r.z + r.x * w.z);
      cI.yz = Ir.yz - (w.y * r.z + r.y * w.z);
      ch = hr + r * cm;
    }
  }
  M3<T> R;
  MAKE_R(R, qx, qy, qz, qw);
  const V3<T> dw = mul(R, dft);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const V3<T> Mf = mul(R, Fp[k].f), Mn = mul(R, Fp[k].n);
    const V3<T> jw = mul(R, jc[k]);
    if (direct) {
      const unsigned x = jxN[k];
      STLX(a.M, 6, 0, x, Mf.x);
      STLX(a.M, midx18(1, 1) + 5, 0, x, Mf.y);
      STLX(a.M, midx18(2, 2) + 4, 0, x, Mf.z);
      STLX(a.M, midx18(3, 3) + 3, 0, x, Mn.x);
      STLX(a.M, midx18(4, 4) + 2, 0, x, Mn.y);
      STLX(a.M, midx18(5, 5) + 1, 0, x, Mn.z);
      STLX(a.Jc, 0 * 18 + 6, 54, x, jw.x);  // overwrites a zero written above (same lane, program order)
      STLX(a.Jc, 1 * 18 + 6, 54, x, jw.y);
      STLX(a.Jc, 2 * 18 + 6, 54, x, jw.z);
    } else {
      hl[(6 + 0 + k) * 64] = Mf.x; hl[(6 + 3 + k) * 64] = Mf.y; hl[(6 + 6 + k) * 64] = Mf.z;
      hl[(6 + 9 + k) * 64] = Mn.x; hl[(6 + 12 + k) * 64] = Mn.y; hl[(6 + 15 + k) * 64] = Mn.z;
      hl[(24 + 0 + k) * 64] = jw.x; hl[(24 + 3 + k) * 64] = jw.y; hl[(24 + 6 + k) * 64] = jw.z;
    }
  }
  if (direct) {
    STL(a.Jc, 0 * 18 + 4, 54, dw.z);  STL(a.Jc, 0 * 18 + 5, 54, -dw.y);
    STL(a.Jc, 1 * 18 + 3, 54, -dw.z); STL(a.Jc, 1 * 18 + 5, 54, dw.x);
    STL(a.Jc, 2 * 18 + 3, 54, dw.y);  STL(a.Jc, 2 * 18 + 4, 54, -dw.x);
    if (a.pf) {
      STL(a.pf, 0, 3, ldq(0) + dw.x);
      STL(a.pf, 1, 3, ldq(1) + dw.y);
      STL(a.pf, 2, 3, ldq(2) + dw.z);
    }
  } else {
    hl[33 * 64] = dw.x; hl[34 * 64] = dw.y; hl[35 * 64] = dw.z;
    // pf goes out HERE, in front of the hand-over: it adds the base position, which a rollout workgroup reads from its LDS state image (SIMG) -- and behind after_hand() the
    // integrator's phase 2 (another wavefront) overwrites that image with the NEXT state.  Stored with M and Jc behind the hand-over (rounds 5-6) a stalled store queue let
    // phase 2 win about once in a thousand rollouts: pf = new base position + old lever arm, 4e-4 off (tools/soak.py, seed 101 case 1607; profiles/r06zzz_soak_long.log)
    if (!a.skip_mats && a.pf) {
      STL(a.pf, 0, 3, ldq(0) + dw.x);
      STL(a.pf, 1, 3, ldq(1) + dw.y);
      STL(a.pf, 2, 3, ldq(2) + dw.z);
    }
  }
  {
    const T bm = model->base_m;
    const V3<T> bh = mk<T>(model->base_h[0], model->base_h[1], model->base_h[2]);
    const T tm = xrow_sum(cm) + bm;
    const V3<T> th = xrow_sum(ch) + bh;
    S3<T> tI;
    tI.xx = xrow_sum(cI.xx) + model->base_Io[0]; tI.xy = xrow_sum(cI.xy) + model->base_Io[1]; tI.xz = xrow_sum(cI.xz) + model->base_Io[2];
    tI.yy = xrow_sum(cI.yy) + model->base_Io[3]; tI.yz = xrow_sum(cI.yz) + model->base_Io[4]; tI.zz = xrow_sum(cI.zz) + model->base_Io[5];
    const V3<T> hw = mul(R, th);
    const S3<T> Iw = congr(R, tI);
    if (direct) {
      T* M = a.M;
      ST4(M, midx18(0, 0), tm, midx18(1, 1), tm, midx18(2, 2), tm, midx18(0, 4), hw.z);
      ST4(M, midx18(0, 5), -hw.y, midx18(1, 3), -hw.z, midx18(1, 5), hw.x, midx18(2, 3), hw.y);
      ST4(M, midx18(2, 4), -hw.x, midx18(3, 3), Iw.xx, midx18(3, 4), Iw.xy, midx18(3, 5), Iw.xz);
      if (leg < 3) STV(M, sel4<int>(leg, midx18(4, 4), midx18(4, 5), midx18(5, 5), 0), sel4<T>(leg, Iw.yy, Iw.yz, Iw.zz, Iw.zz));
    } else {
      hl[36 * 64] = tm; hl[37 * 64] = hw.x; hl[38 * 64] = hw.y; hl[39 * 64] = hw.z;
      hl[40 * 64] = Iw.xx; hl[41 * 64] = Iw.xy; hl[42 * 64] = Iw.xz; hl[43 * 64] = Iw.yy; hl[44 * 64] = Iw.yz; hl[45 * 64] = Iw.zz;
    }
  }
  if (!direct) {
    after_hand();   // the image is complete: the integrator may start
    if (!a.skip_mats) {   // M, Jc, pf for the caller, from the image (my own LDS words: program order of one lane)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int j = k; j < 3; ++j) {
          int i = 6 + jx[k], jj = 6 + jx[j];
          if (i > jj) { const int t = i; i = jj; jj = t; }
          STV(a.M, i * 18 - i * (i - 1) / 2 + (jj - i), hl[(k * 3 - k * (k - 1) / 2 + (j - k)) * 64]);
        }
        const unsigned x = jxN[k];
        STLX(a.M, 6, 0, x, hl[(6 + 0 + k) * 64]);
        STLX(a.M, midx18(1, 1) + 5, 0, x, hl[(6 + 3 + k) * 64]);
        STLX(a.M, midx18(2, 2) + 4, 0, x, hl[(6 + 6 + k) * 64]);
        STLX(a.M, midx18(3, 3) + 3, 0, x, hl[(6 + 9 + k) * 64]);
        STLX(a.M, midx18(4, 4) + 2, 0, x, hl[(6 + 12 + k) * 64]);
        STLX(a.M, midx18(5, 5) + 1, 0, x, hl[(6 + 15 + k) * 64]);
        STLX(a.Jc, 0 * 18 + 6, 54, x, hl[(24 + 0 + k) * 64]);
        STLX(a.Jc, 1 * 18 + 6, 54, x, hl[(24 + 3 + k) * 64]);
        STLX(a.Jc, 2 * 18 + 6, 54, x, hl[(24 + 6 + k) * 64]);
      }
      const T dx = hl[33 * 64], dy = hl[34 * 64], dz = hl[35 * 64];
      STL(a.Jc, 0 * 18 + 4, 54, dz);  STL(a.Jc, 0 * 18 + 5, 54, -dy);
      STL(a.Jc, 1 * 18 + 3, 54, -dz); STL(a.Jc, 1 * 18 + 5, 54, dx);
      STL(a.Jc, 2 * 18 + 3, 54, dy);  STL(a.Jc, 2 * 18 + 4, 54, -dx);
      const T tm = hl[36 * 64], hwx = hl[37 * 64], hwy = hl[38 * 64], hwz = hl[39 * 64];
      T* M = a.M;
      ST4(M, midx18(0, 0), tm, midx18(1, 1), tm, midx18(2, 2), tm, midx18(0, 4), hwz);
      ST4(M, midx18(0, 5), -hwy, midx18(1, 3), -hwz, midx18(1, 5), hwx, midx18(2, 3), hwy);
      ST4(M, midx18(2, 4), -hwx, midx18(3, 3), hl[40 * 64], midx18(3, 4), hl[41 * 64], midx18(3, 5), hl[42 * 64]);
      if (leg < 3) STV(M, sel4<int>(leg, midx18(4, 4), midx18(4, 5), midx18(5, 5), 0), sel4<T>(leg, hl[43 * 64], hl[44 * 64], hl[45 * 64], hl[45 * 64]));
    }
  }
  if constexpr (ZEROS == 2) { if (!a.skip_consts) write_consts(); }   // (own-leg columns were written above: the constants never overlap them)
}

template <class T, int BLOCK>
__global__ __launch_bounds__(BLOCK, WBC_MJ_WAVES) void mass_jac_kernel(const DevModel<T>* __restrict__ model, SweepArgs<T> a) {
  mass_jac_body<T, BLOCK, 0>(model, a, nullptr, nullptr);
}

// ======================================================================================================================
// rnea_step_kernel: bias forces h, tau_partial = (M vdot_des + h - rhat) by a second (acceleration-only) force
// recursion, foot geometry for the QP, momentum observer.
// ======================================================================================================================
// EXT: as for mass_jac_body; additionally the step workspace goes to the workgroup's LDS image wsl[word][16].
// `before_refs()` runs after the state loads are issued and before w_des / vdot_des are read (the persistent tracking
// rollout waits there for the planner role that writes them).
// `after_geom()` (wavefront roles, EXT != 0) is called twice, long before tau_partial is ready: when the lever arms WS_D are
// in the LDS image -- all the QP needs to assemble and factor H -- and again when the target wrench WS_B (w_des) is; the
// role counts both on its first flag.
struct NoWait { WBC_DEV void operator()() const {} };
// `after_taup()` (a real hook only in the four-wavefront rollout workgroups): tau_partial goes to the LDS image -- and the hook raises the QP's flag -- BEFORE the base
// rows of h are summed over the legs, rotated and written; only the integrator's phase 2, which starts behind the torque map, needs those (the caller raises a second
// flag behind the body).
template <class T, int MODE, int BLOCK, int EXT, int SPW = 16, class BeforeRefs = NoWait, class AfterGeom = NoWait, class AfterTaup = NoWait>
WBC_DEV void rnea_step_body(const DevModel<T>* __restrict__ model, const DevParams<T> prm, const SweepArgs<T>& a, const T* cst_ext,
                            T* wsl, BeforeRefs before_refs = BeforeRefs(), AfterGeom after_geom = AfterGeom(), T* hres = nullptr, AfterTaup after_taup = AfterTaup()) {
  // hres (persistent rollout): LDS image [..][16] whose rows 0 .. 17 ALSO receive h (the integrator reads it behind an LDS-only barrier)
  static_assert(!EXT || BLOCK == 64 || BLOCK == 128, "one wavefront (128: one of the two role wavefronts of a fused_pair_kernel workgroup, each with its own half of the parking lot)");
  WBC_LAUNDERED_TID(tx);
  constexpr bool WH = (MODE & RS_H) != 0, STEP = (MODE & RS_STEP) != 0, OBS = (MODE & RS_OBS) != 0, WPF = (MODE & RS_PF) != 0, FWD_B = (MODE & RS_NOB) == 0;
  constexpr bool OBSW = (MODE & RS_OBSW) != 0;
  static_assert(!OBSW || (EXT != 0 && OBS && !STEP && !WH), "the observer role exists only inside the fused tick");
  constexpr bool NOJC = (MODE & RS_NOJC) != 0;
  static_assert(!NOJC || (EXT != 0 && STEP && !OBS && !WPF), "RS_NOJC: a step role next to a mass_jac role");
  constexpr bool GEOM = (STEP && !NOJC) || OBS || WPF;     // foot position / own-leg Jacobian needed (the lever arms of a role go out EARLY, from their own chain)
  constexpr bool LANE2 = (MODE & RS_LANE2) != 0;   // the two chains side by side in the lanes (device_types.hpp)
  static_assert(!LANE2 || (EXT != 0 && SPW == 4 && STEP && WH && !OBS), "RS_LANE2: a role of 4-state workgroups that wants h and tau_partial");
  constexpr bool TWO = STEP && WH && !LANE2;    // h and tau_partial both wanted: two force chains in every lane; else one (merged, or one per lane group)
  constexpr bool BASEROWS = WH || OBS;          // base rows of h / p / beta needed
  constexpr bool EARLY = EXT != 0 && STEP;      // role inside the fused tick: publish the foot lever arms first
  __shared__ __attribute__((aligned(512))) T cst_own[EXT ? 1 : CST_WORDS];   // (aligned: first in LDS, see section 4.9 of docs/DESIGN_R04.md)
  if constexpr (EXT == 0) {
    for (int i = tx; i < CST_WORDS; i += blockDim.x) cst_own[i] = model->cst[i];
    __syncthreads();
  }
  const T* cst = EXT ? cst_ext : cst_own;
  WBC_ADDR_MACROS
  // RS_LANE2: slots 4 .. 7 are the acceleration-only recursion of states 0 .. 3 (they stay !live: they store nothing to memory)
  const bool chainB = LANE2 && (int)(tx & 15) >= SPW && (int)(tx & 15) < 2 * SPW;
  if constexpr (LANE2) {
    if (chainB) { const size_t sb = (size_t)blockIdx.x * SPW + (tx & 15) - SPW; s32 = (unsigned)(sb < N ? sb : N - 1); }
  }
#define WSTV(comp, val) do { if constexpr (EXT != 0) wsl[(comp) * 16 + (int)(tx & 15)] = (val); else STV(a.ws, comp, val); } while (0)
#define WST4(c0, v0_, c1, v1_, c2, v2_, c3, v3_) WSTV(sel4<int>(leg, c0, c1, c2, c3), sel4<T>(leg, v0_, v1_, v2_, v3_))
#define WSTL(c0, stride, val) WSTV((c0) + (stride) * leg, val)

  WBC_STATE_MACROS
  T qq[4], vb[6];
#pragma unroll
  for (int c = 0; c < 4; ++c) qq[c] = ldq(3 + c);
#pragma unroll
  for (int c = 0; c < 6; ++c) vb[c] = ldv(c);
  T qpos[3] = {0, 0, 0};   // base position for pf: requested with the rest (loaded where it is used it exposed a memory latency)
  if (WPF && a.pf) {
#pragma unroll
    for (int c = 0; c < 3; ++c) qpos[c] = ldq(c);
  }
  int jx[3];
  unsigned jxN[3];
  jidx_of_leg(model, a.jpack, leg, jx);
#pragma unroll
  for (int k = 0; k < 3; ++k) jxN[k] = (unsigned)jx[k] * N32;
  T ql[3], vl[3], al[3] = {0, 0, 0}, ad[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 3; ++k) { ql[k] = LDQJ(k); vl[k] = LDVJ(k); }
  // The references: requested with the state -- unless somebody has to be waited for first (before_refs is a real hook:
  // the planner role of the tracking rollout), in which case the lever arms, which need q only, go out before that wait.
  constexpr bool LATE_REFS = EARLY && !std::is_same<BeforeRefs, NoWait>::value;
  T bw[6] = {0, 0, 0, 0, 0, 0};
  constexpr bool RIMG = (MODE & RS_REFIMG) != 0;   // (planner in the loop, 4-state rollout workgroups) the references are in the planner role's LDS image
  static_assert(!RIMG || (SIMG && STEP), "RS_REFIMG: a step role of a 4-state rollout workgroup");
  auto load_refs = [&]() __attribute__((always_inline)) {
    before_refs();
    const T* const ri = RIMG ? a.refimg + (si_ - a.simg) : nullptr;
    if (STEP) {
#pragma unroll
      for (int k = 0; k < 3; ++k) al[k] = RIMG ? ri[(6 + 6 + jx[k]) * 16] : LDX(a.vdot_des, 6, jxN[k]);
#pragma unroll
      for (int c = 0; c < 6; ++c) ad[c] = RIMG ? ri[(6 + c) * 16] : LDU(a.vdot_des, c);
    }
    if (STEP && !OBS && FWD_B) {  // observer off: the QP target wrench is just w_des (RS_NOB: the QP kernel reads it itself)
#pragma unroll
      for (int c = 0; c < 6; ++c) bw[c] = RIMG ? ri[c * 16] : LDU(a.w_des, c);
    }
  };
  if constexpr (!LATE_REFS) load_refs();
  auto lane2_mask = [&]() __attribute__((always_inline)) {   // chain A keeps the velocities (and gravity, below), chain B the desired accelerations
    if constexpr (LANE2) {
#pragma unroll
      for (int c = 0; c < 6; ++c) { vb[c] = chainB ? (T)0 : vb[c]; ad[c] = chainB ? ad[c] : (T)0; }
#pragma unroll
      for (int k = 0; k < 3; ++k) { vl[k] = chainB ? (T)0 : vl[k]; al[k] = chainB ? al[k] : (T)0; }
    }
  };
  if constexpr (!LATE_REFS) lane2_mask();
  if constexpr (EXT == 2) __syncthreads();   // tables staged by the other wavefronts while my loads are in flight
  if constexpr (!EARLY) {   // (EARLY: w_des is stored after the lever arms, so that those do not wait for its load)
    if (STEP && !OBS && FWD_B) {
      WST4(WS_B + 0, bw[0], WS_B + 1, bw[1], WS_B + 2, bw[2], WS_B + 3, bw[3]);
      if (leg < 2) WSTV(WS_B + 4 + leg, leg == 0 ? bw[4] : bw[5]);
    }
  }
  T qx, qy, qz, qw;
  {
    const T n = rsqrt_sel<(SIMG || (0 && EXT == 2))>(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
    qx = qq[0] * n; qy = qq[1] * n; qz = qq[2] * n; qw = qq[3] * n;
  }
  T sn3[3] = {0, 0, 0}, cs3[3] = {0, 0, 0};
  // (four-wavefront rollout workgroups: the role has its SIMD's whole register file, so the three joint rotations of the early lever-arm chain are KEPT
  //  for the forward sweep instead of being rebuilt there from the same sin / cos -- 81 multiply-adds off the tick's critical chain; -DWBC_RNEA_KEEP_E=0: rebuilt)
  constexpr bool KEEP_E = EARLY && SIMG && SPW == 4;
  M3<T> Ekeep[KEEP_E ? 3 : 1];
  if constexpr (EARLY) {
    // The QP waves can assemble and factor H from the four lever arms alone, so those go out ahead of the force
    // recursions: d = r_0 + E_0 (r_1 + E_1 (r_2 + E_2 d_foot)), the same expression the return sweep evaluates (which
    // then does not store WS_D again).  The joint rotations are rebuilt in the forward sweep from the same sin / cos
    // (laundered here, so that the compiler does not keep 27 matrix entries alive until then).
#pragma unroll
    for (int k = 0; k < 3; ++k) sincos_t(ql[k], &sn3[k], &cs3[k]);
    V3<T> d = mk<T>(CS(129), CS(130), CS(131));
#pragma unroll
    for (int k = 2; k >= 0; --k) {
      const int o = JOINT_WORDS * k;
      T s_ = sn3[k], c_ = cs3[k];
      if constexpr (!KEEP_E) asm volatile("" : "+v"(s_), "+v"(c_));
      M3<T> E;
#pragma unroll
      for (int e = 0; e < 9; ++e) E.a[e] = CS(o + e) + c_ * CS(o + 9 + e) + s_ * CS(o + 18 + e);
      if constexpr (KEEP_E) Ekeep[k] = E;
      d = mk<T>(CS(o + 27), CS(o + 28), CS(o + 29)) + mul(E, d);
    }
    M3<T> R0;
    MAKE_R(R0, qx, qy, qz, qw);
    const V3<T> dw0 = mul(R0, d);
    WSTL(WS_D + 0, 3, dw0.x);
    WSTL(WS_D + 1, 3, dw0.y);
    WSTL(WS_D + 2, 3, dw0.z);
    after_geom();   // first call: lever arms are out
    if constexpr (LATE_REFS) { load_refs(); lane2_mask(); }
    if (STEP && !OBS && FWD_B) {
      WST4(WS_B + 0, bw[0], WS_B + 1, bw[1], WS_B + 2, bw[2], WS_B + 3, bw[3]);
      if (leg < 2) WSTV(WS_B + 4 + leg, leg == 0 ? bw[4] : bw[5]);
    }
    after_geom();   // second call: w_des is out
  }
  const T bm = model->base_m;
  const V3<T> bh = mk<T>(model->base_h[0], model->base_h[1], model->base_h[2]);
  S3<T> bI;
  bI.xx = model->base_Io[0]; bI.xy = model->base_Io[1]; bI.xz = model->base_Io[2];
  bI.yy = model->base_Io[3]; bI.yz = model->base_Io[4]; bI.zz = model->base_Io[5];

  // parked per joint (k = 0, 1): E 9, chain-b force 6, [chain-a force 6], [OBS: momentum 6, weight 6, velocity 6]
  constexpr int PW = 15 + (TWO ? 6 : 0) + (OBS ? 18 : 0);
  constexpr int PB = BASEROWS ? (OBS ? 18 : 6) : 1;
  __shared__ T park[2 * PW + PB][BLOCK];
  const int ln = EXT ? (int)(tx & (unsigned)(BLOCK - 1)) : (int)tx;
  constexpr int OFF_A = 15, OFF_O = 15 + (TWO ? 6 : 0);

  V3<T> omp, vp, aAp, aLp, gLp, a2Ap, a2Lp;
  {
    M3<T> R;
    MAKE_R(R, qx, qy, qz, qw);
    const V3<T> om0 = tmul(R, mk<T>(vb[3], vb[4], vb[5]));
    const V3<T> v0 = tmul(R, mk<T>(vb[0], vb[1], vb[2]));
    V3<T> gneg = tmul(R, mk<T>(-model->grav[0], -model->grav[1], -model->grav[2]));
    if constexpr (LANE2) { gneg.x = chainB ? (T)0 : gneg.x; gneg.y = chainB ? (T)0 : gneg.y; gneg.z = chainB ? (T)0 : gneg.z; }
    const V3<T> aL0 = gneg - cross(om0, v0);
    if (BASEROWS) {
      const SF<T> Iv0 = inertia_mul(bm, bh, bI, om0, v0);
      const SF<T> Ia0 = inertia_mul(bm, bh, bI, mk<T>(0, 0, 0), aL0);
      T* pb = &park[2 * PW][ln];
      const V3<T> bwn = Ia0.n + cross(om0, Iv0.n) + cross(v0, Iv0.f), bwf = Ia0.f + cross(om0, Iv0.f);
      pb[0] = bwn.x; pb[BLOCK] = bwn.y; pb[BLOCK * 2] = bwn.z; pb[BLOCK * 3] = bwf.x; pb[BLOCK * 4] = bwf.y; pb[BLOCK * 5] = bwf.z;
      if (OBS) {
        const V3<T> gn = cross(bh, gneg), gf = gneg * bm;
        pb[BLOCK * 6] = Iv0.n.x; pb[BLOCK * 7] = Iv0.n.y; pb[BLOCK * 8] = Iv0.n.z; pb[BLOCK * 9] = Iv0.f.x; pb[BLOCK * 10] = Iv0.f.y; pb[BLOCK * 11] = Iv0.f.z;
        pb[BLOCK * 12] = gn.x; pb[BLOCK * 13] = gn.y; pb[BLOCK * 14] = gn.z; pb[BLOCK * 15] = gf.x; pb[BLOCK * 16] = gf.y; pb[BLOCK * 17] = gf.z;
      }
    }
    omp = om0; vp = v0; aAp = mk<T>(0, 0, 0); aLp = aL0; gLp = gneg;
    a2Ap = tmul(R, mk<T>(ad[3], ad[4], ad[5]));   // acceleration-only chain: base [R^T wdot ; R^T pddot]
    a2Lp = tmul(R, mk<T>(ad[0], ad[1], ad[2]));
    if (STEP && !TWO) { aAp = aAp + a2Ap; aLp = aLp + a2Lp; }  // one merged chain: RNEA(q, v, vdot_des)
  }

  // ------------------------------------------------------------------ forward sweep
  M3<T> E2;
  SF<T> f2, fa2, m2, g2;
  V3<T> om2, vv2;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int o = JOINT_WORDS * k;
    T sn, cs;
    if constexpr (EARLY) { sn = sn3[k]; cs = cs3[k]; }
    else sincos_t(ql[k], &sn, &cs);
    M3<T> E;
    if constexpr (KEEP_E) E = Ekeep[k];
    else {
#pragma unroll
      for (int e = 0; e < 9; ++e) E.a[e] = CS(o + e) + cs * CS(o + 9 + e) + sn * CS(o + 18 + e);
    }
    const V3<T> r = mk<T>(CS(o + 27), CS(o + 28), CS(o + 29));
    const V3<T> ax = mk<T>(CS(o + 30), CS(o + 31), CS(o + 32));
    const T m = CS(o + 33);
    const V3<T> h = mk<T>(CS(o + 34), CS(o + 35), CS(o + 36));
    S3<T> Io;
    Io.xx = CS(o + 37); Io.xy = CS(o + 38); Io.xz = CS(o + 39); Io.yy = CS(o + 40); Io.yz = CS(o + 41); Io.zz = CS(o + 42);
    const T qd = vl[k];
    const V3<T> om = tmul(E, omp) + ax * qd;
    const V3<T> vv = tmul(E, vp + cross(omp, r));
    V3<T> aA = tmul(E, aAp) + cross(om, ax) * qd;
    const V3<T> aL = tmul(E, aLp + cross(aAp, r)) + cross(vv, ax) * qd;
    if (STEP && !TWO) aA = aA + ax * al[k];
    const SF<T> Iv = inertia_mul(m, h, Io, om, vv);
    const SF<T> Ia = inertia_mul(m, h, Io, aA, aL);
    SF<T> fk, fak, gk;
    fk.n = Ia.n + cross(om, Iv.n) + cross(vv, Iv.f);
    fk.f = Ia.f + cross(om, Iv.f);
    if (TWO) {
      const V3<T> a2A = tmul(E, a2Ap) + ax * al[k];
      const V3<T> a2L = tmul(E, a2Lp + cross(a2Ap, r));
      fak = inertia_mul(m, h, Io, a2A, a2L);
      a2Ap = a2A; a2Lp = a2L;
    }
    if (OBS) {
      const V3<T> gL = tmul(E, gLp);
      gk.n = cross(h, gL);
      gk.f = gL * m;
      gLp = gL;
    }
    if (k < 2) {
      T* pk = &park[PW * k][ln];
#pragma unroll
      for (int e = 0; e < 9; ++e) pk[BLOCK * e] = E.a[e];
      pk[BLOCK * 9] = fk.n.x; pk[BLOCK * 10] = fk.n.y; pk[BLOCK * 11] = fk.n.z;
      pk[BLOCK * 12] = fk.f.x; pk[BLOCK * 13] = fk.f.y; pk[BLOCK * 14] = fk.f.z;
      if (TWO) {
        pk[BLOCK * (OFF_A + 0)] = fak.n.x; pk[BLOCK * (OFF_A + 1)] = fak.n.y; pk[BLOCK * (OFF_A + 2)] = fak.n.z;
        pk[BLOCK * (OFF_A + 3)] = fak.f.x; pk[BLOCK * (OFF_A + 4)] = fak.f.y; pk[BLOCK * (OFF_A + 5)] = fak.f.z;
      }
      if (OBS) {
        pk[BLOCK * (OFF_O + 0)] = Iv.n.x; pk[BLOCK * (OFF_O + 1)] = Iv.n.y; pk[BLOCK * (OFF_O + 2)] = Iv.n.z;
        pk[BLOCK * (OFF_O + 3)] = Iv.f.x; pk[BLOCK * (OFF_O + 4)] = Iv.f.y; pk[BLOCK * (OFF_O + 5)] = Iv.f.z;
        pk[BLOCK * (OFF_O + 6)] = gk.n.x; pk[BLOCK * (OFF_O + 7)] = gk.n.y; pk[BLOCK * (OFF_O + 8)] = gk.n.z;
        pk[BLOCK * (OFF_O + 9)] = gk.f.x; pk[BLOCK * (OFF_O + 10)] = gk.f.y; pk[BLOCK * (OFF_O + 11)] = gk.f.z;
        pk[BLOCK * (OFF_O + 12)] = om.x; pk[BLOCK * (OFF_O + 13)] = om.y; pk[BLOCK * (OFF_O + 14)] = om.z;
        pk[BLOCK * (OFF_O + 15)] = vv.x; pk[BLOCK * (OFF_O + 16)] = vv.y; pk[BLOCK * (OFF_O + 17)] = vv.z;
      }
    } else {
      E2 = E; f2 = fk;
      if (TWO) fa2 = fak;
      if (OBS) { m2 = Iv; g2 = gk; om2 = om; vv2 = vv; }
    }
    omp = om; vp = vv; aAp = aA; aLp = aL;
  }

  // ------------------------------------------------------------------ return sweep
  T p_leg[3], ct_leg[3], g_leg[3], taup[3] = {0, 0, 0};
  // (persistent rollout: h goes to the integrator through the LDS image `hres`; the caller's buffer gets the LAST tick's -- SweepArgs::skip_mats -- and the three
  //  guarded stores inside the return sweep were ~0.15 us of the rnea role, the tick's critical chain)
  const bool h_mem = !(hres != nullptr && a.skip_mats != 0);
  V3<T> dft = mk<T>(CS(129), CS(130), CS(131));
  V3<T> jc[3];
  SF<T> facc, aacc, macc, gacc;
#pragma unroll
  for (int k = 2; k >= 0; --k) {
    const int o = JOINT_WORDS * k;
    const V3<T> r = mk<T>(CS(o + 27), CS(o + 28), CS(o + 29));
    const V3<T> ax = mk<T>(CS(o + 30), CS(o + 31), CS(o + 32));
    M3<T> E;
    SF<T> fk, fak, mk_, gk;
    V3<T> om, vv;
    if (k == 2) {
      E = E2; fk = f2;
      if (TWO) fak = fa2;
      if (OBS) { mk_ = m2; gk = g2; om = om2; vv = vv2; }
    } else {
      const T* pk = &park[PW * k][ln];
#pragma unroll
      for (int e = 0; e < 9; ++e) E.a[e] = pk[BLOCK * e];
      fk.n = mk<T>(pk[BLOCK * 9], pk[BLOCK * 10], pk[BLOCK * 11]) + facc.n;
      fk.f = mk<T>(pk[BLOCK * 12], pk[BLOCK * 13], pk[BLOCK * 14]) + facc.f;
      if (TWO) {
        fak.n = mk<T>(pk[BLOCK * (OFF_A + 0)], pk[BLOCK * (OFF_A + 1)], pk[BLOCK * (OFF_A + 2)]) + aacc.n;
        fak.f = mk<T>(pk[BLOCK * (OFF_A + 3)], pk[BLOCK * (OFF_A + 4)], pk[BLOCK * (OFF_A + 5)]) + aacc.f;
      }
      if (OBS) {
        mk_.n = mk<T>(pk[BLOCK * (OFF_O + 0)], pk[BLOCK * (OFF_O + 1)], pk[BLOCK * (OFF_O + 2)]) + macc.n;
        mk_.f = mk<T>(pk[BLOCK * (OFF_O + 3)], pk[BLOCK * (OFF_O + 4)], pk[BLOCK * (OFF_O + 5)]) + macc.f;
        gk.n = mk<T>(pk[BLOCK * (OFF_O + 6)], pk[BLOCK * (OFF_O + 7)], pk[BLOCK * (OFF_O + 8)]) + gacc.n;
        gk.f = mk<T>(pk[BLOCK * (OFF_O + 9)], pk[BLOCK * (OFF_O + 10)], pk[BLOCK * (OFF_O + 11)]) + gacc.f;
        om = mk<T>(pk[BLOCK * (OFF_O + 12)], pk[BLOCK * (OFF_O + 13)], pk[BLOCK * (OFF_O + 14)]);
        vv = mk<T>(pk[BLOCK * (OFF_O + 15)], pk[BLOCK * (OFF_O + 16)], pk[BLOCK * (OFF_O + 17)]);
      }
    }
    {
      const T hk = dot(ax, fk.n);  // bias torque (TWO) or merged M vdot_des + h (one chain)
      if (WH) {
        unsigned long long jp = a.jpack;
        asm volatile("" : "+s"(jp));
        const unsigned jx_k = ((unsigned)(jp >> (12 * leg)) >> (4 * k)) & 15u;
        const unsigned jxN_k = jx_k * N32;
        if (h_mem) STLX(a.h, 6, 0, jxN_k, hk);
        if (hres) hres[(6 + (int)jx_k) * 16 + (int)(tx & 15)] = hk;
      }
      if (STEP) taup[k] = hk + (TWO ? dot(ax, fak.n) : (T)0);
    }
    if (OBS) {
      p_leg[k] = dot(ax, mk_.n);
      ct_leg[k] = -dot(ax, cross(om, mk_.n) + cross(vv, mk_.f));
      g_leg[k] = dot(ax, gk.n);
    }
    if (GEOM) {
      jc[k] = cross(ax, dft);
      dft = r + mul(E, dft);
#pragma unroll
      for (int j = k; j < 3; ++j) jc[j] = mul(E, jc[j]);
    }
    if (k > 0 || BASEROWS) facc = to_parent(E, r, fk);
    if (TWO && k > 0) aacc = to_parent(E, r, fak);
    if (OBS) { macc = to_parent(E, r, mk_); gacc = to_parent(E, r, gk); }
  }

  M3<T> R;
  MAKE_R(R, qx, qy, qz, qw);
  V3<T> dw = mk<T>(0, 0, 0), jw[3];
  if (GEOM) {
    dw = mul(R, dft);
#pragma unroll
    for (int k = 0; k < 3; ++k) jw[k] = mul(R, jc[k]);
  }
  if (WPF && a.pf) {
    STL(a.pf, 0, 3, qpos[0] + dw.x);
    STL(a.pf, 1, 3, qpos[1] + dw.y);
    STL(a.pf, 2, 3, qpos[2] + dw.z);
  }
  if (STEP) {
    if constexpr (!EARLY) {
      WSTL(WS_D + 0, 3, dw.x);
      WSTL(WS_D + 1, 3, dw.y);
      WSTL(WS_D + 2, 3, dw.z);
    }
    if constexpr (GEOM) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        WSTL(WS_JCL + 0 + k, 9, jw[k].x);
        WSTL(WS_JCL + 3 + k, 9, jw[k].y);
        WSTL(WS_JCL + 6 + k, 9, jw[k].z);
      }
    }
  }
  constexpr bool TAUP_FIRST = STEP && !OBS && !OBSW && WH && !std::is_same<AfterTaup, NoWait>::value;
  if constexpr (TAUP_FIRST) {
    if constexpr (LANE2) {
#pragma unroll
      for (int k = 0; k < 3; ++k) taup[k] += dpp_mov<0x12C>(taup[k]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) WSTL(WS_TAUP + k, 3, taup[k]);
    after_taup();
  }
  if (WH) {
    const T* pb = &park[2 * PW][ln];
    const V3<T> bfn = xrow_sum(facc.n) + mk<T>(pb[0], pb[BLOCK], pb[BLOCK * 2]);
    const V3<T> bff = xrow_sum(facc.f) + mk<T>(pb[BLOCK * 3], pb[BLOCK * 4], pb[BLOCK * 5]);
    const V3<T> hb_f = mul(R, bff), hb_n = mul(R, bfn);
    if (h_mem) {
      ST4(a.h, 0, hb_f.x, 1, hb_f.y, 2, hb_f.z, 3, hb_n.x);
      if (leg < 2) STV(a.h, 4 + leg, leg == 0 ? hb_n.y : hb_n.z);
    }
    if (hres) {
      hres[sel4<int>(leg, 0, 1, 2, 3) * 16 + (int)(tx & 15)] = sel4<T>(leg, hb_f.x, hb_f.y, hb_f.z, hb_n.x);
      if (leg < 2) hres[(4 + leg) * 16 + (int)(tx & 15)] = leg == 0 ? hb_n.y : hb_n.z;
    }
  }
  T p_b[6], beta_b[6], beta_l[3];
  if (OBS) {
    const T* pb = &park[2 * PW][ln];
    SF<T> mom0, grv0;
    mom0.n = xrow_sum(macc.n) + mk<T>(pb[BLOCK * 6], pb[BLOCK * 7], pb[BLOCK * 8]);
    mom0.f = xrow_sum(macc.f) + mk<T>(pb[BLOCK * 9], pb[BLOCK * 10], pb[BLOCK * 11]);
    grv0.n = xrow_sum(gacc.n) + mk<T>(pb[BLOCK * 12], pb[BLOCK * 13], pb[BLOCK * 14]);
    grv0.f = xrow_sum(gacc.f) + mk<T>(pb[BLOCK * 15], pb[BLOCK * 16], pb[BLOCK * 17]);
    const V3<T> Pl = mul(R, mom0.f), Pa = mul(R, mom0.n);
    const V3<T> gl = mul(R, grv0.f), ga = mul(R, grv0.n);
    const V3<T> cx = cross(mk<T>(vb[0], vb[1], vb[2]), Pl);
    p_b[0] = Pl.x; p_b[1] = Pl.y; p_b[2] = Pl.z; p_b[3] = Pa.x; p_b[4] = Pa.y; p_b[5] = Pa.z;
    beta_b[0] = -gl.x; beta_b[1] = -gl.y; beta_b[2] = -gl.z;
    beta_b[3] = -cx.x - ga.x; beta_b[4] = -cx.y - ga.y; beta_b[5] = -cx.z - ga.z;
#pragma unroll
    for (int k = 0; k < 3; ++k) beta_l[k] = ct_leg[k] - g_leg[k];
    if (a.p) {
      ST4(a.p, 0, p_b[0], 1, p_b[1], 2, p_b[2], 3, p_b[3]);
      if (leg < 2) STV(a.p, 4 + leg, leg == 0 ? p_b[4] : p_b[5]);
#pragma unroll
      for (int k = 0; k < 3; ++k) STLX(a.p, 6, 0, jxN[k], p_leg[k]);
    }
    if (a.beta) {
      ST4(a.beta, 0, beta_b[0], 1, beta_b[1], 2, beta_b[2], 3, beta_b[3]);
      if (leg < 2) STV(a.beta, 4 + leg, leg == 0 ? beta_b[4] : beta_b[5]);
#pragma unroll
      for (int k = 0; k < 3; ++k) STLX(a.beta, 6, 0, jxN[k], beta_l[k]);
    }
  }
  if (STEP || OBSW) {
    T rb[6] = {0, 0, 0, 0, 0, 0}, rl[3] = {0, 0, 0};
    if (OBS && prm.observer_order > 0) {
      const V3<T> fp = mk<T>(LDV(a.f_prev, 3 * leg + 0), LDV(a.f_prev, 3 * leg + 1), LDV(a.f_prev, 3 * leg + 2));
      const V3<T> ub_f = xrow_sum(fp);
      const V3<T> ub_n = xrow_sum(cross(dw, fp));
      const T ub[6] = {ub_f.x, ub_f.y, ub_f.z, ub_n.x, ub_n.y, ub_n.z};
      const T dt = prm.dt;
      const bool o1 = prm.observer_order == 1;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const T r0 = LDU(a.obs_r, c);
        const T ig = LDU(a.obs_integ, c) + dt * (ub[c] + beta_b[c] + r0);
        const T e = p_b[c] - ig;
        rb[c] = o1 ? prm.K1[c] * e : r0 + dt * prm.K2[c] * (prm.K1[c] * e - r0);
        p_b[c] = ig;
      }
      ST4G(a.obs_integ, 0, p_b[0], 1, p_b[1], 2, p_b[2], 3, p_b[3]);
      if (leg < 2) STVG(a.obs_integ, 4 + leg, leg == 0 ? p_b[4] : p_b[5]);
      ST4G(a.obs_r, 0, rb[0], 1, rb[1], 2, rb[2], 3, rb[3]);
      if (leg < 2) STVG(a.obs_r, 4 + leg, leg == 0 ? rb[4] : rb[5]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = 6 + jx[k];
        const T r0 = LDV(a.obs_r, c);
        const T u = LDV(a.tau_prev, jx[k]) + dot(jw[k], fp);
        const T ig = LDV(a.obs_integ, c) + dt * (u + beta_l[k] + r0);
        const T e = p_leg[k] - ig;
        // gains of joint row c = 6 + jx[k] by a select over the 12 joint rows: a run-time index into the kernel-argument
        // struct may become a private (scratch) copy of the whole struct
        T k1 = prm.K1[6], k2 = prm.K2[6];
#pragma unroll
        for (int j = 1; j < 12; ++j) { k1 = (jx[k] == j) ? prm.K1[6 + j] : k1; k2 = (jx[k] == j) ? prm.K2[6 + j] : k2; }
        rl[k] = o1 ? k1 * e : r0 + dt * k2 * (k1 * e - r0);
        STVG(a.obs_integ, c, ig);
        STVG(a.obs_r, c, rl[k]);
      }
    }
    if constexpr (OBSW) {   // hand rhat to the QP waves, which subtract it from b and tau_partial themselves
      WST4(WS_RHAT + 0, rb[0], WS_RHAT + 1, rb[1], WS_RHAT + 2, rb[2], WS_RHAT + 3, rb[3]);
      if (leg < 2) WSTV(WS_RHAT + 4 + leg, leg == 0 ? rb[4] : rb[5]);
#pragma unroll
      for (int k = 0; k < 3; ++k) WSTL(WS_RHAT + 6 + k, 3, rl[k]);
    } else {
      if (OBS) {
        T b[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) b[c] = LDU(a.w_des, c) - rb[c];
        WST4(WS_B + 0, b[0], WS_B + 1, b[1], WS_B + 2, b[2], WS_B + 3, b[3]);
        if (leg < 2) WSTV(WS_B + 4 + leg, leg == 0 ? b[4] : b[5]);
      }
      if constexpr (!TAUP_FIRST) {
      if constexpr (LANE2) {   // tau_partial = h (my chain, slot s) + M vdot_des (slot s + 4): row_ror:12 hands lane i the value of lane (i + 4) mod 16
#pragma unroll
        for (int k = 0; k < 3; ++k) taup[k] += dpp_mov<0x12C>(taup[k]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) WSTL(WS_TAUP + k, 3, taup[k] - rl[k]);
      }
    }
  }
#undef WSTL
#undef WST4
#undef WSTV
}

template <class T, int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK, WBC_RS_WAVES) void rnea_step_kernel(const DevModel<T>* __restrict__ model, DevParams<T> prm,
                                                                         SweepArgs<T> a) {
  if (a.qp_todo && blockIdx.x == 0 && threadIdx.x == 0) a.qp_todo[0] = 0;
  rnea_step_body<T, MODE, BLOCK, 0>(model, prm, a, nullptr, nullptr);
}

#undef MAKE_R
#undef ST4
#undef ST4G
#undef STVG
#undef STLX
#undef STL
#undef STV
#undef LDX
#undef LDV
#undef LDU
#undef CS

}  // namespace wbc

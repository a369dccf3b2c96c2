// GRF QP + torque map kernel for gfx950, register/DPP-resident: SURVEY.md 8(a) units a7, a8, a9.
//
// Mapping: ONE QP PER 16-LANE DPP ROW, four QPs per wavefront.  A CDNA DPP "row" is 16 lanes, and
// row_newbcast / row_mirror / row_half_mirror / quad_perm move data inside a row at VALU speed without
// touching LDS -- so the 12x12 factors of the Goldfarb-Idnani method live entirely in VGPRs:
//   lane 4f+a (a<3) of a row owns variable (foot f, axis a); lane 4f+3 is a spare that carries constraints
//   Jc[12] = column `me` of J,  Jr[12] = row `me` of J,  Rr[12] = row `me` of R   (static register indices)
// Products with J^T use the column copy, products with J use the row copy, the rank-one Householder update
// of an added constraint touches both; every cross-lane operand is a row broadcast (2 DPP movs per double).
// All four feet are always variables (swing feet decouple: H_ii = alpha, g_i = 0 => f = 0), so there is no
// per-QP dimension and no dynamic register index; per-QP control flow is predication, wave-level control
// flow is a ballot.  Dropping a constraint (rare) restores J from a copy of J0 = L^-T kept in LDS and
// re-adds the remaining active constraints -- no Givens chain.  LDS is used only for that copy and for the
// one transposition that derives the row copy of J0 from the column copy.
//
// Same algorithm and tolerances as the CPU oracle (dual active set; "no primal step" and "new diagonal"
// share |d2|); iterates differ from the oracle's only by orthogonal transformations of the free columns.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_types.hpp"

namespace wbc {

template <class T> struct Lim;
template <> struct Lim<double> { static constexpr double eps = 2.220446049250313e-16; static constexpr double inf = __builtin_huge_val(); };
template <> struct Lim<float> { static constexpr float eps = 1.1920929e-07f; static constexpr float inf = __builtin_huge_valf(); };
WBC_DEV double fabs_t(double x) { return fabs(x); }
WBC_DEV float fabs_t(float x) { return fabsf(x); }

template <int I, int N, class F> WBC_DEV void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
template <int I, int N, class F> WBC_DEV void sfor_down(F&& f) {  // I = N-1 ... 0
  if constexpr (N > 0) { f(std::integral_constant<int, N - 1>{}); sfor_down<I, N - 1>(f); }
}

template <int CTRL> WBC_DEV float dppx(float x) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL> WBC_DEV double dppx(double x) {
  // 64-bit form: row_newbcast becomes ONE v_mov_b64_dpp (the only DPP64 control on gfx90a+); the other
  // controls are split by the compiler into two 32-bit DPP movs
  const long long xi = __double_as_longlong(x);
  return __longlong_as_double(__builtin_amdgcn_update_dpp(xi, xi, CTRL, 0xF, 0xF, true));
}
template <int CTRL> WBC_DEV int dppx(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true); }

constexpr int LJ(int j) { return j + j / 3; }  // lane (within the row) of variable j
// broadcast the value held by variable j's lane to the whole row
template <int J, class T> WBC_DEV T gbc(T x) { return dppx<0x150 + LJ(J)>(x); }
// all-reduce over the 16 lanes of a row: xor-1, xor-2 inside quads, then half-row and row mirrors
template <class T> WBC_DEV T gsum(T x) {
  x += dppx<0xB1>(x); x += dppx<0x4E>(x); x += dppx<0x141>(x); x += dppx<0x140>(x);
  return x;
}
// Row-wide argmin without a compare/select chain: the 5-bit id is written into the low mantissa bits of the
// value, which makes all keys distinct, so four DPP steps of v_min are an all-reduce that every lane agrees on.
// The value moves by < 2^-47 relative (f64) / 2^-18 (f32) -- selection only; callers that need the exact value
// of the winner fetch it from the winning lane.  BIG (finite) stands for "no candidate": inf | id would be a NaN.
template <class T> struct KeyT;
template <> struct KeyT<double> {
  static constexpr double BIG = 1e300;
  static WBC_DEV double pack(double v, int id) { return __longlong_as_double((__double_as_longlong(v) & ~63ll) | (long long)id); }
  static WBC_DEV int id(double k) { return (int)(__double_as_longlong(k) & 63ll); }
  static WBC_DEV double val(double k) { return __longlong_as_double(__double_as_longlong(k) & ~63ll); }
  // (v_min_f64 itself: fmin() makes the compiler canonicalise both operands first -- two v_max_f64 per step of the reduction,
  // eight dependent instructions per row argmin -- because it cannot know that the DPP-moved bit patterns are ordinary numbers)
  static WBC_DEV double mn(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
};
template <> struct KeyT<float> {
  static constexpr float BIG = 1e30f;
  static WBC_DEV float pack(float v, int id) { return __int_as_float((__float_as_int(v) & ~63) | id); }
  static WBC_DEV int id(float k) { return __float_as_int(k) & 63; }
  static WBC_DEV float val(float k) { return __int_as_float(__float_as_int(k) & ~63); }
  static WBC_DEV float mn(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
};
// in: v (BIG = none), id in [0,64).  out: row-uniform winner id in `id`, its value (low 6 mantissa bits cleared:
// relative error < 2^-46 in f64, 2^-17 in f32) in `v`; returns whether any lane had a candidate
template <class T> WBC_DEV bool gargmin(T& v, int& id) {
  using K = KeyT<T>;
  T k = K::pack(v, id);
  k = K::mn(k, dppx<0xB1>(k)); k = K::mn(k, dppx<0x4E>(k)); k = K::mn(k, dppx<0x141>(k)); k = K::mn(k, dppx<0x140>(k));
  id = K::id(k);
  v = K::val(k);
  return k < (T)(K::BIG * (T)0.5);
}
// read from a run-time lane of my own row
template <class T> WBC_DEV T gread(T x, int lane16, int rowbase) { return __shfl(x, rowbase | lane16); }

// Newton steps after v_rsq_f64 / v_rcp_f64 (2^-23 relative): 2 -> full double
WBC_DEV double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = fma(-x * y, y, 1.0); y = fma(0.5 * y, e, y);
  e = fma(-x * y, y, 1.0); y = fma(0.5 * y, e, y);
  return y;
}
WBC_DEV float rsqrt_nr(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  float e = fmaf(-x * y, y, 1.0f); y = fmaf(0.5f * y, e, y);
  return y;
}
WBC_DEV double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0); y = fma(y, e, y);
  e = fma(-x, y, 1.0); y = fma(y, e, y);
  return y;
}
// one Newton step: 2^-46 relative (where the result only scales a step or a projector row)
WBC_DEV double rcp_nr1(double x) {
  double y = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, y, 1.0);
  return fma(y, e, y);
}
WBC_DEV float rcp_nr(float x) {
  float y = __builtin_amdgcn_rcpf(x);
  float e = fmaf(-x, y, 1.0f); y = fmaf(y, e, y);
  return y;
}

// per wave: four 12x12 images of J0 ([i*12 + c]) and four images of R ([position*12 + variable]: row `me` of R is
// addressed by a run-time position, which LDS allows and a register array does not)
// ... and the 32 constraint rows (3 coefficients each) so that a candidate's normal is three broadcast reads
template <class T> struct G16Lds { T J0[4][144]; T R[4][12 * 12]; T C[4][32 * 3]; };

constexpr int WBC_QP_WAVES = 2;
// WBC_QP_RINV = 1: the active-set factor is kept as U = R^-1 (row `me` of U per variable lane, in the LDS image that held R).
// The dual step r = R^-1 d1 is then a matrix-vector product -- twelve broadcasts and FMAs in three independent chains --
// instead of a back-substitution whose iq steps each wait for the one before (round 2 stamps: 610 of the ~3 200 cycles of an
// iteration).  Appending a constraint needs nothing new: with r = R^-1 d1 at hand the new column of U is
// [-r / delta ; 1 / delta], delta = the new diagonal entry of R.
// One-wave workgroups (stand-alone kernel): every wavefront is its own workgroup, so its LDS and wave slot are released the
// moment ITS four QPs are done and the CU backfills.  Workgroups that share a 128-byte line of the inputs are mapped to the
// same XCD (L2).
// WSLDS (fused_tick.hip.hpp): the step workspace of the workgroup's 16 states is read from LDS (wsl[word][16], written
// by the sweep phase of the same workgroup) instead of from HBM.
// RHAT (with WSLDS, observer-on fused tick): the observer role left rhat in the LDS image; b and tau_partial are
// completed here (b -= rhat_base, tau_partial -= rhat_joint).
// QpSync (WSLDS only): the producer roles of the fused tick publish the workspace in three steps, each behind an LDS
// counter -- lever arms, then w_des, early; rhat when the observer role is done, tau_partial and the own-leg Jacobian
// blocks when the force recursions are -- and the QP waits for each only where it first needs it: H and its factor
// come from the lever arms alone, the target wrench enters with g, tau_partial only in the torque map.
struct QpSync {
  int* geom; int* rhat; int* fin;
  int need_geom, need_b, need_rhat, need_fin;   // `geom` counts twice per tick: lever arms out, then w_des out
#ifdef WBC_FUSED_STAMP   // diagnostic build (tools/fused_stamp.py): 100 MHz timestamps of the roles, one column per workgroup
  double* stamp; unsigned stampN;
#endif
  // (round 5) the speculative start reads the observer state r_prev that the observer role of the SAME launch rewrites in place: every QP wavefront
  // counts here once its read has returned, and the observer role stores r only behind that count (fused_tick.hip.hpp) -- the read is ordered in
  // front of the write, so that iters and the last bits of tau do not depend on timing
  int* rp_ack = nullptr;
  // (round 5, persistent rollout) LDS image [42][16] of the solver's scalar type that ALSO receives this tick's results: tau (rows 0 .. 11, caller's joint
  // order), f (12 .. 23), and -- from the rnea role -- h (24 .. 41).  The integrator takes them from there behind an LDS-only barrier, instead of waiting
  // for the global stores to be acknowledged and reading them back through L2
  void* res = nullptr;
  // (round 5, persistent rollout with the image) true: tau, f, status, iters of THIS tick go to the image only -- nobody reads them from memory before the
  // launch's last tick (the next tick's observer takes tau_prev, f_prev from the image, the integrator tau and f)
  bool skip_out = false;
  // (four-wavefront rollout workgroups) the mass_jac role's LDS image [46][64] (dyn_split.hip.hpp, MJ_HAND_WORDS): the torque map takes the own-leg Jacobian
  // blocks from its words 24 .. 32 (row m of foot f, joint k at (24 + 3 m + k) * 64 + 16 f + slot) once `hand_flag` has reached `need_hand`, and the rnea
  // role does not compute them (RS_NOJC)
  const void* hand = nullptr; int* hand_flag = nullptr; int need_hand = 0;
};
#ifdef WBC_FUSED_STAMP
// (one column per workgroup: column = first state of the workgroup, i.e. blockIdx.x * states-per-workgroup)
#define WBC_FSTAMP_S(ptr, N_, slot, spw) do { if ((threadIdx.x & 63) == 0) (ptr)[(size_t)(slot) * (N_) + (size_t)blockIdx.x * (spw)] = (double)wall_clock64(); } while (0)
#define WBC_FSTAMP(ptr, N_, slot) WBC_FSTAMP_S(ptr, N_, slot, 16)
// -DWBC_RO_STAMP_ALT (tools/rollout_stamp.py): slots 1, 2, 3 belong to the mass_jac / integrator / rnea roles of the rollout kernel instead of to QP wavefront 0
#ifdef WBC_RO_STAMP_ALT
#define WBC_QSTAMP_OK(slot) ((slot) != 1 && (slot) != 2 && (slot) != 3)
#else
#define WBC_QSTAMP_OK(slot) true
#endif
#define WBC_QSTAMP(slot) do { if (WBC_QSTAMP_OK(slot) && sync && sync->stamp && (tx >> 6) == 0) WBC_FSTAMP_S(sync->stamp, sync->stampN, slot, SPW); } while (0)
#define WBC_QSTAMP3(slot) do { if (!RHAT && sync && sync->stamp && (tx >> 6) == 3) WBC_FSTAMP_S(sync->stamp, sync->stampN, slot, SPW); } while (0)
#else
#define WBC_QSTAMP(slot) do {} while (0)
#define WBC_QSTAMP3(slot) do {} while (0)
#endif
WBC_DEV void qp_wait(int* flag, int need) {
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// TILED (qp_tile_kernel below): the four rows of the wavefront solve the states `who` names (dealt by predicted work)
// instead of four consecutive ones; the workgroup is four such wavefronts.
struct QpWho {
  size_t state; bool live;
  const double* pre = nullptr;   // the tile predictor's record of this state (qp_kernels.hip.hpp), LDS
  // STG (staged tiles, qp_kernels.hip.hpp): the tile's inputs wait in an LDS image [ST_WORDS][stride] of the solver's scalar type, the state is column
  // `slot`; f, tau go back into the image and status / iters / active set into iimg ([4][tile]: mask, status, iters, set) -- the body touches no memory
  void* img = nullptr; int* iimg = nullptr; int slot = 0, stride = 0, tile = 0;
  void* tab = nullptr;   // this wavefront's solver tables (S16Lds<double>)
};
// rows of the staged image: normals 12, mu 4, lever arms 12, b = w_des - rhat_base 6, tau_partial - rhat_joint 12 (leg-major), own-leg Jacobian blocks 36
// (9 f + 3 m + k), then the results f 12, tau 12 (caller's joint order)
constexpr int ST_N = 0, ST_MU = 12, ST_D = 16, ST_B = 28, ST_TAUP = 34, ST_JCL = 46, ST_F = 82, ST_TAU = 94, ST_WORDS = 106;
struct QpNoIdle { WBC_DEV void operator()() const {} };
// `idle()` (fused tick) runs after this wavefront has requested its own inputs and before it first waits for the producer roles:
// work that would otherwise sit on a producer's critical path (the structural constants of M and Jc, dyn_split.hip.hpp).
template <class T, bool WSLDS, bool RHAT = false, int SPW = 16, bool TILED = false, int WPB = (WSLDS || TILED) ? 4 : 1, class Idle = QpNoIdle>
WBC_DEV void qp_group16_body(const DevParams<T>& prm, const QpArgs<T>& a, const QpJidx& jmap, const T* wsl, const QpSync* sync = nullptr,
                             const QpWho who = QpWho{0, false}, Idle idle = Idle()) {
  static_assert(!(WSLDS && TILED), "tiles are dealt by the stand-alone kernel only");
  // WPB: wavefronts of the workgroup that run this body (the fused kernels pair their producer wavefronts with four QP wavefronts)
  __shared__ G16Lds<T> lds_all[WPB];
  unsigned tx = threadIdx.x;
  asm volatile("" : "+v"(tx));   // lane-derived predicates stay inside this call (see WBC_LAUNDERED_TID, dyn_split.hip.hpp)
  const int lane = tx & 63;
  const int l16 = lane & 15;
  const int rowbase = lane & 48;   // first lane of my 16-lane row
  const int grp = lane >> 4;
  const int f = l16 >> 2, c3 = l16 & 3;
  const bool isvar = c3 < 3;
  const int v = 3 * f + (isvar ? c3 : 0);  // variable index of this lane (spare lanes: unused)
  T* J0 = lds_all[tx >> 6].J0[grp];
  T* Rl = lds_all[tx >> 6].R[grp] + v;  // my row of R: Rl[12 * position] (spare lanes alias a variable lane; they never write)
  T* Cl = lds_all[tx >> 6].C[grp];        // constraint rows by id
  const size_t N = a.N;
  const unsigned N32 = (unsigned)N;
  // XCD-aware order for one-wave workgroups: workgroups b, b+8, b+16, b+24 land on the same XCD (round-robin dispatch)
  // and take four CONSECUTIVE 4-state groups = one whole 128-byte line per component row.  Speed only.
  size_t wg = blockIdx.x;
  if (WPB == 1 && (gridDim.x & 31) == 0) wg = (wg & ~(size_t)31) + ((wg & 7) << 2) + ((wg >> 3) & 3);
  static_assert(SPW == 16 || WSLDS, "fewer states per workgroup only inside the fused kernels");
  const size_t qp_raw = TILED ? who.state
                              : WSLDS ? (size_t)blockIdx.x * SPW + (tx >> 4)   // fused tick: QP wavefronts are threads 0..255 of a larger workgroup
                                      : (wg * blockDim.x + tx) >> 4;
  bool live = TILED ? who.live : (qp_raw < N && (SPW == 16 || (int)(tx >> 4) < SPW));
  unsigned s32 = (unsigned)(live ? qp_raw : N - 1);
  const T INF = Lim<T>::inf, EPS = Lim<T>::eps;
#define GLD(ptr, comp) (*(const T*)((const char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))))
#define WSLD(comp) (WSLDS ? wsl[(comp) * 16 + (int)(tx >> 4)] : GLD(a.ws, comp))
  // target wrench: LDS image (roles), the caller's w_des (two-kernel ticks whose front half does not change it) or the workspace
#define BLD(c) (WSLDS ? wsl[(WS_B + (c)) * 16 + (int)(tx >> 4)] : (a.wdes ? GLD(a.wdes, c) : GLD(a.ws, WS_B + (c))))
#define GST(ptr, comp, val) (*(T*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))) = (val))

#ifdef WBC_QP_STAMP
  const long long st_t0 = __builtin_readcyclecounter();
#endif
  // ------------------------------------------------------------------ inputs
  int mask = a.mask[s32] & 0xF;
  bool on = (mask >> f) & 1;
  const bool geom_jc = !WSLDS && a.Jc != nullptr;   // uniform: lever arms / own-leg blocks from Jc (see QpArgs)
  const T n_ld = isvar ? GLD(a.normals, v) : (T)0;
  const T mu_f = GLD(a.mu, f);
  WBC_QSTAMP(1);
  idle();
  if constexpr (WSLDS) { if (sync) qp_wait(sync->geom, sync->need_geom); }
  WBC_QSTAMP(2);
  T d_me = 0;
  if (geom_jc) {   // -[d]x block of my foot's Jacobian rows: d_x = Jc[(3f+1), 5], d_y = Jc[(3f+2), 3], d_z = Jc[(3f), 4]
    const int comp = c3 == 0 ? (3 * f + 1) * 18 + 5 : (c3 == 1 ? (3 * f + 2) * 18 + 3 : (3 * f) * 18 + 4);
    if (isvar) d_me = GLD(a.Jc, comp);
  } else if (isvar) d_me = WSLD(WS_D + v);

  // ------------------------------------------------------------------ a square root of H^-1 (a7), from the structure of H
  // H = alpha I + B^T B,  B = S^(1/2) A (6 x 12),  A = [I ; [d_f]x] per stance foot (zero columns for swing feet).
  // The dual active-set method needs ANY J with J J^T = H^-1 -- it only ever applies orthogonal transformations to the
  // columns of J -- so the dense 12x12 Cholesky factor and its inverse are not needed.  With the 6x6 matrix
  //     G = alpha I + B B^T = L L^T                         (B B^T = S^(1/2) [nc I, -[sd]x ; [sd]x, sum(|d|^2 I - d d^T)] S^(1/2))
  //     J = (I - B^T K B) / sqrt(alpha),   K = (G + sqrt(alpha) L^T)^-1 = L^-T (L + sqrt(alpha) I)^-1
  // because J J^T = (I - B^T (K + K^T - K (G - alpha I) K^T) B) / alpha and K + K^T - K (G - alpha I) K^T = G^-1 (Woodbury).
  // Every lane forms G and its factor redundantly from the twelve broadcast lever-arm components -- no cross-lane step,
  // six pivots instead of twelve -- then applies K and K^T to ITS column b_v of B; entries of J are three FMAs each.
  // Accuracy: the alpha-dominated directions are handled analytically, which in fp32 is ~100x closer to H^-1 than the
  // 12x12 Cholesky route (cond(H) ~ 1e4).
  const T dqx = dppx<0x00>(d_me), dqy = dppx<0x55>(d_me), dqz = dppx<0xAA>(d_me);  // quad_perm [0000],[1111],[2222]: my foot's lever arm
  const T onv = (on && isvar) ? (T)1 : (T)0;
  T u0, u1, u2;  // column c3 of [d_f]x for my own foot: [[0,-dz,dy],[dz,0,-dx],[-dy,dx,0]]
  u0 = (c3 == 0) ? (T)0 : (c3 == 1 ? -dqz : dqy);
  u1 = (c3 == 0) ? dqz : (c3 == 1 ? (T)0 : -dqx);
  u2 = (c3 == 0) ? -dqy : (c3 == 1 ? dqx : (T)0);
  // (the weights pass through an empty asm: they are kernel-uniform, so inside the persistent rollout kernel and the tile
  // loop every product of two of them would otherwise be hoisted out of the loop and held in registers for the whole kernel)
  T s0 = prm.sS[0], s1 = prm.sS[1], s2 = prm.sS[2], s3 = prm.sS[3], s4 = prm.sS[4], s5 = prm.sS[5];
  T alpha_l = prm.alpha, sqa = prm.sqrt_alpha, rsa = prm.rsqrt_alpha;
  if constexpr (WSLDS || TILED) asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(alpha_l), "+v"(sqa), "+v"(rsa));
  // my column of B: bf on the force rows (one non-zero, at row c3), bm on the moment rows
  const T bfs = onv * (c3 == 0 ? s0 : (c3 == 1 ? s1 : s2));
  const T bm0 = onv * s3 * u0, bm1 = onv * s4 * u1, bm2 = onv * s5 * u2;
  T Dx[4], Dy[4], Dz[4], Of[4];   // lever arms of the four feet, zeroed for swing feet; stance flags
  sfor<0, 4>([&](auto fc) __attribute__((always_inline)) {
    constexpr int fj = decltype(fc)::value;
    const bool onj = (mask >> fj) & 1;
    Of[fj] = onj ? (T)1 : (T)0;
    Dx[fj] = onj ? gbc<3 * fj>(d_me) : (T)0; Dy[fj] = onj ? gbc<3 * fj + 1>(d_me) : (T)0; Dz[fj] = onj ? gbc<3 * fj + 2>(d_me) : (T)0;
  });
  // lower factor L of G: rows 0..2 diagonal (l0 l1 l2); row 3: [0 a01 a02 | l3]; row 4: [a10 0 a12 | b10 l4]; row 5: [a20 a21 0 | b20 b21 l5]
  T a01, a02, a10, a12, a20, a21, b10, b20, b21;
  T il[6], ld[6];   // 1 / L_kk and L_kk
  {
    const T nc = (Of[0] + Of[1]) + (Of[2] + Of[3]);
    const T sx = (Dx[0] + Dx[1]) + (Dx[2] + Dx[3]), sy = (Dy[0] + Dy[1]) + (Dy[2] + Dy[3]), sz = (Dz[0] + Dz[1]) + (Dz[2] + Dz[3]);
    T Pxx = 0, Pxy = 0, Pxz = 0, Pyy = 0, Pyz = 0, Pzz = 0;
    sfor<0, 4>([&](auto fc) __attribute__((always_inline)) {
      constexpr int fj = decltype(fc)::value;
      Pxx += Dx[fj] * Dx[fj]; Pxy += Dx[fj] * Dy[fj]; Pxz += Dx[fj] * Dz[fj];
      Pyy += Dy[fj] * Dy[fj]; Pyz += Dy[fj] * Dz[fj]; Pzz += Dz[fj] * Dz[fj];
    });
    const T g00 = alpha_l + s0 * s0 * nc, g11 = alpha_l + s1 * s1 * nc, g22 = alpha_l + s2 * s2 * nc;
    // moment-force block s_(3+i) s_j [sd]x_ij, moment-moment block alpha I + s_(3+i) s_(3+j) (|d|^2 I - d d^T)_ij summed over stance feet
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
    ld[0] = g00 * il[0]; ld[1] = g11 * il[1]; ld[2] = g22 * il[2]; ld[3] = c00 * il[3]; ld[4] = t11 * il[4]; ld[5] = t22 * il[5];
  }
  // w = (L [+ sqrt(alpha) I])^-1 r  and  y = (L [+ sqrt(alpha) I])^-T w, the diagonal given through its reciprocals `inv`
  auto fwd = [&](const T* inv, T r0, T r1, T r2, T r3, T r4, T r5, T* w) __attribute__((always_inline)) {
    w[0] = r0 * inv[0]; w[1] = r1 * inv[1]; w[2] = r2 * inv[2];
    w[3] = (r3 - a01 * w[1] - a02 * w[2]) * inv[3];
    w[4] = (r4 - a10 * w[0] - a12 * w[2] - b10 * w[3]) * inv[4];
    w[5] = (r5 - a20 * w[0] - a21 * w[1] - b20 * w[3] - b21 * w[4]) * inv[5];
  };
  auto bwd = [&](const T* inv, const T* w, T* y) __attribute__((always_inline)) {
    y[5] = w[5] * inv[5];
    y[4] = (w[4] - b21 * y[5]) * inv[4];
    y[3] = (w[3] - b10 * y[4] - b20 * y[5]) * inv[3];
    y[2] = (w[2] - a02 * y[3] - a12 * y[4]) * inv[2];
    y[1] = (w[1] - a01 * y[3] - a21 * y[5]) * inv[1];
    y[0] = (w[0] - a10 * y[4] - a20 * y[5]) * inv[0];
  };
  const T bf0 = (c3 == 0) ? bfs : (T)0, bf1 = (c3 == 1) ? bfs : (T)0, bf2 = (c3 == 2) ? bfs : (T)0;
  // ------------------------------------------------------------------ unconstrained minimum x0 = -H^-1 g = B^T G^-1 S^(1/2) b
  // (b = w_des - rhat_base).  Stand-alone kernel: b is there, one 6x6 solve with the factor at hand.  Inside the fused
  // kernels (WSLDS) b may still be on its way from the observer role: x0 = -J J^T g is formed later from J alone, after
  // the constraint rows below are set up (those need the terrain only).
  T x_me = 0;
  if constexpr (!WSLDS) {
    const T b_ld = (l16 < 6) ? BLD(l16) - (RHAT ? WSLD(WS_RHAT + l16) : (T)0) : (T)0;
    T w[6], z[6];
    fwd(il, s0 * dppx<0x150 + 0>(b_ld), s1 * dppx<0x150 + 1>(b_ld), s2 * dppx<0x150 + 2>(b_ld), s3 * dppx<0x150 + 3>(b_ld),
        s4 * dppx<0x150 + 4>(b_ld), s5 * dppx<0x150 + 5>(b_ld), w);
    bwd(il, w, z);
    x_me = (bf0 * z[0] + bf1 * z[1] + bf2 * z[2]) + (bm0 * z[3] + bm1 * z[4] + bm2 * z[5]);
  }
  // ------------------------------------------------------------------ my column and my row of J
  // (stand-alone kernels build J only when some row of the wavefront has a violated constraint at x0: a wavefront whose
  // four unconstrained minima are feasible -- common once tiles are dealt by predicted work -- is done after the 6x6 solve)
  T Jc[12], Jr[12];
  auto build_J = [&]() __attribute__((always_inline)) {
    T ia[6];   // 1 / (L_kk + sqrt(alpha))
    sfor<0, 6>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; ia[k] = rcp_nr(ld[k] + sqa); });
    T w[6], y[6], yt[6];
    fwd(ia, bf0, bf1, bf2, bm0, bm1, bm2, w);   // y = K b_v = L^-T (L + sqrt(alpha) I)^-1 b_v
    bwd(il, w, y);
    fwd(il, bf0, bf1, bf2, bm0, bm1, bm2, w);   // yt = K^T b_v = (L + sqrt(alpha) I)^-T L^-1 b_v
    bwd(ia, w, yt);
    const T yf[3] = {s0 * rsa * y[0], s1 * rsa * y[1], s2 * rsa * y[2]}, ym[3] = {s3 * rsa * y[3], s4 * rsa * y[4], s5 * rsa * y[5]};
    const T tf[3] = {s0 * rsa * yt[0], s1 * rsa * yt[1], s2 * rsa * yt[2]}, tm[3] = {s3 * rsa * yt[3], s4 * rsa * yt[4], s5 * rsa * yt[5]};
    sfor<0, 12>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr int fi = i / 3, ai = i % 3;
      const T dlt = (isvar && v == i) ? rsa : (T)0;
      // b_i . y with b_i = on_i [s_ai e_ai ; s_(3..5) * column ai of [d_i]x]   (Dx.. are already zero for swing feet)
      T cj = dlt - Of[fi] * yf[ai], rj = dlt - Of[fi] * tf[ai];
      if (ai == 0) { cj = cj - Dz[fi] * ym[1] + Dy[fi] * ym[2]; rj = rj - Dz[fi] * tm[1] + Dy[fi] * tm[2]; }
      else if (ai == 1) { cj = cj + Dz[fi] * ym[0] - Dx[fi] * ym[2]; rj = rj + Dz[fi] * tm[0] - Dx[fi] * tm[2]; }
      else { cj = cj - Dy[fi] * ym[0] + Dx[fi] * ym[1]; rj = rj - Dy[fi] * tm[0] + Dx[fi] * tm[1]; }
      Jc[i] = cj;   // J[i][v]
      Jr[i] = rj;   // J[v][i]
    });
    // keep the initial J in LDS: dropping a constraint (rare) restores it from there
    if (isvar) sfor<0, 12>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; J0[i * 12 + v] = Jc[i]; });
  };
  if constexpr (WSLDS) build_J();   // fused kernels: J is built while b is still on its way
  auto solve_x0 = [&]() __attribute__((always_inline)) {   // fused kernels only: x0 = -J J^T g once b has arrived
    WBC_QSTAMP(3);
    if constexpr (WSLDS) { if (sync) qp_wait(sync->geom, sync->need_b); }
    if constexpr (WSLDS && RHAT) { if (sync) qp_wait(sync->rhat, sync->need_rhat); }
    WBC_QSTAMP(4);
    const T b_ld = (l16 < 6) ? BLD(l16) - (RHAT ? WSLD(WS_RHAT + l16) : (T)0) : (T)0;
    T b[6];
    b[0] = dppx<0x150 + 0>(b_ld); b[1] = dppx<0x150 + 1>(b_ld); b[2] = dppx<0x150 + 2>(b_ld);
    b[3] = dppx<0x150 + 3>(b_ld); b[4] = dppx<0x150 + 4>(b_ld); b[5] = dppx<0x150 + 5>(b_ld);
    // g = -A^T S b = -B^T S^(1/2) b
    const T g_me = -((bf0 * (s0 * b[0]) + bf1 * (s1 * b[1]) + bf2 * (s2 * b[2])) + (bm0 * (s3 * b[3]) + bm1 * (s4 * b[4]) + bm2 * (s5 * b[5])));
    T t_me = 0;
    sfor<0, 12>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; t_me += Jc[i] * gbc<i>(g_me); });
    T acc = 0;
    sfor<0, 12>([&](auto cc) __attribute__((always_inline)) { constexpr int c = decltype(cc)::value; acc += Jr[c] * gbc<c>(t_me); });
    x_me = -acc;
  };

  // ------------------------------------------------------------------ my constraints (friction pyramid, force box)
  T cAx, cAy, cAz, rA, cBx = 0, cBy = 0, cBz = 0;
  const bool hasB = c3 < 2;
  {
    T nx = dppx<0x00>(n_ld), ny = dppx<0x55>(n_ld), nz = dppx<0xAA>(n_ld);
    const T il = rsqrt_nr(nx * nx + ny * ny + nz * nz);
    nx *= il; ny *= il; nz *= il;
    const bool usex = fabs_t(nx) < (T)0.9;
    const T rx = usex ? (T)1 : (T)0, ry = usex ? (T)0 : (T)1;
    const T rd = rx * nx + ry * ny;
    T t1x = rx - nx * rd, t1y = ry - ny * rd, t1z = -nz * rd;
    const T it = rsqrt_nr(t1x * t1x + t1y * t1y + t1z * t1z);
    t1x *= it; t1y *= it; t1z *= it;
    const T t2x = ny * t1z - nz * t1y, t2y = nz * t1x - nx * t1z, t2z = nx * t1y - ny * t1x;
    const T mt = mu_f * prm.mu_scale;
    const T tx = (c3 == 0) ? t1x : t2x, ty = (c3 == 0) ? t1y : t2y, tz = (c3 == 0) ? t1z : t2z;
    if (hasB) {
      cAx = mt * nx - tx; cAy = mt * ny - ty; cAz = mt * nz - tz; rA = 0;
      cBx = mt * nx + tx; cBy = mt * ny + ty; cBz = mt * nz + tz;
    } else if (c3 == 2) { cAx = nx; cAy = ny; cAz = nz; rA = prm.fn_min; }
    else { cAx = -nx; cAy = -ny; cAz = -nz; rA = -prm.fn_max; }
  }

  {
    T* c = Cl + 3 * (2 * l16);
    c[0] = cAx; c[1] = cAy; c[2] = cAz; c[3] = cBx; c[4] = cBy; c[5] = cBz;
    // the image of U starts at zero for EVERY QP: columns beyond the active set enter the dual step multiplied by a masked
    // zero, so whatever they hold must be finite -- a NaN left by an earlier state of this row slot (tiles, rollouts) must not leak
    if (isvar) sfor<0, 12>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; Rl[12 * k] = (T)0; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // (LDS hand-over inside the wavefront: no global traffic to wait for)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    if constexpr (WSLDS) solve_x0();
  }

  // ------------------------------------------------------------------ dual active-set iterations (a8)
  int iq = 0, ip = -1, status = 0, iter = 0;
  bool done = !live;
  bool actA = false, actB = false;
  T sip = 0, Rnorm = 1, u_me = 0;
  T u_c = 0;  // multiplier of the candidate constraint (row-uniform; it has no position until it is added)
  int Aid = -1;

  // slack of my constraints at the current x
  auto slacks = [&](T& sA, T& sB) __attribute__((always_inline)) {
    const T xq0 = dppx<0x00>(x_me), xq1 = dppx<0x55>(x_me), xq2 = dppx<0xAA>(x_me);
    sA = cAx * xq0 + cAy * xq1 + cAz * xq2 - rA;
    sB = cBx * xq0 + cBy * xq1 + cBz * xq2;
  };
  // step 1 for rows that need a new candidate: most violated inactive constraint
  auto pick = [&]() __attribute__((always_inline)) {
    T sA, sB;
    slacks(sA, sB);
    T val = KeyT<T>::BIG;
    int id = 2 * l16;
    if (on && !actA && sA < -prm.qp_tol) { val = sA; }
    if (on && hasB && !actB && sB < -prm.qp_tol && sB < val) { val = sB; id = 2 * l16 + 1; }
    const bool found = gargmin(val, id);
    const T sw = val;  // winner's slack, exact to 2^-46
    const bool need = !done && ip < 0;
    if (need) {
      if (found) {
        ip = id; sip = sw; u_c = 0;
      } else done = true;
    }
  };
  // add constraint with normal (np0,np1,np2) on foot fp at position `pos` for rows where `doit`;
  // dd = J^T np of my column (valid when isvar), dn2 = |dd[pos..)|^2.  Householder on J[:, pos..12).
  auto add_column = [&](bool doit, int pos, T dd, T dn2, T& nr_out, T r_in) __attribute__((always_inline)) {
    const T a0 = gsum((isvar && v == pos) ? dd : (T)0);  // d[pos]: row sum of a one-hot, no LDS round trip
    const T inr = rsqrt_nr(dn2 > 0 ? dn2 : (T)1);
    const T nr = dn2 * inr;  // |d2| = dn2 / sqrt(dn2)
    const T sg = (a0 >= 0) ? (T)1 : (T)-1;
    const T beta = rcp_nr(nr * (nr + fabs_t(a0)));
    T w_me = 0;
    if (doit && isvar) w_me = (v == pos) ? a0 + sg * nr : (v > pos ? dd : (T)0);
    T ya[3] = {0, 0, 0};
    sfor<0, 12>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; ya[j % 3] += Jr[j] * gbc<j>(w_me); });
    T y_me = (ya[0] + ya[1]) + ya[2];
    y_me = doit ? y_me * beta : (T)0;
    sfor<0, 12>([&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; Jr[j] -= y_me * gbc<j>(w_me); });
    sfor<0, 12>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; Jc[i] -= gbc<i>(y_me) * w_me; });
    const T rdn = -sg * inr;   // 1 / (new diagonal entry of R)
    if (doit && isvar) Rl[12 * pos] = (v < pos) ? -r_in * rdn : (v == pos ? rdn : (T)0);   // column `pos` of U = R^-1, every row
    nr_out = nr;
  };
  // d = J^T np for my column, np = (n0,n1,n2) on the variables of foot fp
  auto jt_np = [&](int fp, T n0, T n1, T n2) __attribute__((always_inline)) -> T {
    // foot selection by 0/1 weights, not by a select chain: clang turns `fp==0 ? Jc[0] : fp==1 ? Jc[3] ...` into a
    // run-time index, which forces the whole register array into scratch memory
    const T m0 = fp == 0 ? (T)1 : (T)0, m1 = fp == 1 ? (T)1 : (T)0, m2 = fp == 2 ? (T)1 : (T)0, m3 = fp == 3 ? (T)1 : (T)0;
    const T j0 = m0 * Jc[0] + m1 * Jc[3] + m2 * Jc[6] + m3 * Jc[9];
    const T j1 = m0 * Jc[1] + m1 * Jc[4] + m2 * Jc[7] + m3 * Jc[10];
    const T j2 = m0 * Jc[2] + m1 * Jc[5] + m2 * Jc[8] + m3 * Jc[11];
    return isvar ? j0 * n0 + j1 * n1 + j2 * n2 : (T)0;
  };
  // r = U d1 (U = R^-1 by rows, d1 = d[0 .. q)); kmax = wave-uniform bound on q
  auto dual_step = [&](T dd, int q, int kmax) __attribute__((always_inline)) -> T {
    T Urow[12];
    sfor<0, 12>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; Urow[k] = Rl[12 * k]; });
    const T ddm = (isvar && v < q) ? dd : (T)0;   // masked at the source lane: columns >= q of the image are stale
    T ra[3] = {0, 0, 0};
    sfor<0, 12>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      if (k < kmax) ra[k % 3] += Urow[k] * gbc<k>(ddm);
    });
    return (isvar && v < q) ? (ra[0] + ra[1]) + ra[2] : (T)0;
  };
  // normal of constraint id: its three coefficients live in lane (id >> 1)
  auto normal_of = [&](int id, T& n0, T& n1, T& n2) __attribute__((always_inline)) {
    const T* c = Cl + 3 * (id & 31);
    n0 = c[0]; n1 = c[1]; n2 = c[2];
  };

#ifdef WBC_QP_STAMP
  const long long st_t1 = __builtin_readcyclecounter();
#endif
  pick();
  if constexpr (!WSLDS) { if (__ballot(!done && ip >= 0) != 0ull) build_J(); }
  int guard = 0;
#ifdef WBC_QP_STAMP
  long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long st_last = __builtin_readcyclecounter();
#define SEG(i) do { const long long now_ = __builtin_readcyclecounter(); seg[i] += now_ - st_last; st_last = now_; } while (0)
#else
#define SEG(i) do {} while (0)
#endif
  while (__ballot(!done && ip >= 0) != 0ull) {
    if (++guard > 4 * prm.max_iter + 8) break;  // hard stop; per-row limits are enforced below
    bool go = !done && ip >= 0;
    if (go && ++iter > prm.max_iter) { status = 1; done = true; go = false; }
    const int ipc = ip < 0 ? 0 : ip;
    const int lp = (ipc >> 1) & 15, fp = lp >> 2;
    T np0, np1, np2;
    normal_of(ipc, np0, np1, np2);
    const T dd = jt_np(fp, np0, np1, np2);
    const T dn2 = gsum((isvar && v >= iq) ? dd * dd : (T)0);
    SEG(0);
    // z = J2 d2 (row copy, d broadcast and masked below iq)
    T z_me;
    {
      T za[3] = {0, 0, 0};  // three independent chains instead of one 12-deep FMA chain
      const T ddz = (isvar && v >= iq) ? dd : (T)0;  // mask at the source lane: one select instead of twelve
      sfor<0, 12>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        za[j % 3] += Jr[j] * gbc<j>(ddz);
      });
      z_me = (za[0] + za[1]) + za[2];
    }
    SEG(1);
    // r = R^-1 d1 : column-oriented back-substitution, row k of R lives in variable lane k
    T r_me = 0;
    {
      const int iqg = go ? iq : 0;
      int iqmax = __builtin_amdgcn_readlane(iqg, 0);
      { const int b1 = __builtin_amdgcn_readlane(iqg, 16), b2 = __builtin_amdgcn_readlane(iqg, 32), b3 = __builtin_amdgcn_readlane(iqg, 48);
        iqmax = iqmax > b1 ? iqmax : b1; iqmax = iqmax > b2 ? iqmax : b2; iqmax = iqmax > b3 ? iqmax : b3; }
      r_me = dual_step(dd, iq, iqmax);
    }
    SEG(2);
    // step lengths
    T t1 = INF;
    int kmin = isvar ? v : 15;
    {
      T t1k = KeyT<T>::BIG;
      if (isvar && v < iq && r_me > 0) t1k = u_me * rcp_nr(r_me);
      const bool found = gargmin(t1k, kmin);
      if (found) t1 = t1k;
    }
    T t2 = INF;
    if (dn2 > (EPS * Rnorm) * (EPS * Rnorm)) t2 = -sip * rcp_nr(dn2);  // z.np = |d2|^2
    if (go && !(t1 < INF) && !(t2 < INF)) { status = 2; done = true; go = false; }
    const bool dual_only = !(t2 < INF);
    const bool full = !dual_only && !(t1 < t2);
    const T t = full ? t2 : t1;
    if (go) {
      if (!dual_only) x_me += t * z_me;
      if (isvar && v < iq) u_me -= t * r_me;
      u_c += t;
    }
    SEG(3);
    // ---- full step: the candidate joins the active set
    // Executed unconditionally (predicated per row) and followed directly by the search for the next candidate, so
    // that the factor update and the independent slack/argmin chain share one basic block and interleave.
    const bool addg = go && full;
    {
      T nr;
      add_column(addg, iq, dd, dn2, nr, r_me);
      if (addg) {
        Rnorm = (nr > Rnorm) ? nr : Rnorm;
        if (l16 == lp) { if (ipc & 1) actB = true; else actA = true; }
        if (isvar && v == iq) { u_me = u_c; Aid = ipc; }
        ++iq;
        ip = -1;
      }
    }
    // rows that just added look for their next candidate now (rows that drop keep theirs: ip >= 0 there)
    pick();
    SEG(4);
    // ---- partial / dual-only step: the blocking constraint leaves, factors are rebuilt
    const bool dropg = go && !full;
    if (__ballot(dropg) != 0ull) {
      const int kq = dropg ? kmin : 0;
      const int cid = gread(Aid, (kq + kq / 3) & 15, rowbase);
      if (dropg && l16 == ((cid >> 1) & 15)) { if (cid & 1) actB = false; else actA = false; }
      // shift multipliers and ids down over the hole (positions kq+1..iq-1 move to kq..iq-2)
      {
        const int nxt = v + 1;
        const T un = gread(u_me, (nxt + nxt / 3) & 15, rowbase);
        const int An = gread(Aid, (nxt + nxt / 3) & 15, rowbase);
        if (dropg && isvar && v >= kq && v < iq - 1) { u_me = un; Aid = An; }
        if (dropg && isvar && v == iq - 1) { u_me = 0; Aid = -1; }
      }
      if (dropg) --iq;
      // restore J0 and re-add the remaining active constraints in order
      if (dropg) {
        if (isvar) {
          sfor<0, 12>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; Jc[i] = J0[i * 12 + v]; });
          sfor<0, 12>([&](auto cc) __attribute__((always_inline)) { constexpr int c = decltype(cc)::value; Jr[c] = J0[v * 12 + c]; });
        }
        Rnorm = 1;
      }
      for (int p = 0; p < 11; ++p) {
        const bool rg = dropg && p < iq;
        if (__ballot(rg) == 0ull) break;
        const int idp = gread(Aid, (p + p / 3) & 15, rowbase);
        const int idc = rg ? idp : 0;
        T m0, m1, m2;
        normal_of(idc, m0, m1, m2);
        const T dp = jt_np(((idc >> 1) & 15) >> 2, m0, m1, m2);
        const T dp2 = gsum((isvar && v >= p) ? dp * dp : (T)0);
        T nr;
        const T rp = dual_step(dp, rg ? p : 0, p);
        add_column(rg, p, dp, dp2, nr, rp);
        if (rg) Rnorm = (nr > Rnorm) ? nr : Rnorm;
      }
      {  // a partial step moved x: refresh the candidate's slack (cross-lane ops stay unconditional)
        T sA, sB;
        slacks(sA, sB);
        const T sv = gread((ipc & 1) ? sB : sA, lp, rowbase);
        if (dropg && !dual_only) sip = sv;
      }
    }
    SEG(5);
  }

  // ------------------------------------------------------------------ outputs: f, tau (a9), status
  WBC_QSTAMP(5);
  WBC_QSTAMP3(10);
  if constexpr (WSLDS) { if (sync) qp_wait(sync->fin, sync->need_fin); }
  WBC_QSTAMP(6);
  bool to_mem = true;   // (QpSync::skip_out: wavefront-uniform)
  bool from_hand = false;   // (QpSync::hand)
  const T* hand_img = nullptr;
  if constexpr (WSLDS) {
    if (sync) {
      to_mem = !sync->skip_out;
      if (sync->hand) { qp_wait(sync->hand_flag, sync->need_hand); from_hand = true; hand_img = (const T*)sync->hand; }
    }
  }
  if (live) {
    T taup = 0, jl0 = 0, jl1 = 0, jl2 = 0;  // own-leg Jacobian entries d pf_m / d q_(f,c3)
    int jm = 0;   // caller's index of my joint (leg f, joint c3)
    jm = (int)((unsigned)(a.jpack >> (4 * (v & 15))) & 15u);
    if (isvar) {
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
      if (to_mem) {
        GST(a.f, v, on ? x_me : (T)0);
        GST(a.tau, jm, taup - (jl0 * fx + jl1 * fy + jl2 * fz));
      }
      if constexpr (WSLDS) {
        if (sync && sync->res) {
          T* rs = (T*)sync->res + (int)(tx >> 4);
          rs[(RES_F + v) * 16] = on ? x_me : (T)0;
          rs[(RES_TAU + jm) * 16] = taup - (jl0 * fx + jl1 * fy + jl2 * fz);
        }
      }
    }
    if (l16 == 0 && to_mem) {
      a.status[s32] = status;
#ifdef WBC_QP_STAMP  // diagnostic build only: per-wave cycle stamps (two 16-bit fields of cycles/16 per group) instead of iteration counts
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
#undef GST
#undef WSLD
#undef BLD
#undef GLD
}

}  // namespace wbc

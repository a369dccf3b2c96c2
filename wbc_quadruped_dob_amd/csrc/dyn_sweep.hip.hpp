// Dynamics-sweep kernel for gfx950 (CDNA4): SURVEY.md 8(a) units a1-a7/a9-prologue for one batch.
//
// Mapping: ONE LANE PER LEG.  A quadruped's four legs are independent subtrees of the floating
// base; lane 16*leg + s of a wave owns leg `leg` of the wave's state s (16 states per 64-wide wave).
// Everything leg-local (joint transforms, RNEA/CRBA/momentum sweeps up and down the 3-joint
// chain, the leg's Jacobian block) runs without communication; the only cross-lane traffic is
// the leaf->root accumulation into the base (leg wrench, leg momentum, leg composite inertia),
// done with v_permlane16/32_swap row exchanges (gfx950) -- no LDS, no barriers.  Per-leg model constants are staged once
// per block in LDS as cst[word][leg] (four consecutive words per row: conflict-free).
// HBM layout is component-major x[c*N+s]: every load/store instruction of a wave touches four
// fully used 128-byte lines (16 consecutive states x 8 B per leg-specific component), base
// components are distributed over the quad with v_cndmask so no lane issues a redundant store.
//
// Algorithms (textbook; the reference's own dynamics library is in an absent submodule):
// Featherstone RNEA / CRBA in link coordinates with (m, h, Io) rigid-body inertias, mixed
// ("world-aligned at the base origin") generalized velocity -- docs/DESIGN_R04.md section 3.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace wbc {


template <class T> struct V3 { T x, y, z; };
template <class T> WBC_DEV V3<T> mk(T x, T y, T z) { V3<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <class T> WBC_DEV V3<T> operator+(V3<T> a, V3<T> b) { return mk<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> WBC_DEV V3<T> operator-(V3<T> a, V3<T> b) { return mk<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> WBC_DEV V3<T> operator*(V3<T> a, T s) { return mk<T>(a.x * s, a.y * s, a.z * s); }
template <class T> WBC_DEV T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> WBC_DEV V3<T> cross(V3<T> a, V3<T> b) {
  return mk<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// 3x3 row-major
template <class T> struct M3 { T a[9]; };
template <class T> WBC_DEV V3<T> mul(const M3<T>& m, V3<T> v) {
  return mk<T>(m.a[0] * v.x + m.a[1] * v.y + m.a[2] * v.z, m.a[3] * v.x + m.a[4] * v.y + m.a[5] * v.z,
               m.a[6] * v.x + m.a[7] * v.y + m.a[8] * v.z);
}
template <class T> WBC_DEV V3<T> tmul(const M3<T>& m, V3<T> v) {  // m^T v
  return mk<T>(m.a[0] * v.x + m.a[3] * v.y + m.a[6] * v.z, m.a[1] * v.x + m.a[4] * v.y + m.a[7] * v.z,
               m.a[2] * v.x + m.a[5] * v.y + m.a[8] * v.z);
}
// symmetric 3x3: xx,xy,xz,yy,yz,zz
template <class T> struct S3 { T xx, xy, xz, yy, yz, zz; };
template <class T> WBC_DEV V3<T> mul(const S3<T>& s, V3<T> v) {
  return mk<T>(s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z,
               s.xz * v.x + s.yz * v.y + s.zz * v.z);
}
// E S E^T for rotation E
template <class T> WBC_DEV S3<T> congr(const M3<T>& E, const S3<T>& s) {
  T t[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    t[3 * i + 0] = E.a[3 * i] * s.xx + E.a[3 * i + 1] * s.xy + E.a[3 * i + 2] * s.xz;
    t[3 * i + 1] = E.a[3 * i] * s.xy + E.a[3 * i + 1] * s.yy + E.a[3 * i + 2] * s.yz;
    t[3 * i + 2] = E.a[3 * i] * s.xz + E.a[3 * i + 1] * s.yz + E.a[3 * i + 2] * s.zz;
  }
  S3<T> r;
  r.xx = t[0] * E.a[0] + t[1] * E.a[1] + t[2] * E.a[2];
  r.xy = t[0] * E.a[3] + t[1] * E.a[4] + t[2] * E.a[5];
  r.xz = t[0] * E.a[6] + t[1] * E.a[7] + t[2] * E.a[8];
  r.yy = t[3] * E.a[3] + t[4] * E.a[4] + t[5] * E.a[5];
  r.yz = t[3] * E.a[6] + t[4] * E.a[7] + t[5] * E.a[8];
  r.zz = t[6] * E.a[6] + t[7] * E.a[7] + t[8] * E.a[8];
  return r;
}

// spatial force / momentum pair [moment n ; force f]
template <class T> struct SF { V3<T> n, f; };
// rigid-body inertia (m, h, Io) times motion (w, v)
template <class T> WBC_DEV SF<T> inertia_mul(T m, V3<T> h, const S3<T>& Io, V3<T> w, V3<T> v) {
  SF<T> r;
  r.n = mul(Io, w) + cross(h, v);
  r.f = v * m - cross(h, w);
  return r;
}
// child -> parent force transform: [E n + r x (E f) ; E f]
template <class T> WBC_DEV SF<T> to_parent(const M3<T>& E, V3<T> r, const SF<T>& s) {
  SF<T> o;
  o.f = mul(E, s.f);
  o.n = mul(E, s.n) + cross(r, o.f);
  return o;
}

// ---- DPP quad reductions: the four lanes of a state sum a value without touching LDS ----------
template <int CTRL> WBC_DEV float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL> WBC_DEV double dpp_mov(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <class T> WBC_DEV T quad_sum(T x) {
  x += dpp_mov<0xB1>(x);  // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E>(x);  // quad_perm [2,3,0,1]
  return x;
}
template <class T> WBC_DEV V3<T> quad_sum(V3<T> v) { return mk<T>(quad_sum(v.x), quad_sum(v.y), quad_sum(v.z)); }

// ---- cross-ROW reductions: the four legs of a state sit in the four 16-lane rows of the wave (lane = 16*leg + s),
// so that every 16-lane group reads and writes 16 consecutive states of ONE component row = one whole 128-byte
// line (tools/bw_probe.hip: 5.5-6.3 TB/s for this lane order against 4.1-4.7 TB/s when the legs of a state are
// adjacent lanes).  gfx950's v_permlane16_swap / v_permlane32_swap exchange rows inside the VALU: with both
// operands equal they return (even rows replicated, odd rows replicated), whose sum is the xor-16 / xor-32 sum.
WBC_DEV int xsum16_parts(int x, int& other) {
  const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  other = r[1];
  return r[0];
}
WBC_DEV int xsum32_parts(int x, int& other) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  other = r[1];
  return r[0];
}
WBC_DEV double xrow_sum(double x) {
  {
    int lo = __double2loint(x), hi = __double2hiint(x), lo2, hi2;
    lo = xsum16_parts(lo, lo2); hi = xsum16_parts(hi, hi2);
    x = __hiloint2double(hi, lo) + __hiloint2double(hi2, lo2);
  }
  {
    int lo = __double2loint(x), hi = __double2hiint(x), lo2, hi2;
    lo = xsum32_parts(lo, lo2); hi = xsum32_parts(hi, hi2);
    x = __hiloint2double(hi, lo) + __hiloint2double(hi2, lo2);
  }
  return x;
}
WBC_DEV float xrow_sum(float x) {
  int o;
  int a = xsum16_parts(__float_as_int(x), o);
  x = __int_as_float(a) + __int_as_float(o);
  a = xsum32_parts(__float_as_int(x), o);
  return __int_as_float(a) + __int_as_float(o);
}
template <class T> WBC_DEV V3<T> xrow_sum(V3<T> v) { return mk<T>(xrow_sum(v.x), xrow_sum(v.y), xrow_sum(v.z)); }

// K cross-leg sums at once (v_permlane16/32_swap)
template <class T, int K> WBC_DEV void xrow_sum_k(T (&x)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) x[k] = xrow_sum(x[k]);
}

template <class T> WBC_DEV T sel4(int leg, T a, T b, T c, T d) { return leg == 0 ? a : (leg == 1 ? b : (leg == 2 ? c : d)); }

// The caller's joint indices of a lane's leg.  1 (default, round 5): from the kernel argument jpack (nibble 3 leg + k; SweepArgs / IntegrateArgs /
// RefArgs) -- one 64-bit shift and three bit-field extracts.  0: the per-lane load of DevModel::jidx of rounds 1-4, which sat IN FRONT of the
// joint-state loads of every body: two dependent trips through L2 at the head of every role of the fused tick and of every tick of a rollout.
// (the macro lives in device_types.hpp: the QP bodies use it too)
template <class Model> WBC_DEV void jidx_of_leg(const Model* __restrict__ model, unsigned long long jpack, int leg, int (&jx)[3]) {
  const unsigned t = (unsigned)(jpack >> (12 * leg));
  jx[0] = (int)(t & 15u); jx[1] = (int)((t >> 4) & 15u); jx[2] = (int)((t >> 8) & 15u);
}

// fp64 sin/cos, straight-line (no branches, so the three joints of a leg interleave in the pipeline): two-term
// Cody-Waite reduction by pi/2 with FMA, then the classic degree-13 / degree-14 minimax kernels on |r| <= pi/4
// (coefficients of the well-known fdlibm kernels; < 1 ulp there).  Valid for |x| up to ~1e5 rad -- joint angles.
// ocml's sincos() handles 1e300 through a Payne-Hanek branch and costs ~4x the latency of this form.
WBC_DEV void sincos_t(double x, double* sp, double* cp) {
  const double n = rint(x * 0.6366197723675814);             // 2/pi
  double r = fma(-n, 1.5707963267948966, x);                 // pi/2 high part
  r = fma(-n, 6.123233995736766e-17, r);                     // pi/2 low part
  const double z = r * r;
  double ps = 1.58969099521155010221e-10;
  ps = fma(ps, z, -2.50507602534068634195e-08);
  ps = fma(ps, z, 2.75573137070700676789e-06);
  ps = fma(ps, z, -1.98412698298579493134e-04);
  ps = fma(ps, z, 8.33333333332248946124e-03);
  ps = fma(ps, z, -1.66666666666666324348e-01);
  const double sr = fma(r * z, ps, r);
  double pc = -1.13596475577881948265e-11;
  pc = fma(pc, z, 2.08757232129817482790e-09);
  pc = fma(pc, z, -2.75573143513906633035e-07);
  pc = fma(pc, z, 2.48015872894767294178e-05);
  pc = fma(pc, z, -1.38888888888741095749e-03);
  pc = fma(pc, z, 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = (int)n & 3;
  const double s1 = (q & 1) ? cr : sr, c1 = (q & 1) ? sr : cr;
  *sp = (q & 2) ? -s1 : s1;
  *cp = ((q + 1) & 2) ? -c1 : c1;
}
WBC_DEV void sincos_t(float x, float* s, float* c) { sincosf(x, s, c); }
WBC_DEV double rsqrt_t(double x) { return 1.0 / sqrt(x); }
WBC_DEV float rsqrt_t(float x) { return 1.0f / sqrtf(x); }
// 1 / sqrt(x) from the hardware estimate (v_rsq_f64: ~2^-23 relative) and two Newton steps -- 9 dependent operations (~0.05 us for a lone wavefront) where
// `1.0 / sqrt(x)` compiles to a refined square root followed by a full IEEE division (~45, ~0.25 us).  Within 1-2 ulp of the correctly rounded value.  Used by
// the roles of the 4-state rollout workgroups, where the quaternion normalisation at the head of the rnea role is on the tick's critical chain: 9.21 -> 9.05 us
// per tick.  NOT used elsewhere: the one-launch tick at 4 096 states measured 13.3 -> 13.5 us with it, the large sweeps are HBM-bound
// (profiles/r05t_ab_rsqrt.log).
WBC_DEV double rsqrt_fast(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = __builtin_fma(-x * y, y, 1.0); y = __builtin_fma(0.5 * y, e, y);
  e = __builtin_fma(-x * y, y, 1.0); y = __builtin_fma(0.5 * y, e, y);
  return y;
}
WBC_DEV float rsqrt_fast(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  const float e = __builtin_fmaf(-x * y, y, 1.0f);
  return __builtin_fmaf(0.5f * y, e, y);
}
template <bool FAST, class X> WBC_DEV X rsqrt_sel(X x) { if constexpr (FAST) return rsqrt_fast(x); else return rsqrt_t(x); }

__host__ __device__ constexpr int midx18(int i, int j) { return i * 18 - i * (i - 1) / 2 + (j - i); }

// ---- two fp32 states per lane: a packed pair that the compiler maps onto v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (gfx90a+:
// full rate, two fp32 results per lane and instruction).  Scalars of the model (one VGPR) enter through op_sel broadcasts, so
// a splat costs no register and no move.
typedef float wbc_f2v __attribute__((ext_vector_type(2)));
typedef int wbc_i2v __attribute__((ext_vector_type(2)));
struct alignas(8) Pk2f {
  wbc_f2v v;
  Pk2f() = default;
  WBC_DEV Pk2f(float s) : v{s, s} {}
  WBC_DEV Pk2f(wbc_f2v x) : v(x) {}
  WBC_DEV Pk2f& operator+=(Pk2f o) { v += o.v; return *this; }
  WBC_DEV Pk2f& operator-=(Pk2f o) { v -= o.v; return *this; }
};
WBC_DEV Pk2f operator+(Pk2f a, Pk2f b) { return Pk2f(a.v + b.v); }
WBC_DEV Pk2f operator-(Pk2f a, Pk2f b) { return Pk2f(a.v - b.v); }
WBC_DEV Pk2f operator*(Pk2f a, Pk2f b) { return Pk2f(a.v * b.v); }
WBC_DEV Pk2f operator-(Pk2f a) { return Pk2f(-a.v); }
WBC_DEV Pk2f operator+(Pk2f a, float b) { return a + Pk2f(b); }
WBC_DEV Pk2f operator+(float a, Pk2f b) { return Pk2f(a) + b; }
WBC_DEV Pk2f operator-(Pk2f a, float b) { return a - Pk2f(b); }
WBC_DEV Pk2f operator-(float a, Pk2f b) { return Pk2f(a) - b; }
WBC_DEV Pk2f operator*(Pk2f a, float b) { return a * Pk2f(b); }
WBC_DEV Pk2f operator*(float a, Pk2f b) { return Pk2f(a) * b; }
template <class T, int W> struct LaneT { static_assert(W == 1, "two states per lane: fp32 only"); using type = T; };
template <> struct LaneT<float, 2> { using type = Pk2f; };
WBC_DEV Pk2f xrow_sum(Pk2f x) { wbc_f2v r; r.x = xrow_sum(x.v.x); r.y = xrow_sum(x.v.y); return Pk2f(r); }
// sin / cos of a pair, straight-line: two-term Cody-Waite reduction by pi/2 with FMA, the classic single-precision minimax
// kernels on |r| <= pi/4 (cephes sinf / cosf coefficients, < 1 ulp there), quadrant fix-up on the bits.  Joint angles only
// (|x| up to ~1e4 rad keeps the reduction exact enough: the second term carries 24 more bits of pi/2).
WBC_DEV void sincos_t(Pk2f x, Pk2f* sp, Pk2f* cp) {
  const wbc_f2v xv = x.v;
  wbc_f2v n;
  n.x = __builtin_rintf(xv.x * 0.63661977236758134f); n.y = __builtin_rintf(xv.y * 0.63661977236758134f);
  wbc_f2v r = __builtin_elementwise_fma(-n, wbc_f2v{1.5707963705062866f, 1.5707963705062866f}, xv);
  r = __builtin_elementwise_fma(-n, wbc_f2v{-4.3711388286737929e-08f, -4.3711388286737929e-08f}, r);
  const wbc_f2v z = r * r;
  const auto K = [](float c) __attribute__((always_inline)) { return wbc_f2v{c, c}; };
  wbc_f2v ps = __builtin_elementwise_fma(K(-1.9515295891e-4f), z, K(8.3321608736e-3f));
  ps = __builtin_elementwise_fma(ps, z, K(-1.6666654611e-1f));
  const wbc_f2v sr = __builtin_elementwise_fma(r * z, ps, r);
  wbc_f2v pc = __builtin_elementwise_fma(K(2.443315711809948e-5f), z, K(-1.388731625493765e-3f));
  pc = __builtin_elementwise_fma(pc, z, K(4.166664568298827e-2f));
  const wbc_f2v cr = __builtin_elementwise_fma(z * z, pc, __builtin_elementwise_fma(K(-0.5f), z, K(1.0f)));
  wbc_i2v q;
  q.x = (int)n.x; q.y = (int)n.y;
  wbc_f2v s1, c1;
  s1.x = (q.x & 1) ? cr.x : sr.x; s1.y = (q.y & 1) ? cr.y : sr.y;
  c1.x = (q.x & 1) ? sr.x : cr.x; c1.y = (q.y & 1) ? sr.y : cr.y;
  wbc_f2v so, co;   // sign: sin flips in quadrants 2, 3; cos in quadrants 1, 2
  so.x = __int_as_float(__float_as_int(s1.x) ^ ((q.x & 2) << 30)); so.y = __int_as_float(__float_as_int(s1.y) ^ ((q.y & 2) << 30));
  co.x = __int_as_float(__float_as_int(c1.x) ^ (((q.x + 1) & 2) << 30)); co.y = __int_as_float(__float_as_int(c1.y) ^ (((q.y + 1) & 2) << 30));
  *sp = Pk2f(so); *cp = Pk2f(co);
}
WBC_DEV Pk2f rsqrt_t(Pk2f x) { wbc_f2v r; r.x = 1.0f / sqrtf(x.v.x); r.y = 1.0f / sqrtf(x.v.y); return Pk2f(r); }
// a value passed through an empty asm: the compiler can neither move its computation across this point nor rematerialise it later
WBC_DEV void pin(double& x) { asm volatile("" : "+v"(x)); }
WBC_DEV void pin(float& x) { asm volatile("" : "+v"(x)); }
WBC_DEV void pin(Pk2f& x) { asm volatile("" : "+v"(x.v)); }
template <class X> WBC_DEV void pin(V3<X>& v) { pin(v.x); pin(v.y); pin(v.z); }

// MODE bits

constexpr int WBC_SWEEP_WAVES = 2;
// BLOCK = 64 for small batches (N/16 workgroups: one per CU at N = 4096) or 256 for large ones (four waves
// share one constant table, which lets two workgroups = 8 waves fit the CU's 160 KB of LDS).
// The observer variants may use more than 256 VGPRs (their occupancy is LDS-bound anyway); forcing two waves per
// SIMD there makes the compiler spill to scratch.
// (Measured and removed: (1) a "leg per wavefront" mapping for small batches -- wave w runs leg w in 16 lanes, cross-leg sums
// through LDS -- to spread the 117 store instructions of a state group over four SIMDs: bitwise-equal results, 8-10 %
// SLOWER at N = 1 024 ... 4 096, the sweep wave is bound by its dependent fp64 chain, not by store issue; (2) this body as
// wave 0 of a fused sweep+QP workgroup with the workspace in LDS: -7 % per tick with the observer off, nothing with it
// on; superseded by the role-split fused tick in fused_tick.hip.hpp, which does not use this body.)
// W = states per lane.  W = 2 (fp32 only, even N): a lane owns TWO consecutive states as one packed pair (Pk2f above), so
// every 16-lane row still moves whole 128-byte lines (16 lanes x 8 B) and the arithmetic is v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 -- a wavefront then carries 32 states and the batch needs half the wavefronts.  With one fp32 state per lane a
// row moved half a line per instruction and the kernel took as long as the fp64 one (round 2: 24.1 vs 24.7 us at 32 768 states).
// The workgroup's LDS (declared by the KERNEL and handed to the body, so that a kernel that runs this body as one of two roles can overlay
// it with the other role's: sweep_obs_kernel below).  Layout: see the comment in dyn_sweep_body.
// XR (tile_tick.hip.hpp): the body is one role wavefront of a big workgroup that SHARES one constant table among its roles and takes the roles' step-workspace
// words (tau_partial from the sweep, rhat from the observer) in LDS instead of through memory: RoleShare below; the object then holds no table of its own.
template <class T, int MODE, int BLOCK, int W, bool XR = false> struct SweepLds {
  using V = typename LaneT<T, W>::type;
  static constexpr bool OBS = (MODE & SW_OBS) != 0;
  T cst[XR ? 4 : CST_WORDS]; T kgain[OBS ? 36 : 2]; int zidx_s[64]; V park[2 * 8 + 6 + (OBS ? 2 : 0) + 9][BLOCK];
};
template <bool XR, class L, class T> WBC_DEV decltype(auto) role_cst(L& lds, T* shared) { if constexpr (XR) return shared; else return (lds.cst); }
// blk: the index of this workgroup among the workgroups that run this body (blockIdx.x in the stand-alone kernel)
template <class T, int MODE, int BLOCK, int W = 1, bool XR = false>
WBC_DEV void dyn_sweep_body(const DevModel<T>* __restrict__ model, const DevParams<T> prm, const SweepArgs<T>& a, SweepLds<T, MODE, BLOCK, W, XR>& lds, const unsigned blk,
                            const RoleShare<T> xr = RoleShare<T>()) {
  using V = typename LaneT<T, W>::type;   // what a lane computes with: T, or a packed pair of T
  constexpr bool MATS = (MODE & SW_MATS) != 0, STEP = (MODE & SW_STEP) != 0, OBS = (MODE & SW_OBS) != 0, FWD_B = (MODE & SW_NOB) == 0;
  // ONE LDS object with the constant table first: ds_read / ds_write reach base + 16-bit offset, and the compiler places
  // separate __shared__ arrays largest-first -- behind the 74 kB parking area of a 256-thread workgroup every table word
  // needed an address register of its own (150 v_or_b32 in the tick's sweep, and the registers to hold them)
  // Parked per lane ([word][lane], conflict-free): for joints 0 and 1 sin / cos of the joint angle (E is rebuilt from them in
  // the return sweep: 18 FMAs on table words; parking E itself took 9 words) and the body wrench; the base body's wrench; and
  // the inputs the END of the kernel needs -- base acceleration and base position -- which are loaded with everything else at
  // the top: a wavefront that loads them where it uses them exposes a whole memory latency each time (two wavefronts per SIMD
  // do not hide it), measured 251-264 -> 224-226 us at N = 262 144.
  constexpr int PW = 8;                    // parked words per joint: sin, cos + body wrench 6
  constexpr int PB = 6;                    // parked words of the base body's own wrench
  constexpr int PE2 = OBS ? 2 : 0;         // observer: sin, cos of joint 2 too (the momentum pass walks the leg again)
  constexpr int PX = 9;                    // vdot_des base rows 6, base position 3
  static_assert(sizeof(lds.park) == sizeof(V) * (2 * PW + PB + PE2 + PX) * BLOCK, "SweepLds::park is sized for this layout");
  static_assert(!XR || (BLOCK == 64 && MATS && STEP && !OBS && !FWD_B), "a shared role is one observer-free sweep wavefront of a tile tick");
  decltype(auto) cst = role_cst<XR>(lds, xr.cst);
  int (&zidx_s)[64] = lds.zidx_s;
  T (&kgain)[OBS ? 36 : 2] = lds.kgain;
  V (&park)[2 * PW + PB + PE2 + PX][BLOCK] = lds.park;
#ifdef WBC_SWEEP_STAMP  // diagnostic build only: cycle stamps per phase, written over the pf output
  long long stp[13];
  int stn = 0;
#define SSTAMP() do { stp[stn++] = __builtin_readcyclecounter(); } while (0)
#else
#define SSTAMP() do {} while (0)
#endif
  SSTAMP();  // 0: kernel entry

  const size_t N = a.N;
  const unsigned N32 = (unsigned)N;
  // lane = 16*leg + (state within the wave): each 16-lane row owns one leg of 16 consecutive states
  const unsigned tix = threadIdx.x & (unsigned)(BLOCK - 1);   // thread within the BLOCK threads that run this body (a kernel may run several bodies side by side: tile_tick.hip.hpp)
  const int leg = (int)((tix & 63) >> 4);
  const size_t s_raw = (((size_t)blk * (BLOCK / 64) + (tix >> 6)) * 16 + (tix & 15)) * W;   // first state of this lane
  const bool live = s_raw < N;                            // (W = 2: N is even, so both states of a lane are in range together)
  // Dead lanes (beyond the batch) recompute the last state(s) and STORE what they computed: bit-identical duplicates of the live
  // lane's values at the same addresses.  Guarding every store with `if (live)` made each of the ~100 stores its own exec region
  // (s_and_saveexec / s_cbranch_execz / s_or: ~45 cycles each for a lone wavefront, tools/issue_probe.hip).  Only the observer state,
  // which is read-modified-written, keeps the guard (STVG): an all-dead wavefront could read it after the live one wrote it.
  // (The workspace words that DEPEND on that state -- b = w_des - rhat, tau_partial - rhat -- are stored unguarded too; that is only
  // sound while an observer variant's workgroup is ONE wavefront: static_assert in dyn_sweep_kernel.)
  const unsigned s32 = (unsigned)(live ? s_raw : N - W);
  const unsigned legN = (unsigned)leg * N32;
#define CS(i) cst[(i) * 4 + leg]
  // Addressing: every array is < 4 GiB (max_batch is capped at create), so a component row is reached as
  // (uniform 64-bit base in SGPRs) + (32-bit per-lane byte offset): no 64-bit vector address arithmetic.
  //   LDU/STU: component index is wave-uniform;  LDV/STV: component index differs per lane.
#define LDU(ptr, comp) (*(const V*)((const char*)((ptr) + (size_t)(comp) * N) + (size_t)(s32 * (unsigned)sizeof(T))))
#define LDV(ptr, comp) (*(const V*)((const char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))))
#define STU(ptr, comp, val) do { *(V*)((char*)((ptr) + (size_t)(comp) * N) + (size_t)(s32 * (unsigned)sizeof(T))) = (val); } while (0)
#define STV(ptr, comp, val) do { *(V*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))) = (val); } while (0)
#define STVG(ptr, comp, val) do { if (live) *(V*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))) = (val); } while (0)   /* in/out state (observer): dead lanes of OTHER wavefronts would race with the live one */
  // leg-strided component: comp = c0 + stride*leg (+ per-lane extra element offset xN = x*N)
#define STL(ptr, c0, stride, val) do { *(V*)((char*)((ptr) + (size_t)(c0) * N) + (size_t)(((unsigned)(stride) * legN + s32) * (unsigned)sizeof(T))) = (val); } while (0)
#define STLX(ptr, c0, stride, xN, val) do { *(V*)((char*)((ptr) + (size_t)(c0) * N) + (size_t)(((unsigned)(stride) * legN + (xN) + s32) * (unsigned)sizeof(T))) = (val); } while (0)
  // four base-replicated values, one per lane of the quad
#define ST4(ptr, c0, v0_, c1, v1_, c2, v2_, c3, v3_) STV(ptr, sel4<int>(leg, c0, c1, c2, c3), sel4<V>(leg, v0_, v1_, v2_, v3_))
#define ST4G(ptr, c0, v0_, c1, v1_, c2, v2_, c3, v3_) STVG(ptr, sel4<int>(leg, c0, c1, c2, c3), sel4<V>(leg, v0_, v1_, v2_, v3_))
  const int hcol = xr.col0 + (int)(tix & 15) * W;   // (XR) my first state's column of the tile
  (void)hcol;
#define WSTV(comp, val) do { if constexpr (XR) *(V*)(xr.hand + ((comp) - WS_TAUP + HAND_TAUP) * xr.hs + hcol) = (val); else STV(a.ws, comp, val); } while (0)   /* step workspace */
#define WST4(c0, v0_, c1, v1_, c2, v2_, c3, v3_) WSTV(sel4<int>(leg, c0, c1, c2, c3), sel4<V>(leg, v0_, v1_, v2_, v3_))
#define WSTL(c0, stride, val) WSTV((c0) + (stride) * leg, val)

  // ------------------------------------------------------------------ loads
  V qb[7], vb[6];
#pragma unroll
  for (int c = 0; c < 7; ++c) qb[c] = LDU(a.q, c);
#pragma unroll
  for (int c = 0; c < 6; ++c) vb[c] = LDU(a.v, c);
  int jx[3];
  unsigned jxN[3];
  jidx_of_leg(model, a.jpack, leg, jx);
#pragma unroll
  for (int k = 0; k < 3; ++k) jxN[k] = (unsigned)jx[k] * N32;
  V ql[3], vl[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    ql[k] = *(const V*)((const char*)(a.q + (size_t)7 * N) + (size_t)((jxN[k] + s32) * (unsigned)sizeof(T)));
    vl[k] = *(const V*)((const char*)(a.v + (size_t)6 * N) + (size_t)((jxN[k] + s32) * (unsigned)sizeof(T)));
  }
  // desired accelerations: needed in and after the return sweep, requested here (see the parking comment above)
  V al[3] = {0, 0, 0}, ad_in[6] = {0, 0, 0, 0, 0, 0};
  if (STEP) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      al[k] = *(const V*)((const char*)(a.vdot_des + (size_t)6 * N) + (size_t)((jxN[k] + s32) * (unsigned)sizeof(T)));
#pragma unroll
    for (int c = 0; c < 6; ++c) ad_in[c] = LDU(a.vdot_des, c);
  }
  // observer variants: the inputs of the update are requested here as well.  These variants run one wavefront per SIMD with
  // registers to spare (the occupancy is LDS-bound), and at the batch sizes that still use them a wavefront is alone on its
  // SIMD: every load issued where it is used (eight places in the observer pass) exposed its whole latency.
  const bool obs_upd = OBS && STEP && prm.observer_order > 0;
  V ob_fp[3] = {0, 0, 0}, ob_rb[6] = {0, 0, 0, 0, 0, 0}, ob_igb[6] = {0, 0, 0, 0, 0, 0}, ob_rl[3] = {0, 0, 0}, ob_igl[3] = {0, 0, 0}, ob_tp[3] = {0, 0, 0};
  V ob_wd[6] = {0, 0, 0, 0, 0, 0};
  if (obs_upd) {
#pragma unroll
    for (int c = 0; c < 3; ++c) ob_fp[c] = LDV(a.f_prev, 3 * leg + c);
#pragma unroll
    for (int c = 0; c < 6; ++c) { ob_rb[c] = LDU(a.obs_r, c); ob_igb[c] = LDU(a.obs_integ, c); }
#pragma unroll
    for (int k = 0; k < 3; ++k) { ob_rl[k] = LDV(a.obs_r, 6 + jx[k]); ob_igl[k] = LDV(a.obs_integ, 6 + jx[k]); ob_tp[k] = LDV(a.tau_prev, jx[k]); }
  }
  if (OBS && STEP) {
#pragma unroll
    for (int c = 0; c < 6; ++c) ob_wd[c] = LDU(a.w_des, c);
  }

  SSTAMP();  // 1: state loads issued
  // the per-leg constant table is staged AFTER the state loads have been issued: one memory round trip, not two
  for (int i = XR ? (int)threadIdx.x : (int)tix; i < CST_WORDS; i += XR ? xr.stage_threads : BLOCK) cst[i] = model->cst[i];
  if (MATS && tix < 64) zidx_s[tix] = model->zidx[tix];
  // observer gains of the joint rows are indexed by a run-time joint number: from LDS (a dynamic index into the
  // kernel-argument struct can end up as a private copy of the whole struct)
  if (OBS && tix == 0) {
#pragma unroll
    for (int i = 0; i < 18; ++i) { kgain[i] = prm.K1[i]; kgain[18 + i] = prm.K2[i]; }  // static indices only
  }
  __syncthreads();
  SSTAMP();  // 2: table staged (barrier passed)

  // observer off: the QP target wrench is just w_des.  The two-kernel ticks run the SW_NOB variants: their QP kernels read the
  // caller's w_des themselves (QpArgs::wdes).  Forwarding it here -- six loads and two stores at the top of the kernel, the
  // stores waiting for the loads -- held back every store behind them: 234 -> 177 us for this kernel at N = 262 144 (a
  // run-time test instead of a variant keeps most of the damage: 214 us).
  if (STEP && !OBS && FWD_B) {
    V b[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) b[c] = LDU(a.w_des, c);
    WST4(WS_B + 0, b[0], WS_B + 1, b[1], WS_B + 2, b[2], WS_B + 3, b[3]);
    if (leg < 2) WSTV(WS_B + 4 + leg, leg == 0 ? b[4] : b[5]);
  }

  // ------------------------------------------------------------------ data-independent stores first:
  // structural zeros and ones of M and Jc go out while the sweeps compute (they overlap the VALU work).
  if (MATS && !a.skip_consts) {
    const V Z = (T)0;
    if ((N & (2 * W - 1)) == 0) {
      // 16 bytes per lane (W = 1; 2 x 8 bytes with W = 2): neighbouring lanes (l, l+1) of a row pair up, the even one takes the
      // even-numbered constants and the odd one the odd-numbered ones, each for the states of BOTH lanes -- half the store
      // instructions for the same bytes (a wave store instruction costs ~90 cycles to issue whatever its width)
      struct alignas(2 * sizeof(V)) T2 { V a, b; };
      const unsigned odd = (s32 / W) & 1u, s2 = s32 & ~(unsigned)(2 * W - 1);
      // (N is a multiple of 2 W here: the states of a lane pair are in range together, so the pair store needs no guard of its own)
#define ST2C(ptr, comp, val) do { *(T2*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s2) * (unsigned)sizeof(T))) = T2{(val), (val)}; } while (0)
      for (int e = 2 * leg + (int)odd; e < 64; e += 8) {
        const int zi = zidx_s[e];
        if (zi >= 0) ST2C(a.M, zi, Z);
      }
      // Jc rows of my foot: 16 constant entries per row (identity block, skew diagonal, 12 joint columns), 48 in all
#pragma unroll
      for (int mrow = 0; mrow < 3; ++mrow) {
        const int rb = 54 * leg + 18 * mrow;
        // pairs (even entry, odd entry): (0,1) (2, 3+mrow) (6,7) (8,9) (10,11) (12,13) (14,15) (16,17)
        ST2C(a.Jc, rb + (odd ? 1 : 0), ((odd ? 1 : 0) == mrow) ? (T)1 : Z);
        ST2C(a.Jc, rb + (odd ? 3 + mrow : 2), (!odd && mrow == 2) ? (T)1 : Z);
        // joint columns: zeros, EXCEPT the three columns of my own leg, which the return sweep writes with data
        // (zero-filling them first cost 9 words/leg = 7 % of the kernel's store bytes, PMC WRITE_SIZE)
#pragma unroll
        for (int c = 6; c < 18; c += 2) {
          const int col = c + (int)odd - 6;
          if (col != jx[0] && col != jx[1] && col != jx[2]) ST2C(a.Jc, rb + c + (int)odd, Z);
        }
      }
#undef ST2C
    } else {
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
          if (c != jx[0] && c != jx[1] && c != jx[2]) STL(a.Jc, 18 * mrow + 6 + c, 54, Z);
      }
    }
  }
  SSTAMP();  // 3: early stores issued
  // unit quaternion kept (4 words); R is rebuilt after the sweeps instead of living through them (9 words)
  V qx, qy, qz, qw;
  {
    const V n = rsqrt_t(qb[3] * qb[3] + qb[4] * qb[4] + qb[5] * qb[5] + qb[6] * qb[6]);
    qx = qb[3] * n; qy = qb[4] * n; qz = qb[5] * n; qw = qb[6] * n;
  }
#define MAKE_R(R_) do { const V x = qx, y = qy, z = qz, w = qw; \
    R_.a[0] = 1 - 2 * (y * y + z * z); R_.a[1] = 2 * (x * y - z * w);     R_.a[2] = 2 * (x * z + y * w); \
    R_.a[3] = 2 * (x * y + z * w);     R_.a[4] = 1 - 2 * (x * x + z * z); R_.a[5] = 2 * (y * z - x * w); \
    R_.a[6] = 2 * (x * z - y * w);     R_.a[7] = 2 * (y * z + x * w);     R_.a[8] = 1 - 2 * (x * x + y * y); } while (0)
  const V bm = model->base_m;
  const V3<V> bh = mk<V>(model->base_h[0], model->base_h[1], model->base_h[2]);
  S3<V> bI;
  bI.xx = model->base_Io[0]; bI.xy = model->base_Io[1]; bI.xz = model->base_Io[2];
  bI.yy = model->base_Io[3]; bI.yz = model->base_Io[4]; bI.zz = model->base_Io[5];
  // the base body's own wrench / momentum / weight are formed now, so that om0, v0, aL0 die after joint 0
  const int ln = (int)tix;
  V* const px = &park[2 * PW + PB + PE2][ln];
  {
#pragma unroll
    for (int c = 0; c < 6; ++c) px[BLOCK * c] = ad_in[c];
    px[BLOCK * 6] = qb[0]; px[BLOCK * 7] = qb[1]; px[BLOCK * 8] = qb[2];
  }
#define XSUM(arr, K) xrow_sum_k<V, K>(arr)
  V3<V> omp, vp, aAp, aLp;
  {
    M3<V> R;
    MAKE_R(R);
    SF<V> bw;
    const V3<V> om0 = tmul(R, mk<V>(vb[3], vb[4], vb[5]));
    const V3<V> v0 = tmul(R, mk<V>(vb[0], vb[1], vb[2]));
    const V3<V> gneg = tmul(R, mk<V>(-model->grav[0], -model->grav[1], -model->grav[2]));  // R^V (-g)
    const V3<V> aL0 = gneg - cross(om0, v0);  // bias pass: vdot = 0, gravity folded in
    const SF<V> Iv0 = inertia_mul(bm, bh, bI, om0, v0);
    const SF<V> Ia0 = inertia_mul(bm, bh, bI, mk<V>(0, 0, 0), aL0);
    bw.n = Ia0.n + cross(om0, Iv0.n) + cross(v0, Iv0.f);
    bw.f = Ia0.f + cross(om0, Iv0.f);
    V* pb = &park[2 * PW][ln];
    pb[0] = bw.n.x; pb[BLOCK] = bw.n.y; pb[BLOCK * 2] = bw.n.z; pb[BLOCK * 3] = bw.f.x; pb[BLOCK * 4] = bw.f.y; pb[BLOCK * 5] = bw.f.z;
    omp = om0; vp = v0; aAp = mk<V>(0, 0, 0); aLp = aL0;
  }

  SSTAMP();  // 4: base quantities done (state loads have arrived)
  // ------------------------------------------------------------------ forward sweep down the leg
  // Per-joint results that the return sweep needs (E, body wrench) are PARKED IN LDS for joints 0 and 1
  // ([word][lane]: conflict-free), which keeps the kernel at two waves per SIMD without scratch; joint 2's stay in
  // registers.  The observer's momentum / gravity recursions run as a SECOND pass over the leg after the main
  // outputs are stored (their registers are free by then), re-using the parked E matrices.
  M3<V> E2;
  SF<V> f2;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int o = JOINT_WORDS * k;
    V sn, cs;
    sincos_t(ql[k], &sn, &cs);
    M3<V> E;
#pragma unroll
    for (int e = 0; e < 9; ++e) E.a[e] = CS(o + e) + cs * CS(o + 9 + e) + sn * CS(o + 18 + e);
    const V3<V> r = mk<V>(CS(o + 27), CS(o + 28), CS(o + 29));
    const V3<V> ax = mk<V>(CS(o + 30), CS(o + 31), CS(o + 32));
    const V m = CS(o + 33);
    const V3<V> h = mk<V>(CS(o + 34), CS(o + 35), CS(o + 36));
    S3<V> Io;
    Io.xx = CS(o + 37); Io.xy = CS(o + 38); Io.xz = CS(o + 39); Io.yy = CS(o + 40); Io.yz = CS(o + 41); Io.zz = CS(o + 42);
    const V qd = vl[k];
    const V3<V> om = tmul(E, omp) + ax * qd;
    const V3<V> vv = tmul(E, vp + cross(omp, r));
    const V3<V> aA = tmul(E, aAp) + cross(om, ax) * qd;
    const V3<V> aL = tmul(E, aLp + cross(aAp, r)) + cross(vv, ax) * qd;
    const SF<V> Iv = inertia_mul(m, h, Io, om, vv);
    const SF<V> Ia = inertia_mul(m, h, Io, aA, aL);
    SF<V> fk;
    fk.n = Ia.n + cross(om, Iv.n) + cross(vv, Iv.f);
    fk.f = Ia.f + cross(om, Iv.f);
    if (k < 2) {
      V* pk = &park[PW * k][ln];
      pk[0] = sn; pk[BLOCK] = cs;
      pk[BLOCK * 2] = fk.n.x; pk[BLOCK * 3] = fk.n.y; pk[BLOCK * 4] = fk.n.z;
      pk[BLOCK * 5] = fk.f.x; pk[BLOCK * 6] = fk.f.y; pk[BLOCK * 7] = fk.f.z;
    } else {
      E2 = E; f2 = fk;
      if (OBS) {
        V* pe = &park[2 * PW + PB][ln];
        pe[0] = sn; pe[BLOCK] = cs;
      }
    }
    omp = om; vp = vv; aAp = aA; aLp = aL;
  }

  SSTAMP();  // 5: forward sweep done
  // ------------------------------------------------------------------ return sweep up the leg
  V taup[3] = {0, 0, 0};  // (M vdot_des) joint rows of this leg, accumulated as M entries appear
  V cm; V3<V> ch; S3<V> cI;  // composite inertia of the subtree rooted at joint k, in frame k
  V3<V> dft = mk<V>(CS(129), CS(130), CS(131));  // foot relative to the current frame origin
  V3<V> jc[3];                                   // foot Jacobian columns, rotated progressively towards the base
  SF<V> Fp[3];                                   // CRBA force columns of joints >= k, carried frame by frame
  SF<V> facc;                                    // children's wrench in the current frame
#pragma unroll
  for (int k = 2; k >= 0; --k) {
    const int o = JOINT_WORDS * k;
    const V3<V> r = mk<V>(CS(o + 27), CS(o + 28), CS(o + 29));
    const V3<V> ax = mk<V>(CS(o + 30), CS(o + 31), CS(o + 32));
    const V m = CS(o + 33);
    const V3<V> h = mk<V>(CS(o + 34), CS(o + 35), CS(o + 36));
    S3<V> Io;
    Io.xx = CS(o + 37); Io.xy = CS(o + 38); Io.xz = CS(o + 39); Io.yy = CS(o + 40); Io.yz = CS(o + 41); Io.zz = CS(o + 42);
    M3<V> E;
    SF<V> fk;
    if (k == 2) {
      E = E2; fk = f2;
    } else {
      const V* pk = &park[PW * k][ln];
      const V sn = pk[0], cs = pk[BLOCK];
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        E.a[e] = CS(o + e) + cs * CS(o + 9 + e) + sn * CS(o + 18 + e);
        asm volatile("" : "+v"(E.a[e]));   // one entry at a time: 27 table words in flight at once would not fit the return sweep's registers
      }
      fk.n = mk<V>(pk[BLOCK * 2], pk[BLOCK * 3], pk[BLOCK * 4]) + facc.n;
      fk.f = mk<V>(pk[BLOCK * 5], pk[BLOCK * 6], pk[BLOCK * 7]) + facc.f;
    }
    // RNEA projection on the joint axis
    {
      const V hk = dot(ax, fk.n);
      if (MATS) STLX(a.h, 6, 0, jxN[k], hk);
      if (STEP) taup[k] += hk;
    }
    // CRBA: close the composite of joint k
    if (k == 2) { cm = m; ch = h; cI = Io; }
    else {
      cm += m; ch = ch + h;
      cI.xx += Io.xx; cI.xy += Io.xy; cI.xz += Io.xz; cI.yy += Io.yy; cI.yz += Io.yz; cI.zz += Io.zz;
    }
    Fp[k].n = mul(cI, ax);
    Fp[k].f = cross(ax, ch);
#pragma unroll
    for (int j = k; j < 3; ++j) {  // M[k][j] for this leg: columns j >= k are all in frame k now
      const V mkj = dot(ax, Fp[j].n);
      if (MATS) {
        int i = 6 + jx[k], jj = 6 + jx[j];
        if (i > jj) { const int t = i; i = jj; jj = t; }
        STV(a.M, i * 18 - i * (i - 1) / 2 + (jj - i), mkj);
      }
      if (STEP) { taup[k] += mkj * al[j]; if (j != k) taup[j] += mkj * al[k]; }
    }
    // Jacobian column of joint k in frame k
    jc[k] = cross(ax, dft);
    // move everything to the parent frame (frame k-1, or the base for k = 0)
    dft = r + mul(E, dft);
#pragma unroll
    for (int j = k; j < 3; ++j) { jc[j] = mul(E, jc[j]); Fp[j] = to_parent(E, r, Fp[j]); }
    facc = to_parent(E, r, fk);
    {
      const V3<V> hr = mul(E, ch);
      const S3<V> Ir = congr(E, cI);
      const V3<V> w = hr + r * (cm * (T)0.5);
      const V sc = 2 * dot(w, r);
      cI.xx = Ir.xx + sc - 2 * w.x * r.x;
      cI.yy = Ir.yy + sc - 2 * w.y * r.y;
      cI.zz = Ir.zz + sc - 2 * w.z * r.z;
      cI.xy = Ir.xy - (w.x * r.y + r.x * w.y);
      cI.xz = Ir.xz - (w.x * r.z + r.x * w.z);
      cI.yz = Ir.yz - (w.y * r.z + r.y * w.z);
      ch = hr + r * cm;
    }
  }
  SSTAMP();  // 6: return sweep done
  // now: facc (macc, gacc) = leg wrench at the base, base coords; (cm,ch,cI) = leg composite in base
  // coords; dft = foot relative to base origin in base coords; jc[], Fp[] in base coords.
  M3<V> R;
  MAKE_R(R);
  const V3<V> dw = mul(R, dft);
  V3<V> jw[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) jw[k] = mul(R, jc[k]);
  {
    V ad[6] = {0, 0, 0, 0, 0, 0};
    if (STEP) {
#pragma unroll
      for (int c = 0; c < 6; ++c) ad[c] = px[BLOCK * c];   // parked at the top of the kernel
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const V3<V> Mf = mul(R, Fp[k].f), Mn = mul(R, Fp[k].n);  // base-leg column, world axes
      if (MATS) {
        const unsigned x = jxN[k];
        STLX(a.M, 6, 0, x, Mf.x);                       // midx18(0, 6+j) = 6 + j
        STLX(a.M, midx18(1, 1) + 5, 0, x, Mf.y);        // midx18(r, 6+j) = midx18(r,r) + 6 - r + j
        STLX(a.M, midx18(2, 2) + 4, 0, x, Mf.z);
        STLX(a.M, midx18(3, 3) + 3, 0, x, Mn.x);
        STLX(a.M, midx18(4, 4) + 2, 0, x, Mn.y);
        STLX(a.M, midx18(5, 5) + 1, 0, x, Mn.z);
      }
      if (STEP) taup[k] += Mf.x * ad[0] + Mf.y * ad[1] + Mf.z * ad[2] + Mn.x * ad[3] + Mn.y * ad[4] + Mn.z * ad[5];
    }
  }
  if (MATS) {
    // Jc rows of this lane's foot, (3*leg+m)*18 + c: base rotational block and own-leg columns
    STL(a.Jc, 0 * 18 + 4, 54, dw.z);  STL(a.Jc, 0 * 18 + 5, 54, -dw.y);
    STL(a.Jc, 1 * 18 + 3, 54, -dw.z); STL(a.Jc, 1 * 18 + 5, 54, dw.x);
    STL(a.Jc, 2 * 18 + 3, 54, dw.y);  STL(a.Jc, 2 * 18 + 4, 54, -dw.x);
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // overwrite the zeros written above (same lane, same address: program order)
      STLX(a.Jc, 0 * 18 + 6, 54, jxN[k], jw[k].x);
      STLX(a.Jc, 1 * 18 + 6, 54, jxN[k], jw[k].y);
      STLX(a.Jc, 2 * 18 + 6, 54, jxN[k], jw[k].z);
    }
  }
  if (a.pf) {   // base position: parked at the top of the kernel
    STL(a.pf, 0, 3, px[BLOCK * 6] + dw.x);
    STL(a.pf, 1, 3, px[BLOCK * 7] + dw.y);
    STL(a.pf, 2, 3, px[BLOCK * 8] + dw.z);
  }
  // the QP's geometry (foot lever arms, own-leg Jacobian blocks: 48 of the 66 workspace words) is a subset of Jc; when
  // Jc is being written anyway the QP kernel reads it from there and these stores are skipped (-9 % store bytes)
  if (!XR && STEP && (!MATS || a.ws_geom)) {
    WSTL(WS_D + 0, 3, dw.x);
    WSTL(WS_D + 1, 3, dw.y);
    WSTL(WS_D + 2, 3, dw.z);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      WSTL(WS_JCL + 0 + k, 9, jw[k].x);
      WSTL(WS_JCL + 3 + k, 9, jw[k].y);
      WSTL(WS_JCL + 6 + k, 9, jw[k].z);
    }
  }

  SSTAMP();  // 7: leg outputs stored
  // ------------------------------------------------------------------ quad reductions into the base
  if (MATS) {
    const V* pb = &park[2 * PW][ln];
    V xa[16] = {facc.n.x, facc.n.y, facc.n.z, facc.f.x, facc.f.y, facc.f.z, cm, ch.x, ch.y, ch.z,
                cI.xx, cI.xy, cI.xz, cI.yy, cI.yz, cI.zz};
    XSUM(xa, 16);   // the four legs' contributions to the base: bias wrench 6, composite mass 1, first moment 3, inertia 6
    const V3<V> bfn = mk<V>(xa[0], xa[1], xa[2]) + mk<V>(pb[0], pb[BLOCK], pb[BLOCK * 2]);   // total bias wrench, base coords
    const V3<V> bff = mk<V>(xa[3], xa[4], xa[5]) + mk<V>(pb[BLOCK * 3], pb[BLOCK * 4], pb[BLOCK * 5]);
    const V3<V> hb_f = mul(R, bff), hb_n = mul(R, bfn);  // h base rows (force, moment), world
    ST4(a.h, 0, hb_f.x, 1, hb_f.y, 2, hb_f.z, 3, hb_n.x);
    if (leg < 2) STV(a.h, 4 + leg, leg == 0 ? hb_n.y : hb_n.z);
    const V tm = xa[6] + bm;
    const V3<V> th = mk<V>(xa[7], xa[8], xa[9]) + bh;
    S3<V> tI;
    tI.xx = xa[10] + bI.xx; tI.xy = xa[11] + bI.xy; tI.xz = xa[12] + bI.xz;
    tI.yy = xa[13] + bI.yy; tI.yz = xa[14] + bI.yz; tI.zz = xa[15] + bI.zz;
    const V3<V> hw = mul(R, th);
    const S3<V> Iw = congr(R, tI);
    // base 6x6 block: 15 data-dependent entries (the 6 structural zeros went out with the early stores)
    T* M = a.M;
    ST4(M, midx18(0, 0), tm, midx18(1, 1), tm, midx18(2, 2), tm, midx18(0, 4), hw.z);
    ST4(M, midx18(0, 5), -hw.y, midx18(1, 3), -hw.z, midx18(1, 5), hw.x, midx18(2, 3), hw.y);
    ST4(M, midx18(2, 4), -hw.x, midx18(3, 3), Iw.xx, midx18(3, 4), Iw.xy, midx18(3, 5), Iw.xz);
    if (leg < 3) STV(M, sel4<int>(leg, midx18(4, 4), midx18(4, 5), midx18(5, 5), 0), sel4<V>(leg, Iw.yy, Iw.yz, Iw.zz, Iw.zz));
  }
  // ------------------------------------------------------------------ observer: second pass over the leg
  // body momenta I v, gravity-only forces and their leaf->root accumulation (a5): velocities are re-propagated
  // with the parked E matrices (cheap), so nothing of this lived in registers during the first pass.
  SF<V> mom0, grv0;
  V p_leg[3], ct_leg[3], g_leg[3];
  if (OBS) {
    // keep the compiler from hoisting this pass's loads into the first pass (that is what made the one-pass form spill)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    V3<V> omk[3], vvk[3], glk[3];
    SF<V> ivk[3];
    V3<V> om0, v0, gneg;
    {
      om0 = tmul(R, mk<V>(vb[3], vb[4], vb[5]));   // (vb, vl: kept from the top of the kernel in these variants)
      v0 = tmul(R, mk<V>(vb[0], vb[1], vb[2]));
      gneg = tmul(R, mk<V>(-model->grav[0], -model->grav[1], -model->grav[2]));
      V3<V> omp2 = om0, vp2 = v0, gp2 = gneg;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int o = JOINT_WORDS * k;
        const V* pe = (k < 2) ? &park[PW * k][ln] : &park[2 * PW + PB][ln];
        M3<V> Ek;
        {
          const V sn = pe[0], cs = pe[BLOCK];
#pragma unroll
          for (int e = 0; e < 9; ++e) Ek.a[e] = CS(o + e) + cs * CS(o + 9 + e) + sn * CS(o + 18 + e);
        }
        const V3<V> r = mk<V>(CS(o + 27), CS(o + 28), CS(o + 29));
        const V3<V> ax = mk<V>(CS(o + 30), CS(o + 31), CS(o + 32));
        S3<V> Io;
        Io.xx = CS(o + 37); Io.xy = CS(o + 38); Io.xz = CS(o + 39); Io.yy = CS(o + 40); Io.yz = CS(o + 41); Io.zz = CS(o + 42);
        const V qd = vl[k];
        omk[k] = tmul(Ek, omp2) + ax * qd;
        vvk[k] = tmul(Ek, vp2 + cross(omp2, r));
        glk[k] = tmul(Ek, gp2);
        ivk[k] = inertia_mul(V(CS(o + 33)), mk<V>(CS(o + 34), CS(o + 35), CS(o + 36)), Io, omk[k], vvk[k]);
        omp2 = omk[k]; vp2 = vvk[k]; gp2 = glk[k];
      }
    }
    SF<V> macc, gacc;
#pragma unroll
    for (int k = 2; k >= 0; --k) {
      const int o = JOINT_WORDS * k;
      const V3<V> r = mk<V>(CS(o + 27), CS(o + 28), CS(o + 29));
      const V3<V> ax = mk<V>(CS(o + 30), CS(o + 31), CS(o + 32));
      const V m = CS(o + 33);
      const V3<V> h = mk<V>(CS(o + 34), CS(o + 35), CS(o + 36));
      SF<V> mk_, gk;
      mk_ = ivk[k];
      gk.n = cross(h, glk[k]);
      gk.f = glk[k] * m;
      if (k < 2) { mk_.n = mk_.n + macc.n; mk_.f = mk_.f + macc.f; gk.n = gk.n + gacc.n; gk.f = gk.f + gacc.f; }
      p_leg[k] = dot(ax, mk_.n);
      ct_leg[k] = -dot(ax, cross(omk[k], mk_.n) + cross(vvk[k], mk_.f));
      g_leg[k] = dot(ax, gk.n);
      {
        const V* pe = (k < 2) ? &park[PW * k][ln] : &park[2 * PW + PB][ln];  // E again, rebuilt rather than held in 27 registers
        M3<V> Ek;
        {
          const V sn = pe[0], cs = pe[BLOCK];
#pragma unroll
          for (int e = 0; e < 9; ++e) Ek.a[e] = CS(o + e) + cs * CS(o + 9 + e) + sn * CS(o + 18 + e);
        }
        macc = to_parent(Ek, r, mk_);
        gacc = to_parent(Ek, r, gk);
      }
    }
    const SF<V> Iv0 = inertia_mul(bm, bh, bI, om0, v0);
    V xb[12] = {macc.n.x, macc.n.y, macc.n.z, macc.f.x, macc.f.y, macc.f.z, gacc.n.x, gacc.n.y, gacc.n.z, gacc.f.x, gacc.f.y, gacc.f.z};
    XSUM(xb, 12);
    mom0.n = mk<V>(xb[0], xb[1], xb[2]) + Iv0.n;
    mom0.f = mk<V>(xb[3], xb[4], xb[5]) + Iv0.f;
    grv0.n = mk<V>(xb[6], xb[7], xb[8]) + cross(bh, gneg);
    grv0.f = mk<V>(xb[9], xb[10], xb[11]) + gneg * bm;
  }

  SSTAMP();  // 8: base block stored
  // ------------------------------------------------------------------ momentum, beta = C^T v - g
  V p_b[6], beta_b[6], beta_l[3];
  if (OBS) {
    const V3<V> Pl = mul(R, mom0.f), Pa = mul(R, mom0.n);
    const V3<V> gl = mul(R, grv0.f), ga = mul(R, grv0.n);
    const V3<V> cx = cross(mk<V>(vb[0], vb[1], vb[2]), Pl);
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

  // ------------------------------------------------------------------ step-mode prologue for the QP
  if (STEP) {
    V rb[6] = {0, 0, 0, 0, 0, 0}, rl[3] = {0, 0, 0};
    if (OBS && prm.observer_order > 0) {
      // generalized force of the previous commands at the current configuration
      const V3<V> fp = mk<V>(ob_fp[0], ob_fp[1], ob_fp[2]);
      const V3<V> dxf = cross(dw, fp);
      V ub[6] = {fp.x, fp.y, fp.z, dxf.x, dxf.y, dxf.z};
      XSUM(ub, 6);
      const V dt = prm.dt;
      const bool o1 = prm.observer_order == 1;
#pragma unroll
      for (int c = 0; c < 6; ++c) {  // replicated over the quad (same values in all four lanes)
        const V r0 = ob_rb[c];
        const V ig = ob_igb[c] + dt * (ub[c] + beta_b[c] + r0);
        const V e = p_b[c] - ig;
        rb[c] = o1 ? kgain[c] * e : r0 + dt * kgain[18 + c] * (kgain[c] * e - r0);   // (from LDS: 72 SGPRs of gains held to the end of the kernel meant SGPR spills read back 800 times)
        p_b[c] = ig;  // reuse as the new integ for the store below
      }
      // (the replicated rows were read by every lane of the wave at the top of the kernel, long before any lane stores to them)
      ST4G(a.obs_integ, 0, p_b[0], 1, p_b[1], 2, p_b[2], 3, p_b[3]);
      if (leg < 2) STVG(a.obs_integ, 4 + leg, leg == 0 ? p_b[4] : p_b[5]);
      ST4G(a.obs_r, 0, rb[0], 1, rb[1], 2, rb[2], 3, rb[3]);
      if (leg < 2) STVG(a.obs_r, 4 + leg, leg == 0 ? rb[4] : rb[5]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = 6 + jx[k];
        const V r0 = ob_rl[k];
        const V u = ob_tp[k] + dot(jw[k], fp);
        const V ig = ob_igl[k] + dt * (u + beta_l[k] + r0);
        const V e = p_leg[k] - ig;
        rl[k] = o1 ? kgain[c] * e : r0 + dt * kgain[18 + c] * (kgain[c] * e - r0);
        STVG(a.obs_integ, c, ig);
        STVG(a.obs_r, c, rl[k]);
      }
    }
    if (OBS) {
      V b[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) b[c] = ob_wd[c] - rb[c];
      WST4(WS_B + 0, b[0], WS_B + 1, b[1], WS_B + 2, b[2], WS_B + 3, b[3]);
      if (leg < 2) WSTV(WS_B + 4 + leg, leg == 0 ? b[4] : b[5]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) WSTL(WS_TAUP + k, 3, taup[k] - rl[k]);
  }
#ifdef WBC_SWEEP_STAMP
  SSTAMP();  // 9: everything issued
  __builtin_amdgcn_s_waitcnt(0);  // drain: all stores acknowledged
  SSTAMP();  // 10
  if (a.pf && leg < 3)
    for (int m = 0; m < 3; ++m) STL(a.pf, m, 3, (T)(stp[3 * leg + m + 1] - stp[3 * leg + m]));
  if (a.pf && leg == 3) STL(a.pf, 0, 3, (T)(stp[10] - stp[9]));
#endif
#undef SSTAMP
#undef XSUM
#undef MAKE_R
#undef WSTL
#undef WST4
#undef WSTV
#undef ST4
#undef ST4G
#undef STVG
#undef STLX
#undef STL
#undef STV
#undef STU
#undef LDV
#undef LDU
#undef CS
}

template <class T, int MODE, int BLOCK, int W = 1>
__global__ __launch_bounds__(BLOCK, (MODE & SW_OBS) ? 1 : WBC_SWEEP_WAVES) void dyn_sweep_kernel(const DevModel<T>* __restrict__ model, DevParams<T> prm,
                                                        SweepArgs<T> a) {
  // The observer variants store the QP target wrench and tau_partial (which depend on the observer state they loaded) from dead
  // lanes unguarded: bit-identical duplicates ONLY while no all-dead wavefront can re-read {integ, r} after the live wavefront
  // of the same state updated them, i.e. while a workgroup is one wavefront.
  static_assert(!((MODE & SW_OBS) && BLOCK > 64), "observer variants of the sweep are launched as 64-thread workgroups only");
  if (a.qp_todo && blockIdx.x == 0 && threadIdx.x == 0) a.qp_todo[0] = 0;
  __shared__ SweepLds<T, MODE, BLOCK, W> lds;
  dyn_sweep_body<T, MODE, BLOCK, W>(model, prm, a, lds, blockIdx.x);
}

}  // namespace wbc

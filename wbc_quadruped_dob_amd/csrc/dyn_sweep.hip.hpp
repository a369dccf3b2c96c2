// Dynamics-sweep kernel for gfx950 (CDNA4): SURVEY.md 8(a) units a1-a7/a9-prologue for one batch.
//
// Mapping: ONE LANE PER LEG.  A quadruped's four legs are independent subtrees of the floating
// base, so lanes 4s..4s+3 of a wave own the four legs of state s (16 states per 64-wide wave).
// Everything leg-local (joint transforms, RNEA/CRBA/momentum sweeps up and down the 3-joint
// chain, the leg's Jacobian block) runs without communication; the only cross-lane traffic is
// the leaf->root accumulation into the base (leg wrench, leg momentum, leg composite inertia),
// done with DPP quad_perm adds -- no LDS, no barriers.  Per-leg model constants are staged once
// per block in LDS as cst[word][leg] (four consecutive words per row: conflict-free).
// HBM layout is component-major x[c*N+s]: every load/store instruction of a wave touches four
// fully used 128-byte lines (16 consecutive states x 8 B per leg-specific component), base
// components are distributed over the quad with v_cndmask so no lane issues a redundant store.
//
// Algorithms (textbook; the reference's own dynamics library is in an absent submodule):
// Featherstone RNEA / CRBA in link coordinates with (m, h, Io) rigid-body inertias, mixed
// ("world-aligned at the base origin") generalized velocity -- DESIGN.md section 3.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace wbc {

#define WBC_DEV __device__ __forceinline__

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
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL> WBC_DEV double dpp_mov(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <class T> WBC_DEV T quad_sum(T x) {
  x += dpp_mov<0xB1>(x);  // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E>(x);  // quad_perm [2,3,0,1]
  return x;
}
template <class T> WBC_DEV V3<T> quad_sum(V3<T> v) { return mk<T>(quad_sum(v.x), quad_sum(v.y), quad_sum(v.z)); }

template <class T> WBC_DEV T sel4(int leg, T a, T b, T c, T d) { return leg == 0 ? a : (leg == 1 ? b : (leg == 2 ? c : d)); }

WBC_DEV void sincos_t(double x, double* s, double* c) { sincos(x, s, c); }
WBC_DEV void sincos_t(float x, float* s, float* c) { sincosf(x, s, c); }
WBC_DEV double rsqrt_t(double x) { return 1.0 / sqrt(x); }
WBC_DEV float rsqrt_t(float x) { return 1.0f / sqrtf(x); }

__host__ __device__ constexpr int midx18(int i, int j) { return i * 18 - i * (i - 1) / 2 + (j - i); }

// MODE bits
constexpr int SW_MATS = 1;  // write M, h, Jc
constexpr int SW_STEP = 2;  // write the step workspace (d, b, taup, JcL)
constexpr int SW_OBS = 4;   // momentum observer update (needs SW_STEP) / p, beta outputs

template <class T, int MODE>
__global__ __launch_bounds__(64) void dyn_sweep_kernel(const DevModel<T>* __restrict__ model, DevParams<T> prm,
                                                        SweepArgs<T> a) {
  constexpr bool MATS = (MODE & SW_MATS) != 0, STEP = (MODE & SW_STEP) != 0, OBS = (MODE & SW_OBS) != 0;
  __shared__ T cst[CST_WORDS];
  for (int i = threadIdx.x; i < CST_WORDS; i += blockDim.x) cst[i] = model->cst[i];
  __syncthreads();

  const size_t N = a.N;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int leg = (int)(gid & 3);
  const size_t s_raw = gid >> 2;
  const bool live = s_raw < N;
  const size_t s = live ? s_raw : N - 1;  // dead lanes recompute the last state, stores are masked
#define CS(i) cst[(i) * 4 + leg]

  // ------------------------------------------------------------------ loads
  T qb[7], vb[6];
#pragma unroll
  for (int c = 0; c < 7; ++c) qb[c] = a.q[(size_t)c * N + s];
#pragma unroll
  for (int c = 0; c < 6; ++c) vb[c] = a.v[(size_t)c * N + s];
  int jx[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) jx[k] = model->jidx[leg][k];
  T ql[3], vl[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    ql[k] = a.q[(size_t)(7 + jx[k]) * N + s];
    vl[k] = a.v[(size_t)(6 + jx[k]) * N + s];
  }

  // ------------------------------------------------------------------ base
  M3<T> R;
  {
    T n = rsqrt_t(qb[3] * qb[3] + qb[4] * qb[4] + qb[5] * qb[5] + qb[6] * qb[6]);
    T x = qb[3] * n, y = qb[4] * n, z = qb[5] * n, w = qb[6] * n;
    R.a[0] = 1 - 2 * (y * y + z * z); R.a[1] = 2 * (x * y - z * w);     R.a[2] = 2 * (x * z + y * w);
    R.a[3] = 2 * (x * y + z * w);     R.a[4] = 1 - 2 * (x * x + z * z); R.a[5] = 2 * (y * z - x * w);
    R.a[6] = 2 * (x * z - y * w);     R.a[7] = 2 * (y * z + x * w);     R.a[8] = 1 - 2 * (x * x + y * y);
  }
  const V3<T> vlin_w = mk<T>(vb[0], vb[1], vb[2]);
  const V3<T> om0 = tmul(R, mk<T>(vb[3], vb[4], vb[5]));
  const V3<T> v0 = tmul(R, vlin_w);
  const V3<T> gneg = tmul(R, mk<T>(-model->grav[0], -model->grav[1], -model->grav[2]));  // R^T (-g)
  const V3<T> aL0 = gneg - cross(om0, v0);  // bias pass: vdot = 0, gravity folded in
  const T bm = model->base_m;
  const V3<T> bh = mk<T>(model->base_h[0], model->base_h[1], model->base_h[2]);
  S3<T> bI;
  bI.xx = model->base_Io[0]; bI.xy = model->base_Io[1]; bI.xz = model->base_Io[2];
  bI.yy = model->base_Io[3]; bI.yz = model->base_Io[4]; bI.zz = model->base_Io[5];

  // ------------------------------------------------------------------ forward sweep down the leg
  M3<T> E[3];
  V3<T> om[3], vv[3];
  SF<T> frc[3];   // RNEA body force  I a + v x* I v
  SF<T> mom[3];   // body momentum    I v                 (OBS)
  SF<T> grv[3];   // gravity-only body force              (OBS)
  {
    V3<T> omp = om0, vp = v0, aAp = mk<T>(0, 0, 0), aLp = aL0, gLp = gneg;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int o = JOINT_WORDS * k;
      T sn, cs;
      sincos_t(ql[k], &sn, &cs);
#pragma unroll
      for (int e = 0; e < 9; ++e) E[k].a[e] = CS(o + e) + cs * CS(o + 9 + e) + sn * CS(o + 18 + e);
      const V3<T> r = mk<T>(CS(o + 27), CS(o + 28), CS(o + 29));
      const V3<T> ax = mk<T>(CS(o + 30), CS(o + 31), CS(o + 32));
      const T m = CS(o + 33);
      const V3<T> h = mk<T>(CS(o + 34), CS(o + 35), CS(o + 36));
      S3<T> Io;
      Io.xx = CS(o + 37); Io.xy = CS(o + 38); Io.xz = CS(o + 39); Io.yy = CS(o + 40); Io.yz = CS(o + 41); Io.zz = CS(o + 42);
      const T qd = vl[k];
      om[k] = tmul(E[k], omp) + ax * qd;
      vv[k] = tmul(E[k], vp + cross(omp, r));
      const V3<T> aA = tmul(E[k], aAp) + cross(om[k], ax) * qd;
      const V3<T> aL = tmul(E[k], aLp + cross(aAp, r)) + cross(vv[k], ax) * qd;
      const SF<T> Iv = inertia_mul(m, h, Io, om[k], vv[k]);
      const SF<T> Ia = inertia_mul(m, h, Io, aA, aL);
      frc[k].n = Ia.n + cross(om[k], Iv.n) + cross(vv[k], Iv.f);
      frc[k].f = Ia.f + cross(om[k], Iv.f);
      if (OBS) {
        mom[k] = Iv;
        const V3<T> gL = tmul(E[k], gLp);  // angular part stays zero
        grv[k].n = cross(h, gL);
        grv[k].f = gL * m;
        gLp = gL;
      }
      omp = om[k]; vp = vv[k]; aAp = aA; aLp = aL;
    }
  }

  // ------------------------------------------------------------------ backward sweep up the leg
  T h_leg[3], p_leg[3], ct_leg[3], g_leg[3];
  T Mll[3][3];          // leg block (upper part used)
  V3<T> Mbl_f[3], Mbl_n[3];  // base-leg columns in BASE coordinates (force, moment)
  // composite inertia of the subtree rooted at joint k, in frame k
  T cm; V3<T> ch; S3<T> cI;
  // foot Jacobian columns, progressively rotated towards the base frame
  V3<T> dft = mk<T>(CS(129), CS(130), CS(131));
  V3<T> jc[3];
#pragma unroll
  for (int k = 2; k >= 0; --k) {
    const int o = JOINT_WORDS * k;
    const V3<T> r = mk<T>(CS(o + 27), CS(o + 28), CS(o + 29));
    const V3<T> ax = mk<T>(CS(o + 30), CS(o + 31), CS(o + 32));
    const T m = CS(o + 33);
    const V3<T> h = mk<T>(CS(o + 34), CS(o + 35), CS(o + 36));
    S3<T> Io;
    Io.xx = CS(o + 37); Io.xy = CS(o + 38); Io.xz = CS(o + 39); Io.yy = CS(o + 40); Io.yz = CS(o + 41); Io.zz = CS(o + 42);
    // RNEA
    h_leg[k] = dot(ax, frc[k].n);
    if (OBS) {
      p_leg[k] = dot(ax, mom[k].n);
      ct_leg[k] = -dot(ax, cross(om[k], mom[k].n) + cross(vv[k], mom[k].f));
      g_leg[k] = dot(ax, grv[k].n);
    }
    // CRBA: close the composite of joint k
    if (k == 2) { cm = m; ch = h; cI = Io; }
    else {
      cm += m; ch = ch + h;
      cI.xx += Io.xx; cI.xy += Io.xy; cI.xz += Io.xz; cI.yy += Io.yy; cI.yz += Io.yz; cI.zz += Io.zz;
    }
    SF<T> F;
    F.n = mul(cI, ax);
    F.f = cross(ax, ch);
    Mll[k][k] = dot(ax, F.n);
#pragma unroll
    for (int j = k; j >= 1; --j) {  // up the chain: frame j -> frame j-1
      const int oj = JOINT_WORDS * j;
      F = to_parent(E[j], mk<T>(CS(oj + 27), CS(oj + 28), CS(oj + 29)), F);
      Mll[j - 1][k] = dot(mk<T>(CS(oj - JOINT_WORDS + 30), CS(oj - JOINT_WORDS + 31), CS(oj - JOINT_WORDS + 32)), F.n);
    }
    F = to_parent(E[0], mk<T>(CS(27), CS(28), CS(29)), F);
    Mbl_f[k] = F.f; Mbl_n[k] = F.n;
    // Jacobian column of joint k in frame k, then move everything below into frame k-1 (or base)
    jc[k] = cross(ax, dft);
    dft = r + mul(E[k], dft);
#pragma unroll
    for (int j = k; j < 3; ++j) jc[j] = mul(E[k], jc[j]);
    // hand the accumulators to the parent frame
    {
      const SF<T> fp = to_parent(E[k], r, frc[k]);
      if (k > 0) { frc[k - 1].n = frc[k - 1].n + fp.n; frc[k - 1].f = frc[k - 1].f + fp.f; } else frc[0] = fp;
      if (OBS) {
        const SF<T> mp = to_parent(E[k], r, mom[k]);
        const SF<T> gp = to_parent(E[k], r, grv[k]);
        if (k > 0) {
          mom[k - 1].n = mom[k - 1].n + mp.n; mom[k - 1].f = mom[k - 1].f + mp.f;
          grv[k - 1].n = grv[k - 1].n + gp.n; grv[k - 1].f = grv[k - 1].f + gp.f;
        } else { mom[0] = mp; grv[0] = gp; }
      }
      // composite inertia into the parent frame
      const V3<T> hr = mul(E[k], ch);
      const S3<T> Ir = congr(E[k], cI);
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
  // now: frc[0] (and mom[0], grv[0]) = leg wrench at the base, base coords; (cm,ch,cI) = leg composite
  // in base coords; dft = foot relative to base origin in base coords; jc[] in base coords.

  // ------------------------------------------------------------------ quad reductions into the base
  SF<T> bf;  // total bias wrench on the base, base coords
  {
    const SF<T> Iv0 = inertia_mul(bm, bh, bI, om0, v0);
    const SF<T> Ia0 = inertia_mul(bm, bh, bI, mk<T>(0, 0, 0), aL0);
    bf.n = quad_sum(frc[0].n) + Ia0.n + cross(om0, Iv0.n) + cross(v0, Iv0.f);
    bf.f = quad_sum(frc[0].f) + Ia0.f + cross(om0, Iv0.f);
    if (OBS) {
      mom[0].n = quad_sum(mom[0].n) + Iv0.n;
      mom[0].f = quad_sum(mom[0].f) + Iv0.f;
      grv[0].n = quad_sum(grv[0].n) + cross(bh, gneg);
      grv[0].f = quad_sum(grv[0].f) + gneg * bm;
    }
  }
  const V3<T> hb_f = mul(R, bf.f), hb_n = mul(R, bf.n);  // h base rows (force, moment), world
  const T tm = quad_sum(cm) + bm;
  const V3<T> th = quad_sum(ch) + bh;
  S3<T> tI;
  tI.xx = quad_sum(cI.xx) + bI.xx; tI.xy = quad_sum(cI.xy) + bI.xy; tI.xz = quad_sum(cI.xz) + bI.xz;
  tI.yy = quad_sum(cI.yy) + bI.yy; tI.yz = quad_sum(cI.yz) + bI.yz; tI.zz = quad_sum(cI.zz) + bI.zz;
  const V3<T> hw = mul(R, th);
  const S3<T> Iw = congr(R, tI);
  // world-frame leg quantities
  const V3<T> dw = mul(R, dft);
  V3<T> jw[3], Mf[3], Mn[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { jw[k] = mul(R, jc[k]); Mf[k] = mul(R, Mbl_f[k]); Mn[k] = mul(R, Mbl_n[k]); }
  Mll[1][0] = Mll[0][1]; Mll[2][0] = Mll[0][2]; Mll[2][1] = Mll[1][2];

#define ST(ptr, comp, val) do { if (live) (ptr)[(size_t)(comp) * N + s] = (val); } while (0)
  // store four base-replicated values, one per lane of the quad
#define ST4(ptr, c0, v0_, c1, v1_, c2, v2_, c3, v3_) \
  ST(ptr, sel4<int>(leg, c0, c1, c2, c3), sel4<T>(leg, v0_, v1_, v2_, v3_))

  // ------------------------------------------------------------------ M, h, Jc, pf
  if (MATS) {
    T* M = a.M;
    // base 6x6 block, 21 unique entries (rows/cols 0..5)
    const T Z = (T)0;
    ST4(M, midx18(0, 0), tm, midx18(0, 1), Z, midx18(0, 2), Z, midx18(0, 3), Z);
    ST4(M, midx18(0, 4), hw.z, midx18(0, 5), -hw.y, midx18(1, 1), tm, midx18(1, 2), Z);
    ST4(M, midx18(1, 3), -hw.z, midx18(1, 4), Z, midx18(1, 5), hw.x, midx18(2, 2), tm);
    ST4(M, midx18(2, 3), hw.y, midx18(2, 4), -hw.x, midx18(2, 5), Z, midx18(3, 3), Iw.xx);
    ST4(M, midx18(3, 4), Iw.xy, midx18(3, 5), Iw.xz, midx18(4, 4), Iw.yy, midx18(4, 5), Iw.yz);
    if (leg == 0) ST(M, midx18(5, 5), Iw.zz);
    // base-leg block: rows 0..5, cols 6+jx[k]
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = 6 + jx[k];
      ST(M, 0 * 18 - 0 + c, Mf[k].x);          // midx18(0,c) = c
      ST(M, midx18(1, 1) + (c - 1), Mf[k].y);
      ST(M, midx18(2, 2) + (c - 2), Mf[k].z);
      ST(M, midx18(3, 3) + (c - 3), Mn[k].x);
      ST(M, midx18(4, 4) + (c - 4), Mn[k].y);
      ST(M, midx18(5, 5) + (c - 5), Mn[k].z);
    }
    // leg block (upper triangle in the caller's joint order)
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
      for (int k2 = k1; k2 < 3; ++k2) {
        int i = 6 + jx[k1], j = 6 + jx[k2];
        if (i > j) { int t = i; i = j; j = t; }
        ST(M, i * 18 - i * (i - 1) / 2 + (j - i), Mll[k1][k2]);
      }
    // structural zeros between different legs: 6 leg pairs x 9 = 54 entries, 14 per lane
    for (int e = leg; e < 54; e += 4) {
      const int pr = e / 9, rem = e - 9 * pr, ka = rem / 3, kb = rem - 3 * ka;
      // pair table {01,02,03,12,13,23}
      const int l1 = pr < 3 ? 0 : (pr < 5 ? 1 : 2);
      const int l2 = pr < 3 ? pr + 1 : (pr < 5 ? pr - 1 : 3);
      int i = 6 + model->jidx[l1][ka], j = 6 + model->jidx[l2][kb];
      if (i > j) { int t = i; i = j; j = t; }
      ST(M, i * 18 - i * (i - 1) / 2 + (j - i), Z);
    }
    // h
    T* H = a.h;
    ST4(H, 0, hb_f.x, 1, hb_f.y, 2, hb_f.z, 3, hb_n.x);
    if (leg < 2) ST(H, 4 + leg, leg == 0 ? hb_n.y : hb_n.z);
#pragma unroll
    for (int k = 0; k < 3; ++k) ST(H, 6 + jx[k], h_leg[k]);
    // Jc rows of this lane's foot: (3*leg+m)*18 + c
    T* J = a.Jc;
    const T ONE = (T)1;
    const T jb[3][6] = {{ONE, Z, Z, Z, dw.z, -dw.y}, {Z, ONE, Z, -dw.z, Z, dw.x}, {Z, Z, ONE, dw.y, -dw.x, Z}};
#pragma unroll
    for (int mrow = 0; mrow < 3; ++mrow) {
      const int rb = (3 * leg + mrow) * 18;
#pragma unroll
      for (int c = 0; c < 6; ++c) ST(J, rb + c, jb[mrow][c]);
#pragma unroll
      for (int c = 0; c < 12; ++c) ST(J, rb + 6 + c, Z);
    }
    // own-leg columns overwrite the zeros (same lane, program order)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ST(J, (3 * leg + 0) * 18 + 6 + jx[k], jw[k].x);
      ST(J, (3 * leg + 1) * 18 + 6 + jx[k], jw[k].y);
      ST(J, (3 * leg + 2) * 18 + 6 + jx[k], jw[k].z);
    }
  }
  if (a.pf) {
    ST(a.pf, 3 * leg + 0, qb[0] + dw.x);
    ST(a.pf, 3 * leg + 1, qb[1] + dw.y);
    ST(a.pf, 3 * leg + 2, qb[2] + dw.z);
  }

  // ------------------------------------------------------------------ momentum, beta = C^T v - g
  T p_b[6], beta_b[6], beta_l[3];
  if (OBS) {
    const V3<T> Pl = mul(R, mom[0].f), Pa = mul(R, mom[0].n);
    const V3<T> gl = mul(R, grv[0].f), ga = mul(R, grv[0].n);
    const V3<T> cx = cross(vlin_w, Pl);
    p_b[0] = Pl.x; p_b[1] = Pl.y; p_b[2] = Pl.z; p_b[3] = Pa.x; p_b[4] = Pa.y; p_b[5] = Pa.z;
    beta_b[0] = -gl.x; beta_b[1] = -gl.y; beta_b[2] = -gl.z;
    beta_b[3] = -cx.x - ga.x; beta_b[4] = -cx.y - ga.y; beta_b[5] = -cx.z - ga.z;
#pragma unroll
    for (int k = 0; k < 3; ++k) beta_l[k] = ct_leg[k] - g_leg[k];
    if (a.p) {
      ST4(a.p, 0, p_b[0], 1, p_b[1], 2, p_b[2], 3, p_b[3]);
      if (leg < 2) ST(a.p, 4 + leg, leg == 0 ? p_b[4] : p_b[5]);
#pragma unroll
      for (int k = 0; k < 3; ++k) ST(a.p, 6 + jx[k], p_leg[k]);
    }
    if (a.beta) {
      ST4(a.beta, 0, beta_b[0], 1, beta_b[1], 2, beta_b[2], 3, beta_b[3]);
      if (leg < 2) ST(a.beta, 4 + leg, leg == 0 ? beta_b[4] : beta_b[5]);
#pragma unroll
      for (int k = 0; k < 3; ++k) ST(a.beta, 6 + jx[k], beta_l[k]);
    }
  }

  // ------------------------------------------------------------------ step-mode prologue for the QP
  if (STEP) {
    T rb[6] = {0, 0, 0, 0, 0, 0}, rl[3] = {0, 0, 0};
    if (OBS && prm.observer_order > 0) {
      // generalized force of the previous commands at the current configuration
      const V3<T> fp = mk<T>(a.f_prev[(size_t)(3 * leg + 0) * N + s], a.f_prev[(size_t)(3 * leg + 1) * N + s],
                             a.f_prev[(size_t)(3 * leg + 2) * N + s]);
      const V3<T> ub_f = quad_sum(fp);
      const V3<T> ub_n = quad_sum(cross(dw, fp));
      const T ub[6] = {ub_f.x, ub_f.y, ub_f.z, ub_n.x, ub_n.y, ub_n.z};
      const T dt = prm.dt;
      const bool o1 = prm.observer_order == 1;
#pragma unroll
      for (int c = 0; c < 6; ++c) {  // replicated over the quad (same values in all four lanes)
        const T r0 = a.obs_r[(size_t)c * N + s];
        const T ig = a.obs_integ[(size_t)c * N + s] + dt * (ub[c] + beta_b[c] + r0);
        const T e = p_b[c] - ig;
        rb[c] = o1 ? prm.K1[c] * e : r0 + dt * prm.K2[c] * (prm.K1[c] * e - r0);
        p_b[c] = ig;  // reuse as the new integ for the store below
      }
      // all loads of the replicated rows are done in every lane before any lane stores them
      ST4(a.obs_integ, 0, p_b[0], 1, p_b[1], 2, p_b[2], 3, p_b[3]);
      if (leg < 2) ST(a.obs_integ, 4 + leg, leg == 0 ? p_b[4] : p_b[5]);
      ST4(a.obs_r, 0, rb[0], 1, rb[1], 2, rb[2], 3, rb[3]);
      if (leg < 2) ST(a.obs_r, 4 + leg, leg == 0 ? rb[4] : rb[5]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = 6 + jx[k];
        const T r0 = a.obs_r[(size_t)c * N + s];
        const T u = a.tau_prev[(size_t)jx[k] * N + s] + dot(jw[k], fp);
        const T ig = a.obs_integ[(size_t)c * N + s] + dt * (u + beta_l[k] + r0);
        const T e = p_leg[k] - ig;
        rl[k] = o1 ? prm.K1[c] * e : r0 + dt * prm.K2[c] * (prm.K1[c] * e - r0);
        ST(a.obs_integ, c, ig);
        ST(a.obs_r, c, rl[k]);
      }
    }
    T* ws = a.ws;
    ST(ws, WS_D + 3 * leg + 0, dw.x);
    ST(ws, WS_D + 3 * leg + 1, dw.y);
    ST(ws, WS_D + 3 * leg + 2, dw.z);
    {
      T b[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) b[c] = a.w_des[(size_t)c * N + s] - rb[c];
      ST4(ws, WS_B + 0, b[0], WS_B + 1, b[1], WS_B + 2, b[2], WS_B + 3, b[3]);
      if (leg < 2) ST(ws, WS_B + 4 + leg, leg == 0 ? b[4] : b[5]);
    }
    {
      T ad[6], al[3];
#pragma unroll
      for (int c = 0; c < 6; ++c) ad[c] = a.vdot_des[(size_t)c * N + s];
#pragma unroll
      for (int k = 0; k < 3; ++k) al[k] = a.vdot_des[(size_t)(6 + jx[k]) * N + s];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        T t = h_leg[k] - rl[k];
        t += Mf[k].x * ad[0] + Mf[k].y * ad[1] + Mf[k].z * ad[2] + Mn[k].x * ad[3] + Mn[k].y * ad[4] + Mn[k].z * ad[5];
        t += Mll[k][0] * al[0] + Mll[k][1] * al[1] + Mll[k][2] * al[2];
        ST(ws, WS_TAUP + 3 * leg + k, t);
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      ST(ws, WS_JCL + 9 * leg + 0 + k, jw[k].x);
      ST(ws, WS_JCL + 9 * leg + 3 + k, jw[k].y);
      ST(ws, WS_JCL + 9 * leg + 6 + k, jw[k].z);
    }
  }
#undef ST4
#undef ST
#undef CS
}

}  // namespace wbc

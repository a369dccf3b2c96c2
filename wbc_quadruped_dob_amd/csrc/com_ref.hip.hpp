// CoM reference generator: SURVEY.md 8(f)-3, "a motion planner for the trajectory of the robot's center of mass"
// (/root/reference/README.md:11) -- the caller on the INPUT side of the hot path: it produces the (w_des, vdot_des)
// that wbc_step_batch consumes.  The planner's source is absent from the reference, so this is the build's own
// definition (stated in include/wbc_hip.h at wbc_reference_batch):
//   plan [12][N]: c0 (3) start CoM, c1 (3) goal CoM (world), T duration, t0 elapsed, quat_des (x,y,z,w)
//   rest-to-rest quintic in u = clamp((t0+t)/T, 0, 1); PD on CoM position/velocity and on attitude; joint posture PD;
//   w_des = [F ; (c - p_b) x F + R diag(I_nom) R^T alpha_cmd],  F = m_tot (a_cmd - g).
//
// Same lane mapping as the dynamics sweep (lane = 16*leg + state): every lane runs the kinematics of ITS leg, the
// leg's first moment (m, m c) and linear momentum go leaf -> root in link coordinates, and the four legs are summed
// with v_permlane16/32_swap.  In 45 + 12 words, out 24 (+6): HBM/latency-trivial next to the sweep.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"
#include "dyn_sweep.hip.hpp"

namespace wbc {

// EXT (persistent rollout kernel, fused_tick.hip.hpp): one wavefront of a larger workgroup, tables already in LDS.
template <class T, bool EXT, int SPW = 16, bool SIMG = false>   // SIMG: role of a rollout workgroup that keeps states / plans / references in LDS
WBC_DEV void com_reference_body(const DevModel<T>* __restrict__ model, const DevRefParams<T>* __restrict__ G, const RefArgs<T>& a,
                                const T* cst_ext) {
  __shared__ T cst_own[EXT ? 1 : CST_WORDS];
  if constexpr (!EXT) {
    for (int i = threadIdx.x; i < CST_WORDS; i += blockDim.x) cst_own[i] = model->cst[i];
    __syncthreads();
  }
  const T* cst = EXT ? cst_ext : cst_own;
  unsigned tx = threadIdx.x;
  asm volatile("" : "+v"(tx));   // see WBC_LAUNDERED_TID (dyn_split.hip.hpp)
  const size_t N = a.N;
  const unsigned N32 = (unsigned)N;
  const int leg = (int)((tx & 63) >> 4);
  const size_t s_raw = (size_t)blockIdx.x * SPW + (tx & 15);
  const bool slot_ok = SPW == 16 || (int)(tx & 15) < SPW;   // (SPW <= 16 states per workgroup, see WBC_ADDR_MACROS)
  const bool live = slot_ok && s_raw < N;
  const unsigned s32 = (unsigned)(live ? s_raw : (slot_ok ? N - 1 : (size_t)blockIdx.x * SPW));
#define RCS(i) cst[(i) * 4 + leg]
#define RLDU(ptr, comp) (*(const T*)((const char*)((ptr) + (size_t)(comp) * N) + (size_t)(s32 * (unsigned)sizeof(T))))
#define RLDV(ptr, comp) (*(const T*)((const char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))))
#define RSTV(ptr, comp, val) do { if (live) *(T*)((char*)(ptr) + (size_t)(((unsigned)(comp) * N32 + s32) * (unsigned)sizeof(T))) = (val); } while (0)
  static_assert(!SIMG || EXT, "the LDS images belong to the rollout kernel's roles");   // (the state from the workgroup's LDS image: WBC_STATE_MACROS, dyn_split.hip.hpp)
  const T* const si_ = SIMG ? a.simg + (int)(s32 - (unsigned)((size_t)blockIdx.x * SPW)) : nullptr;
  T qb[7], vb[6], pl[PLAN_WORDS];
#pragma unroll
  for (int c = 0; c < 7; ++c) qb[c] = SIMG ? si_[c * 16] : RLDU(a.q, c);
#pragma unroll
  for (int c = 0; c < 6; ++c) vb[c] = SIMG ? si_[(SIMG_V + c) * 16] : RLDU(a.v, c);
#pragma unroll
  for (int c = 0; c < PLAN_WORDS; ++c) pl[c] = SIMG ? a.planimg[c * 16 + (si_ - a.simg)] : RLDU(a.plan, c);
  int jx[3];
  jidx_of_leg(model, a.jpack, leg, jx);
  T ql[3], vl[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    ql[k] = SIMG ? si_[(7 + jx[k]) * 16] : RLDV(a.q, 7 + jx[k]);
    vl[k] = SIMG ? si_[(SIMG_V + 6 + jx[k]) * 16] : RLDV(a.v, 6 + jx[k]);
  }
  T qx, qy, qz, qw;
  {
    const T n = rsqrt_sel<SIMG>(qb[3] * qb[3] + qb[4] * qb[4] + qb[5] * qb[5] + qb[6] * qb[6]);
    qx = qb[3] * n; qy = qb[4] * n; qz = qb[5] * n; qw = qb[6] * n;
  }
  M3<T> R;
  {
    const T x = qx, y = qy, z = qz, w = qw;
    R.a[0] = 1 - 2 * (y * y + z * z); R.a[1] = 2 * (x * y - z * w);     R.a[2] = 2 * (x * z + y * w);
    R.a[3] = 2 * (x * y + z * w);     R.a[4] = 1 - 2 * (x * x + z * z); R.a[5] = 2 * (y * z - x * w);
    R.a[6] = 2 * (x * z - y * w);     R.a[7] = 2 * (y * z + x * w);     R.a[8] = 1 - 2 * (x * x + y * y);
  }
  const V3<T> om0 = tmul(R, mk<T>(vb[3], vb[4], vb[5]));
  const V3<T> v0 = tmul(R, mk<T>(vb[0], vb[1], vb[2]));

  // forward sweep down the leg: joint rotations, body velocities, body linear momenta (link coordinates)
  M3<T> E[3];
  V3<T> pk[3];
  V3<T> omp = om0, vp = v0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int o = JOINT_WORDS * k;
    T sn, cs;
    sincos_t(ql[k], &sn, &cs);
#pragma unroll
    for (int e = 0; e < 9; ++e) E[k].a[e] = RCS(o + e) + cs * RCS(o + 9 + e) + sn * RCS(o + 18 + e);
    const V3<T> r = mk<T>(RCS(o + 27), RCS(o + 28), RCS(o + 29));
    const V3<T> ax = mk<T>(RCS(o + 30), RCS(o + 31), RCS(o + 32));
    const V3<T> h = mk<T>(RCS(o + 34), RCS(o + 35), RCS(o + 36));
    const V3<T> om = tmul(E[k], omp) + ax * vl[k];
    const V3<T> vv = tmul(E[k], vp + cross(omp, r));
    pk[k] = vv * RCS(o + 33) + cross(om, h);
    omp = om; vp = vv;
  }
  // return sweep: first moment (cm, ch) and linear momentum P of the leg, expressed in the base frame at the end
  T cm = RCS(2 * JOINT_WORDS + 33);
  V3<T> ch = mk<T>(RCS(2 * JOINT_WORDS + 34), RCS(2 * JOINT_WORDS + 35), RCS(2 * JOINT_WORDS + 36));
  V3<T> P = pk[2];
#pragma unroll
  for (int k = 2; k >= 0; --k) {
    const int o = JOINT_WORDS * k;
    const V3<T> r = mk<T>(RCS(o + 27), RCS(o + 28), RCS(o + 29));
    ch = mul(E[k], ch) + r * cm;       // now in the parent's frame
    P = mul(E[k], P);
    if (k > 0) {
      const int op = JOINT_WORDS * (k - 1);
      cm += RCS(op + 33);
      ch = ch + mk<T>(RCS(op + 34), RCS(op + 35), RCS(op + 36));
      P = P + pk[k - 1];
    }
  }
  const T bm = model->base_m;
  const V3<T> bh = mk<T>(model->base_h[0], model->base_h[1], model->base_h[2]);
  const T mtot = xrow_sum(cm) + bm;
  const V3<T> hb = xrow_sum(ch) + bh;
  const V3<T> Pb = xrow_sum(P) + v0 * bm + cross(om0, bh);
  const T im = (T)1 / mtot;
  const V3<T> crel = mul(R, hb) * im;                  // CoM relative to the base origin, world axes
  const V3<T> c = crel + mk<T>(qb[0], qb[1], qb[2]);
  const V3<T> cd = mul(R, Pb) * im;

  // quintic time law
  const T Tp = pl[6];
  const bool hasT = Tp > (T)0;
  const T iT = hasT ? (T)1 / Tp : (T)0;
  T u = hasT ? (pl[7] + a.t) * iT : (T)1;
  u = u < (T)0 ? (T)0 : (u > (T)1 ? (T)1 : u);
  const T u2 = u * u, u3 = u2 * u;
  const T s0 = u3 * (10 + u * (-15 + 6 * u));
  const T s1 = u2 * (30 + u * (-60 + 30 * u)) * iT;
  const T s2 = u * (60 + u * (-180 + 120 * u)) * iT * iT;
  T acmd[3], alcmd[3];
  {
    const T cc[3] = {c.x, c.y, c.z}, cdv[3] = {cd.x, cd.y, cd.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const T d = pl[3 + i] - pl[i];
      acmd[i] = s2 * d + G->kp_com[i] * (pl[i] + s0 * d - cc[i]) + G->kd_com[i] * (s1 * d - cdv[i]);
    }
  }
  {
    const T dn = rsqrt_sel<SIMG>(pl[8] * pl[8] + pl[9] * pl[9] + pl[10] * pl[10] + pl[11] * pl[11]);
    const T dx = pl[8] * dn, dy = pl[9] * dn, dz = pl[10] * dn, dw = pl[11] * dn;
    const T x = -qx, y = -qy, z = -qz, w = qw;
    T ex = dw * x + dx * w + dy * z - dz * y;
    T ey = dw * y - dx * z + dy * w + dz * x;
    T ez = dw * z + dx * y - dy * x + dz * w;
    const T ew = dw * w - dx * x - dy * y - dz * z;
    const T sg = ew < (T)0 ? (T)-2 : (T)2;
    alcmd[0] = G->kp_rot[0] * (sg * ex) - G->kd_rot[0] * vb[3];
    alcmd[1] = G->kp_rot[1] * (sg * ey) - G->kd_rot[1] * vb[4];
    alcmd[2] = G->kp_rot[2] * (sg * ez) - G->kd_rot[2] * vb[5];
  }
  const V3<T> F = mk<T>(acmd[0] - model->grav[0], acmd[1] - model->grav[1], acmd[2] - model->grav[2]) * mtot;
  const V3<T> lb = tmul(R, mk<T>(alcmd[0], alcmd[1], alcmd[2]));
  const V3<T> Mo = cross(crel, F) + mul(R, mk<T>(G->inertia_nom[0] * lb.x, G->inertia_nom[1] * lb.y, G->inertia_nom[2] * lb.z));

  // outputs: the base-replicated words are dealt over the four leg rows (no redundant store), joints by their owner
  T adj[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) adj[k] = G->kp_joint * (G->q_nom[jx[k]] - ql[k]) - G->kd_joint * vl[k];
  bool to_mem = true;
  if constexpr (SIMG) {   // the rnea role of the same workgroup reads the references in LDS (its caller raises a flag behind an LDS-only fence)
    T* const ri = a.refimg + (si_ - a.simg);
    ri[sel4<int>(leg, 0, 1, 2, 3) * 16] = sel4<T>(leg, F.x, F.y, F.z, Mo.x);
    if (leg < 2) ri[(4 + leg) * 16] = leg == 0 ? Mo.y : Mo.z;
    ri[(6 + sel4<int>(leg, 0, 1, 2, 3)) * 16] = sel4<T>(leg, acmd[0], acmd[1], acmd[2], alcmd[0]);
    if (leg >= 2) ri[(6 + 2 + leg) * 16] = leg == 2 ? alcmd[1] : alcmd[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) ri[(6 + 6 + jx[k]) * 16] = adj[k];
    to_mem = a.skip_out == 0;
  }
  if (to_mem) {
    RSTV(a.w_des, sel4<int>(leg, 0, 1, 2, 3), sel4<T>(leg, F.x, F.y, F.z, Mo.x));
    if (leg < 2) RSTV(a.w_des, 4 + leg, leg == 0 ? Mo.y : Mo.z);
    RSTV(a.vdot_des, sel4<int>(leg, 0, 1, 2, 3), sel4<T>(leg, acmd[0], acmd[1], acmd[2], alcmd[0]));
    if (leg >= 2) RSTV(a.vdot_des, 2 + leg, leg == 2 ? alcmd[1] : alcmd[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) RSTV(a.vdot_des, 6 + jx[k], adj[k]);
  }
  if (a.com) {
    RSTV(a.com, sel4<int>(leg, 0, 1, 2, 3), sel4<T>(leg, c.x, c.y, c.z, cd.x));
    if (leg < 2) RSTV(a.com, 4 + leg, leg == 0 ? cd.y : cd.z);
  }
#undef RSTV
#undef RLDV
#undef RLDU
#undef RCS
}

template <class T>
__global__ __launch_bounds__(64) void com_reference_kernel(const DevModel<T>* __restrict__ model,
                                                           const DevRefParams<T>* __restrict__ G, RefArgs<T> a) {
  com_reference_body<T, false>(model, G, a, nullptr);
}

}  // namespace wbc

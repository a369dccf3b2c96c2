// Shared host/device plain-data types of the HIP path (product code).
#pragma once
#include <cstddef>

namespace wbc {

// ---- per-leg model constants, as read by the dynamics-sweep kernel -------------------------
// Table layout in memory: cst[idx * 4 + leg]; one lane owns one leg, so a wave reads four
// distinct consecutive words per table row (conflict-free in LDS).
// Per joint k (0..2) of a leg, 43 words at offset 43*k:
//   +0  A0[9]  = Rt a a^T            E(q) = A0 + cos(q) A1 + sin(q) A2   (child -> parent rotation)
//   +9  A1[9]  = Rt - A0
//   +18 A2[9]  = Rt [a]x
//   +27 rt[3]    joint origin in parent coordinates
//   +30 ax[3]    joint axis in child coordinates
//   +33 m
//   +34 h[3]   = m * com
//   +37 Io[6]    rotational inertia about the link origin (xx,xy,xz,yy,yz,zz)
// then foot_off[3] at offset 129.  132 words per leg.
constexpr int JOINT_WORDS = 43;
constexpr int LEG_WORDS = 3 * JOINT_WORDS + 3;  // 132
constexpr int CST_WORDS = LEG_WORDS * 4;        // 528

template <class T> struct DevModel {
  T cst[CST_WORDS];
  T base_m, base_h[3], base_Io[6];
  T grav[3];
  int jidx[4][3];  // joint index (0..nj-1) of leg l joint k in the caller's q/v ordering
  int zidx[64];    // packed-M indices that are structurally zero (cross-leg blocks, base block), -1 padded
};

template <class T> struct DevParams {
  T S[6];
  T alpha, fn_min, fn_max, mu_scale, dt, qp_tol;
  int observer_order, max_iter;
  T K1[18], K2[18];
};

// step-mode workspace written by the sweep kernel and read by the QP kernel: ws[c * N + s]
//   0..11  d      foot position relative to the base origin, world axes (foot-major)
//   12..17 b      QP target wrench  w_des - rhat_base
//   18..29 taup   (M vdot_des + h - rhat) joint rows, leg-major (leg l joint k at 18+3l+k)
//   30..65 JcL    own-leg Jacobian block of foot l: 30 + 9 l + 3 m + k = d pf_m / d q_{l,k}
constexpr int WS_D = 0, WS_B = 12, WS_TAUP = 18, WS_JCL = 30, WS_WORDS = 66;
// LDS image of the fused tick only: 66..83 rhat (observer estimate: base rows 6, joint rows leg-major 12)
constexpr int WS_RHAT = 66, WS_LDS_WORDS = 84;

template <class T> struct SweepArgs {
  size_t N;
  const T* q; const T* v;
  // dynamics outputs (nullable as a group: M,h,Jc; pf, p, beta individually)
  T* M; T* h; T* Jc; T* pf; T* p; T* beta;
  // step mode
  const T* w_des; const T* vdot_des; const T* tau_prev; const T* f_prev;
  T* obs_integ; T* obs_r;
  T* ws;
  int ws_geom;   // 1: also write d and the own-leg Jacobian blocks to the workspace; 0: the QP reads them from Jc (M/h/Jc ticks)
};

template <class T> struct QpArgs {
  size_t N;
  const T* ws; const T* normals; const T* mu; const int* mask;
  const T* Jc;   // non-null: foot lever arms and own-leg Jacobian blocks come from the Jacobian the sweep wrote, not from ws
  T* tau; T* f; int* status; int* iters;
};

}  // namespace wbc

// Shared host/device plain-data types of the HIP path (product code).
#pragma once
#include <cstddef>

// (Stores of pure outputs by lanes beyond the batch carry no guard: those lanes recompute the last state, so what they store is a bit-identical duplicate at
//  the same address -- no exec region per store.  Joint indices come from the packed kernel argument jpack: jidx_of_leg, dyn_sweep.hip.hpp.)

#define WBC_DEV __device__ __forceinline__

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
  T sS[6];                        // sqrt(S)
  T sqrt_alpha, rsqrt_alpha;      // sqrt(alpha), 1 / sqrt(alpha)
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
// Roles of a tile tick (tile_tick.hip.hpp) share their workgroup's constant table and hand their step-workspace words to the QP stage in LDS instead of through memory.
// cst: the workgroup's table (CST_WORDS), staged by `stage_threads` role threads (threadIdx.x below that) in front of the roles' common barrier;
// hand: [HAND_ROWS][hs] words, row r of the tile's state c at hand[r * hs + c] (tau_partial from the sweep role, rhat from the observer role);
// col0: the tile column of this wavefront's first state
constexpr int HAND_TAUP = 0, HAND_RHAT = 12, HAND_ROWS = 30;
template <class T> struct RoleShare { T* cst; T* hand; int hs; int stage_threads; int col0; };

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
  int* qp_todo;  // non-null: the hand-over list of the per-lane QP kernel that follows; this kernel empties it (qp_lane.hip.hpp)
  int skip_consts;  // 1: the structural zeros / ones of M and Jc are already in the caller's buffers (wbc_solver_options.keep_structural): not rewritten
  int skip_mats;    // (persistent rollout, set by the kernel) 1: nobody will read the M / Jc / pf of THIS tick from memory -- the integrator takes them from
                    // the mass_jac role's LDS image -- so they are not stored (every tick of a launch but the last)
  unsigned long long jpack;   // the caller's joint index of leg l joint k in nibble 3 l + k (pack_jidx): the bodies take it from these two
                              // SGPRs instead of loading DevModel::jidx -- a per-lane global load in FRONT of the joint-state loads, i.e. one
                              // more dependent trip through L2 at the head of every role of every tick (round 5)
  const T* simg;              // (set by the persistent rollout kernel, never by the host) the workgroup's state image in LDS: see WBC_RO_MERGE
  const T* refimg;            // (likewise, planner in the loop) this tick's references [24][16]: w_des (rows 0 .. 5), vdot_des (6 .. 23), written by the planner role
  const T* resimg;            // (likewise) the result image: the observer role of a 4-state workgroup takes tau_prev, f_prev from the rows the QP of the
                              // previous tick wrote there, instead of from memory
};
// The state image of a 4-state rollout workgroup (WBC_RO_MERGE, fused_tick.hip.hpp): q (rows 0 .. 18) and v (19 .. 36) of the workgroup's states,
// [row][16 slots], in LDS for the whole launch.  The integrator writes the new state there (and to memory); the roles of the next tick read it from
// there instead of waiting for those stores and a trip through L2 at the head of the tick.
constexpr int SIMG_V = 19, SIMG_WORDS = 37;
// The result image of a rollout workgroup (QpSync::res, qp_group16.hip.hpp): this tick's tau (rows 0 .. 11, caller's joint order), f (12 .. 23), h (24 .. 41),
// [row][16 slots]; rows 42 .. 59 hold the external torques of the workgroup's states for the whole launch.
constexpr int RES_TAU = 0, RES_F = 12, RES_H = 24, RES_WORDS = 42;

template <class T> struct QpArgs {
  size_t N;
  const T* ws; const T* normals; const T* mu; const int* mask;
  const T* Jc;   // non-null: foot lever arms and own-leg Jacobian blocks come from the Jacobian the sweep wrote, not from ws
  const T* wdes; // non-null: the target wrench b is read from the caller's w_des (the front half forwarded nothing to WS_B)
  T* tau; T* f; int* status; int* iters;
  const int* aset_in;  // warm-start kernels: the active set each state's iteration starts from (null: cold); encoding: include/wbc_hip.h
  int* aset_out;       // non-null: receives the active set at the solution (structured QP kernels)
  const T* rprev;      // fused observer-on ticks: the observer state r as the tick finds it ([18][N], base rows used) -- the QP starts on b~ = w_des - r_prev
                       // while the observer role is still computing r (qp_struct16.hip.hpp, SPEC); null: opt out -- the QP waits for rhat as before
  unsigned long long jpack;   // QpJidx as nibbles (pack_jidx): what the structured / dense QP bodies index the torque map with
};

struct QpJidx { int j[12]; };  // caller's joint index of leg-major joint 3l+k
inline unsigned long long pack_jidx(const int* j12) {   // nibble 3 l + k = joint index (0 .. 11) of leg l joint k
  unsigned long long p = 0;
  for (int i = 0; i < 12; ++i) p |= (unsigned long long)(j12[i] & 15) << (4 * i);
  return p;
}

// forward dynamics + integrator (integrate.hip.hpp)
template <class T> struct IntegrateArgs {
  size_t N;
  T* q; T* v;                       // in/out
  const T* M; const T* h; const T* Jc;   // from the sweep of the same tick
  const T* tau; const T* f;         // this tick's outputs
  const T* tau_ext;                 // [nv][N] or null
  T* tau_traj;                      // [nj][N] slice for this tick, or null
  T dt;
  unsigned long long jpack;         // see SweepArgs::jpack
  T* simg;                          // see SweepArgs::simg (the integrator reads the state there and writes the new one to both)
  int skip_state;                   // (set by the persistent rollout kernel) 1: the new q, v go to the LDS image only -- every tick of a launch but the last
#ifdef WBC_FUSED_STAMP   // diagnostic build (tools/rollout_stamp.py): the integrator's own phases, column 1 of its workgroup
  double* istamp; unsigned istampN;
#endif
};

// CoM reference generator (com_ref.hip.hpp)
template <class T> struct DevRefParams {
  T kp_com[3], kd_com[3], kp_rot[3], kd_rot[3];
  T kp_joint, kd_joint;
  T inertia_nom[3];
  T q_nom[12];  // in the caller's joint ordering
};
constexpr int PLAN_WORDS = 12;
template <class T> struct RefArgs {
  size_t N;
  const T* q; const T* v; const T* plan;
  T t;
  T* w_des; T* vdot_des;
  T* com;   // [6][N] or null
  unsigned long long jpack;   // see SweepArgs::jpack
  const T* simg;              // see SweepArgs::simg
  T* refimg;                  // see SweepArgs::refimg: where the planner role of a 4-state rollout workgroup leaves the references (and in memory in the last tick)
  const T* planimg;           // (likewise) the plans of the workgroup's states [PLAN_WORDS][16], parked in LDS for the whole launch
  int skip_out;               // (likewise) 1: w_des, vdot_des go to the LDS image only -- every tick of a launch but the last
};

// MODE bits of dyn_sweep_kernel (dyn_sweep.hip.hpp)
constexpr int SW_MATS = 1;  // write M, h, Jc
constexpr int SW_STEP = 2;  // write the step workspace (d, b, taup, JcL)
constexpr int SW_OBS = 4;   // momentum observer update (needs SW_STEP) / p, beta outputs
constexpr int SW_NOB = 8;   // step mode without forwarding w_des to the workspace: the QP kernel reads the caller's w_des (QpArgs::wdes)
// MODE bits of rnea_step_kernel (dyn_split.hip.hpp)
constexpr int RS_H = 1;     // write h (bias forces)
constexpr int RS_STEP = 2;  // write the step workspace (d, b, taup, JcL)
constexpr int RS_OBS = 4;   // momentum / gravity recursions: p, beta outputs and the observer update
constexpr int RS_PF = 8;    // write pf (when mass_jac does not run)
constexpr int RS_OBSW = 16; // observer ROLE of the fused tick (with RS_OBS, without RS_STEP / RS_H): no force recursion, the
                            // momentum observer is updated and rhat (18 words) goes to the LDS image at WS_RHAT
constexpr int RS_NOB = 32;  // (stand-alone kernel, observer off) w_des is not forwarded to the workspace: see SW_NOB
constexpr int RS_REFIMG = 256; // (role with RS_STEP, four-wavefront rollout workgroups with the planner in the loop) w_des, vdot_des come from the planner role's LDS
                               // image (SweepArgs::refimg), not from memory
constexpr int RS_NOJC = 128; // (role with RS_STEP, four-wavefront rollout workgroups) the own-leg Jacobian blocks are NOT propagated up the return sweep and not written
                             // to WS_JCL: the QP's torque map takes them from the mass_jac role's LDS image, which is complete ~2 us before this role ends
constexpr int RS_LANE2 = 64; // (roles with 4 states per workgroup, RS_STEP | RS_H) the two force recursions SIDE BY SIDE in the lanes instead of one after the
                             // other: slots 0 .. 3 of every leg row run RNEA(q, v, 0) (the bias forces h), slots 4 .. 7 RNEA(q, 0, vdot_des) without gravity
                             // (M vdot_des) of the same states, as ONE instruction stream; tau_partial = their sum, across lanes (round 5)

}  // namespace wbc

// Fused control tick for SMALL batches: one kernel instead of {dyn_sweep -> qp_group16} while the batch fits one
// workgroup per CU (N <= 4 096 = 256 CUs x 16 states; the bench default).  At that size both kernels are latency-bound
// (one sweep wavefront and four QP wavefronts per CU, nothing to hide behind), and the two-kernel tick pays on top of
// the arithmetic: two launches, the end-of-kernel drain of 3.5 kB/state of M/h/Jc stores -- which the QP never reads
// -- before the QP may start, and a round trip of the 66-word step workspace through HBM.  Here a workgroup owns 16
// consecutive states and hands the workspace over in LDS; M/h/Jc stores drain behind the QP.
//
// The front half is split by consumer into wavefront roles (below).  N = 4 096, observer off: 22.3 us per tick against
// 32.5 us for the two-kernel tick (184 M vs 126 M control-steps/s); observer on (two more wavefronts): 21.2 us against
// 34.3 us (193 M vs 120 M).  (A whole-sweep-then-QP fusion was measured before this one: -7 % / no gain.)
// Larger batches keep the two-kernel tick: the sweep is HBM-bound there and wants all lanes of 8 waves per CU, which
// the fused kernels' LDS (100 kB per workgroup in fp64) does not allow; measured slower from N = 8 192 on.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"
#include "dyn_split.hip.hpp"
#include "qp_struct16.hip.hpp"
#include "integrate.hip.hpp"
#include "com_ref.hip.hpp"
#include "observer.hip.hpp"

namespace wbc {

// fused tick, observer on: the observer role (observer_body: the register-only body of observer.hip.hpp, no LDS parking) is TWO wavefronts -- base rows -> rhat_base,
// which the QP's b waits for; joint rows -> rhat_joint, needed only in the torque map.  (One wavefront doing both, and rnea_step_body as the observer role, were the
// forms of rounds 2-3: docs/DESIGN_R04.md.)  The structural zeros / ones of M and Jc are written by the four QP wavefronts while they wait for the lever arms, not by
// the mass_jac role (~55 store instructions = ~4 us of store issue off that role's path).
constexpr int FUSED_OBS_WAVES = 2;

// The front half is SPLIT by consumer.  Six wavefronts per workgroup of
// 16 states: wave 4 runs rnea_step_body (bias forces h, and the 66-word step workspace -- all the QP needs -- into LDS),
// wave 5 runs mass_jac_body (M, Jc, pf: 3.2 kB/state that nobody on the GPU reads), waves 0..3 wait for wave 4's flag
// and run the GRF QP.  The QP therefore starts after the ~1/3 of the dynamics it depends on, and the CRBA and its
// stores overlap with it on otherwise idle issue slots.  No barrier after the table staging: the hand-over is an
// LDS flag (the four QP waves poll it; all waves of a workgroup are resident, so the producer always runs).
// OBSERVER on: two more wavefronts take the observer role (observer_body: velocities, momenta, gravity terms,
// beta = C^T v - g, the update of {integ, r}), split by rows: wave 6 the base rows (rhat_base -> LDS, flag `oready`; the
// QP's b waits for it), wave 7 the joint rows (rhat_joint -> LDS, counted on `ready`; needed in the torque map only).
// The QP waves subtract rhat from b and tau_partial themselves.
// MATS = false (the caller wants tau, f only): no mass_jac role, and the rnea role runs the single merged force chain.
// Staged hand-over (QpSync, qp_group16.hip.hpp): the rnea role publishes the four lever arms right after its state
// loads and w_des right behind them (`gready` counts both) -- H and its factor need the former only, b the latter --
// rhat follows from the observer role (`oready`,
// first needed for g = -A^T S b) and tau_partial + the own-leg Jacobian blocks when the force recursions are done
// (`ready`, first needed in the torque map).  Observer off, N = 4 096: 25.5 -> 22.5 us per tick.
// Observer off, M/h/Jc wanted: the rnea role is TWO wavefronts (a seventh wavefront; with the observer on the CU's eight slots
// are taken).  Wave 4 runs the ONE merged force recursion RNEA(q, v, vdot_des) -- tau_partial, all the QP's torque map waits
// for -- and wave 6 the bias-force recursion whose only consumer is the caller's h buffer.  One wavefront doing both chains
// (round 2) kept the QP wavefronts waiting for tau_partial until +9.1 us.
// 1 (default, round 5): the observer role stores r only after the QP wavefronts have read r_prev (QpSync::rp_ack); 0: the unordered read of round 4 (A/B)
// (fp64 only: the fp32 tick fits two six-wavefront workgroups on a CU -- 147 VGPRs, 49 kB LDS -- and a seventh wavefront would end that)
// WARM ticks are another matter: the block set-up ends the QP at about +5.5 us, so the tick ends with the rnea role and its torque map, and taking
// the bias-force chain off that role shows: tick kernel 12.0 -> 11.0 us at 1 024 states, 13.2 -> 12.6 at 4 096, 22.2 -> 21.3 at 8 192 in a closed
// loop of drifting states (13.5 -> 13.2 on the bench's own batch; tools/ab_libs.sh with --closed-loop).  On by default for the warm instantiation.
// (the fp32 WARM instantiation holds 254 registers -- one workgroup per CU whatever the wavefront count; with the split 12.4 -> 12.1 us at 4 096 drifting states: too little
//  to carry another instantiation, so fp64 only)
template <class T, bool OBSERVER, bool MATS, bool WARM = false> constexpr bool fused_split_h() {
  return !OBSERVER && MATS && WARM && sizeof(T) == 8;
}
template <class T, bool OBSERVER, bool MATS, bool WARM = false> constexpr int fused_threads() { return OBSERVER ? 384 + 64 * FUSED_OBS_WAVES : (fused_split_h<T, OBSERVER, MATS, WARM>() ? 448 : 384); }
// WARM: the QP of every state starts from the active set in qa.aset_in (wbc_step_batch_warm: dependent ticks of a closed loop)
template <class T, bool OBSERVER, bool MATS, bool WARM = false>
__global__ __launch_bounds__((fused_threads<T, OBSERVER, MATS, WARM>()), 1) void fused_tick_kernel(const DevModel<T>* __restrict__ model, DevParams<T> prm,
                                                                            SweepArgs<T> a, QpArgs<T> qa, QpJidx jmap) {
  __shared__ __attribute__((aligned(512))) T cst[CST_WORDS];   // (the alignment puts the table FIRST in the workgroup's LDS: within reach of the 16-bit ds_read offset, see dyn_sweep.hip.hpp)
  __shared__ int zidx_s[64];
  __shared__ T wsl[WS_LDS_WORDS * 16];
  __shared__ int ready, gready, oready;   // rnea role done / its lever arms are out / observer role done
  __shared__ int rpack;                   // QP wavefronts that have read r_prev (speculative start): the observer role stores r behind all four
  constexpr bool SPEC_ORDER = OBSERVER && !WARM;
  const int wave = (int)(threadIdx.x >> 6);
  // (issue priorities for the QP wavefronts / the rnea role / everybody above the mass_jac role: 13.2 -> 13.4 ... 13.5 us, DESIGN_R05.md section 9)
#ifdef WBC_FUSED_STAMP   // diagnostic build: the pf output carries the role timestamps (slot, workgroup) instead of foot positions
  double* const stamp = (double*)a.pf;
  const unsigned stampN = (unsigned)a.N;
  a.pf = nullptr;
#define FSTAMP(slot) do { if (stamp) WBC_FSTAMP(stamp, stampN, slot); } while (0)
  if (wave == 0) FSTAMP(0);
#else
#define FSTAMP(slot) do {} while (0)
#endif
  // The four QP wavefronts stage the tables; the producers issue their state loads first and join the ONE workgroup
  // barrier from inside their bodies (EXT = 2), so table staging and state loads share a memory round trip.
  if (wave == 4) {
    int* const gflag = &gready;
    constexpr int RMODE = (MATS && !fused_split_h<T, OBSERVER, MATS, WARM>()) ? (RS_STEP | RS_H) : RS_STEP;
    auto geom_out = [=] __device__() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(gflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      FSTAMP(7);
    };
    // (tau_partial handed to the QP wavefronts before the base rows of h are summed, rotated and stored: measured neutral here, profiles/r05*_ab_*taup_first*.log)
    rnea_step_body<T, RMODE, 64, 2>(model, prm, a, cst, wsl, NoWait(), geom_out);
    FSTAMP(8);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");     // my LDS writes first (lgkmcnt only) ...
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ... then the flag
  } else if (wave == 5) {
    if constexpr (MATS) mass_jac_body<T, 64, 2, 16, (0)>(model, a, cst, zidx_s);   // (2: the role writes them LAST)
    else __syncthreads();
    FSTAMP(9);
  } else if (fused_split_h<T, OBSERVER, MATS, WARM>() && wave == 6) {
    if constexpr (fused_split_h<T, OBSERVER, MATS, WARM>()) rnea_step_body<T, RS_H, 64, 2>(model, prm, a, cst, wsl);   // bias forces h -> HBM only
  } else if (OBSERVER && wave == 6) {
    if constexpr (OBSERVER) {
      int* const ack = &rpack;
      auto wait_ack = [ack] __device__() {   // (see QpSync::rp_ack; the QP wavefronts count at about +2.6 us, this role gets here at about +6)
        if constexpr (SPEC_ORDER) { while (__hip_atomic_load(ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(1); }
      };
      observer_body<T, 64, 2, 1, 16, decltype(wait_ack)>(model, prm, a, cst, wsl, wait_ack);   // base rows
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&oready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      FSTAMP(10);      // (observer builds: slot 10 is the observer role's end, otherwise QP wave 3's)
    }
  } else if (OBSERVER && FUSED_OBS_WAVES == 2 && wave == 7) {
    if constexpr (OBSERVER && FUSED_OBS_WAVES == 2) {
      observer_body<T, 64, 2, 2>(model, prm, a, cst, wsl);                                      // joint rows
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // counts with the rnea role: both feed the torque map
    }
  } else {
    for (int i = threadIdx.x; i < CST_WORDS; i += 256) cst[i] = model->cst[i];
    if (threadIdx.x < 64) zidx_s[threadIdx.x] = model->zidx[threadIdx.x];
    if (threadIdx.x == 0) { ready = 0; gready = 0; oready = 0; rpack = 0; }
    __syncthreads();
    constexpr int NFIN = (OBSERVER && FUSED_OBS_WAVES == 2) ? 2 : 1;   // rnea role (+ the observer's joint-row wavefront)
#ifdef WBC_FUSED_STAMP
    QpSync sy{&gready, &oready, &ready, 1, 2, 1, NFIN, stamp, stampN};
#else
    QpSync sy{&gready, &oready, &ready, 1, 2, 1, NFIN};   // the QP waits for each piece where it first needs it
#endif
    if constexpr (SPEC_ORDER) sy.rp_ack = &rpack;
    if constexpr (MATS) {
      const int* const zs = zidx_s;
      const unsigned tq = threadIdx.x;
      auto idle = [=] __device__() { if (!a.skip_consts) structural_consts_quarter<T>(model, a, zs, tq); };
      qp_body<T, true, OBSERVER, 16, false, 4, decltype(idle), false, WARM ? 1 : 0>(prm, qa, jmap, wsl, &sy, QpWho{0, false}, idle);
    } else
    qp_body<T, true, OBSERVER, 16, false, 4, QpNoIdle, false, WARM ? 1 : 0>(prm, qa, jmap, wsl, &sy);
  }
}

// fused_pair_kernel (round 6): TWO of the observer-off, cold, M/h/Jc-writing tick workgroups above as ONE workgroup of twelve wavefronts and 32 states, for batches
// between one round of the 16-state workgroups and one (fp64) or two (fp32) rounds of pairs.  Why a pair and not launch bounds: the fp64 six-wavefront workgroup holds 214
// registers, so a CU takes one of them and 6 144 states cost two full rounds (23.9 us against 13.3 for 4 096).  Compiled to 168 registers (three wavefronts per SIMD) two of
// them still do NOT share a CU: the dispatcher deals a workgroup's wavefronts to the SIMDs round robin from SIMD 0 -- 2 + 2 + 1 + 1, twice = 4 on SIMD 0 (tools/cores_probe.hip:
// 512 six-wavefront workgroups of 168 registers take 1.5 T, not T; profiles/r06u_cores_probe_168.log; the tick itself, 8 192 states 23.9 -> 26.2 us:
// profiles/r06u_ab_fused_two_per_cu_by_launch_bounds_not_kept.log).
// Twelve wavefronts of ONE workgroup land 3 + 3 + 3 + 3: wavefronts 0 .. 3 / 4 .. 7 the QPs of the first / second 16 states, 8 / 9 their rnea roles, 10 / 11 their mass_jac
// roles -- every SIMD holds two QP wavefronts and one role.  The bodies are those of fused_tick_kernel, untouched: they address state blockIdx.x * 16 + slot, so the second
// half works on the batch's upper half through argument pointers advanced by that many states (scalar registers: `half` is wave-uniform); its QP wavefronts are threads
// 256 .. 511, whose slots 16 .. 31 are folded into the same shift (and into the workspace pointer).  Every lane is live and the component stride N is the same for both
// halves; a batch that is not a multiple of 32 gets one more workgroup anchored at its end (below).
// The price in fp64: the rnea and mass_jac roles spill (52 / 36 dwords per lane at 168 registers; 22 MB of scratch traffic per launch at 8 192 states by PMC) and the halves share
// issue slots -- a pair lasts 18.3 us where a lone workgroup lasts 13.3 (fp32, which does not spill: 15.3 against 12.4) -- so the host runs this form only where it saves a
// round: fp64 4 225 ... 8 192 states (6 144: 23.9 -> 20.0 us, 257 -> 307 M steps/s; 8 192: 24.0 -> 20.4 us, 342 -> 402 M), fp32 4 225 ... 16 384 (8 192: 364 -> 457 M);
// profiles/r06v_ab_fused_pair.log, r06y3_ab_fused_pair_ragged.log, r06zzz_ab_pair_f32.log.  Measured on top and not kept (profiles/r06v_pair_variants.log,
// r06y5_pair_knock_and_layout_not_kept.log): issue priorities, the bias-force recursion behind the mass_jac role, the roles of a half on one SIMD.
// The role bodies park joint transforms and forces in static LDS arrays [word][BLOCK] indexed by the thread within BLOCK: the pair instantiates them with BLOCK = 128, so
// that the role wavefronts of the two halves (threads 512 .. 639 and 640 .. 767: 0 .. 63 and 64 .. 127 within 128) own disjoint columns.
template <class P> WBC_DEV void shift_ptr(P*& p, int off) { if (p) p += off; }
template <class T> WBC_DEV void shift_states(SweepArgs<T>& a, int off) {
  shift_ptr(a.q, off); shift_ptr(a.v, off); shift_ptr(a.M, off); shift_ptr(a.h, off); shift_ptr(a.Jc, off); shift_ptr(a.pf, off); shift_ptr(a.p, off); shift_ptr(a.beta, off);
  shift_ptr(a.w_des, off); shift_ptr(a.vdot_des, off); shift_ptr(a.tau_prev, off); shift_ptr(a.f_prev, off); shift_ptr(a.obs_integ, off); shift_ptr(a.obs_r, off); shift_ptr(a.ws, off);
}
template <class T> WBC_DEV void shift_states(QpArgs<T>& a, int off) {
  shift_ptr(a.ws, off); shift_ptr(a.normals, off); shift_ptr(a.mu, off); shift_ptr(a.mask, off); shift_ptr(a.Jc, off); shift_ptr(a.wdes, off);
  shift_ptr(a.tau, off); shift_ptr(a.f, off); shift_ptr(a.status, off); shift_ptr(a.iters, off); shift_ptr(a.aset_in, off); shift_ptr(a.aset_out, off); shift_ptr(a.rprev, off);
}
constexpr int FUSED_PAIR_THREADS = 768;
template <class T>
__global__ __launch_bounds__(FUSED_PAIR_THREADS) void fused_pair_kernel(const DevModel<T>* __restrict__ model, DevParams<T> prm, SweepArgs<T> a, QpArgs<T> qa, QpJidx jmap) {
  __shared__ __attribute__((aligned(512))) T cst[CST_WORDS];   // one constant table for both halves (first in LDS: see fused_tick_kernel)
  __shared__ int zidx_s[64];
  __shared__ T wsl2[2 * WS_LDS_WORDS * 16];
  __shared__ int flags[2][4];   // per half: rnea role done / its lever arms, w_des are out / (observer: unused)
  const int wave = (int)(threadIdx.x >> 6);
  const int half = __builtin_amdgcn_readfirstlane(wave < 8 ? (wave >> 2) : (wave & 1));
  // Workgroups 0 .. N / 32 - 1: the lower half of the first 32 (N / 32) states on the first half's wavefronts, the upper half on the second's.  N not a multiple of 32: one
  // more workgroup, anchored at the END of the batch -- states N - 32 .. N - 17 and N - 16 .. N - 1.  It recomputes up to 31 states a regular workgroup also owns and stores
  // the same bits (same bodies, same inputs, no observer state to advance): every lane of every workgroup is live.  N >= 64: the bodies test their UNSHIFTED state index
  // against N -- the role wavefronts blockIdx.x * 16 + slot <= 16 pairs + 15, the QP wavefronts of a second half 16 pairs + 31 at most, both < N from two pairs on.
  const int pairs = (int)(a.N >> 5);
  const bool tail = (int)blockIdx.x == pairs;
  const int off = tail ? (int)a.N - 32 + 16 * half - 16 * pairs : (half ? 16 * pairs : 0);   // what the bodies' blockIdx.x * 16 + slot lacks
  T* const wsl = wsl2 + half * (WS_LDS_WORDS * 16);
  int* const ready = &flags[half][0];
  int* const gready = &flags[half][1];
  int* const oready = &flags[half][2];
  if (wave >= 8) {
    shift_states(a, off);
    if (wave < 10) {
      auto geom_out = [=] __device__() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(gready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      rnea_step_body<T, RS_STEP | RS_H, 128, 2>(model, prm, a, cst, wsl, NoWait(), geom_out);   // (128: the role's parking lot has a half per role wavefront -- threads 512 .. 639, 640 .. 767)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      mass_jac_body<T, 128, 2, 16, (0)>(model, a, cst, zidx_s);
    }
  } else {
    for (int i = threadIdx.x; i < CST_WORDS; i += 512) cst[i] = model->cst[i];
    if (threadIdx.x < 64) zidx_s[threadIdx.x] = model->zidx[threadIdx.x];
    if (threadIdx.x < 8) (&flags[0][0])[threadIdx.x] = 0;
    __syncthreads();
    shift_states(a, off);
    shift_states(qa, half ? off - 16 : off);   // (QP slots 16 .. 31 of threads 256 .. 511: see above; `off` >= 16 for every second half)
#ifdef WBC_FUSED_STAMP
    QpSync sy{gready, oready, ready, 1, 2, 1, 1, nullptr, 0};
#else
    QpSync sy{gready, oready, ready, 1, 2, 1, 1};
#endif
    const int* const zs = zidx_s;
    const unsigned tq = threadIdx.x & 255u;
    auto idle = [=] __device__() { if (!a.skip_consts) structural_consts_quarter<T>(model, a, zs, tq); };
    qp_body<T, true, false, 16, false, 8, decltype(idle), false, 0>(prm, qa, jmap, wsl - (half ? 16 : 0), &sy, QpWho{0, false}, idle);
  }
}

// Persistent rollout (BASELINE.json configs[4]): `horizon` dependent ticks of {tick roles as above, forward
// dynamics + integrator} in ONE launch.  A workgroup owns its 4 or 16 states for the whole horizon, so no tick boundary ever
// leaves the CU: no launch, no HBM round trip of the workspace.
// Layout (round 5; DESIGN.md 4.7 -- the layouts of rounds 1-4, with an integrator wavefront of its own and two barriers per tick, and every placement
// tried on the way are in docs/DESIGN_R04.md 8.0a and profiles/r05*_ab_rollout_*.log; their code went in round 6): the integrator's factorisation
// (phase 1) runs on the mass_jac wavefront behind its LDS image, its right-hand sides / solves / state update (phase 2) on QP wavefront 0 right behind
// the torque map; ONE barrier per tick; the states, this tick's tau / f / h, the planner's references and plans live in LDS images from tick to tick,
// and the caller's buffers are written in the launch's last tick only.
//   SPW = 4 (up to 1 024 rollouts: every CU gets a workgroup and a tick waits for the slowest of 4 QPs): FOUR wavefronts, one per SIMD --
//     wavefront 0  [planner,] QP of the 4 states, then phase 2        wavefront 1  rnea role, its two force recursions side by side in the lanes (RS_LANE2)
//     wavefront 2  mass_jac role, then phase 1 on its own image       wavefront 3  observer, ONE pass over both sets of rows (rhat_base handed to the QP first)
//   SPW = 16: QP x 4 (phase 2 on wavefront 0 behind all four), rnea (tau_partial handed to the QPs before the base rows of h), mass_jac + phase 1,
//     observer base rows and, on a wavefront of their own, joint rows: six / eight wavefronts, two per SIMD.
// tau_prev / f_prev of the observer are the result image's rows of the previous tick (first tick: the caller's buffers).
// TRACK: the CoM planner in the loop (wbc_rollout_tracking_batch): QP wavefront 0 first runs the reference generator (com_reference_body: w_des, vdot_des
// of this tick -> LDS image, optional CoM record) and raises a flag; the rnea role issues its state loads, then waits for that flag.
// WARM: every tick after the first starts its QPs from the previous tick's active set, carried in an LDS word per state (wbc_solver_options.rollout_warm;
// the QP body of these instantiations is the block set-up of qp_struct16.hip.hpp for EVERY tick -- tick 0 from the empty set, or from qa.aset_in when
// the caller continues an earlier rollout).
__host__ __device__ constexpr int rollout_threads(bool observer, int spw) {
  return (spw == 4) ? (256)
       : (spw == 16) ? (observer ? 512 : 384)   // QP x 4, rnea, mass_jac [, observer base rows, joint rows]: no integrator wavefront
       : (observer ? 512 : 448);
}
template <class T, bool OBSERVER, bool TRACK, int SPW = 16, bool WARM = false>
__global__ __launch_bounds__(rollout_threads(OBSERVER, SPW), 1) void rollout_kernel(const DevModel<T>* __restrict__ model, DevParams<T> prm,
                                                                         SweepArgs<T> a, QpArgs<T> qa, QpJidx jmap, IntegrateArgs<T> ia,
                                                                         int horizon, const DevRefParams<T>* __restrict__ G, RefArgs<T> ra) {
  __shared__ __attribute__((aligned(512))) T cst[CST_WORDS];   // (the alignment puts the table FIRST in the workgroup's LDS: within reach of the 16-bit ds_read offset, see dyn_sweep.hip.hpp)
  __shared__ int zidx_s[64];
  __shared__ T wsl[WS_LDS_WORDS * 16];
  __shared__ int ready, gready, oready, mready, rready, fready, qdone, hready;   // (hready, TAUP_FIRST: the rnea role's h is complete in the result image)   // (qdone: QP wavefronts whose tau, f of this tick are in the result image)
  __shared__ int rpack;   // (QpSync::rp_ack: QP wavefronts that have read r_prev in this tick, counted over the ticks)
  // (four wavefronts, one per SIMD: each may use the SIMD's whole register file -- 512 with the accumulation registers; an eight-wavefront form sat at 256 and spilled)
  static_assert(SPW == 4 || SPW == 16, "4 or 16 states per workgroup");
  constexpr int REXT = 3;   // the roles' EXT: 3 = states from the LDS image (dyn_split.hip.hpp, WBC_STATE_MACROS)
  // 4 states: the observer wavefront runs the WHOLE update in one pass (PART 0: both sets of rows share the sweeps; as two passes the joint rows arrived behind
  // the rnea role: 11.8 against 11.1 us per tick, profiles/r05o_ab_rollout_merge.log)
  constexpr bool OBS_ONE = SPW == 4;
  constexpr bool OBS_FIFTH = OBSERVER && FUSED_OBS_WAVES == 2 && (SPW == 16);   // joint rows on their own wavefront
  constexpr int W_JOINT = SPW == 16 ? 7 : 4;
  // the fp64 rnea role does not propagate the own-leg Jacobian blocks -- the torque map takes them from the mass_jac role's image (RS_NOJC): 9.33 -> 9.23 us per
  // tick at 1 024 robots; fp32 7.57 -> 7.62 and cold 15.75 -> 15.86, hence fp64 only (profiles/r05s_ab_rollout_nojc_refimg.log)
  constexpr bool NOJC = sizeof(T) == 8;
  constexpr bool SPEC_ORDER = OBSERVER && !WARM;
  constexpr int QP_WAVES = SPW / 4;
  for (int i = threadIdx.x; i < CST_WORDS; i += blockDim.x) cst[i] = model->cst[i];
  if (threadIdx.x < 64) zidx_s[threadIdx.x] = model->zidx[threadIdx.x];
  if (threadIdx.x == 0) { ready = 0; gready = 0; oready = 0; mready = 0; rready = 0; fready = 0; qdone = 0; hready = 0; rpack = 0; }
  __syncthreads();
  const int wave = (int)(threadIdx.x >> 6);
  // -DWBC_RO_PRIO=1: the rnea role -- the chain a rollout tick waits for (tools/ro_knock.sh) -- at a higher issue priority than QP wavefront 0, its SIMD-mate
  constexpr int W_RNEA = (SPW == 4) ? 1 : 4, W_MJ = (SPW == 4) ? 2 : 5, W_OBS = (SPW == 4) ? 3 : 6;
  // (where the idle QP wavefronts of a 4-state workgroup were tried as hosts of the observer's joint rows, the bias-force recursion and the integrator: docs/DESIGN_R04.md 8.0a,
  //  DESIGN.md 4.7 -- all measured slower than the four-wavefront layout below and removed in round 6; the A/B logs are profiles/r05b_ab_rollout_*.log)
  constexpr int PLAN_WAVE = (TRACK) ? 0 : ((TRACK && SPW == 4) ? 3 : -1);   // (in front of the QP, whose first input -- the lever arms -- the rnea role
                                                                                         // publishes only after it has waited for these references)
  T* const traj0 = ia.tau_traj;
  T* const com0 = ra.com;
  // (WARM) the active set of each of the workgroup's states, from tick to tick: one LDS word per state, read and written by the state's own
  // QP row only (a register of the QP wavefronts would be live through every role's code of this 256-register kernel)
  __shared__ int aset_sh[16];
  // what the integrator's factorisation needs of M and Jc, handed over by the mass_jac role in LDS (dyn_split.hip.hpp): with the QP warm-started
  // the tick's barrier waits for that factorisation, not for the QP, and its operands should neither wait for the role's stores to drain
  // nor come back through L2
  __shared__ T mj_hand[MJ_HAND_WORDS * 64];
  // 1: the factorisation (phase 1 of the integrator) runs on the mass_jac wavefront, right behind its image; the integrator wavefront runs the observer's
  // joint rows, waits at the tick barrier and does phase 2 with the factors from an LDS image.  In the stamp build the tick's barrier moves from +10.3 to
  // +9.2 us (profiles/r05g_rollout_timeline_spw4.txt); WITHOUT stamps the tick gets slower -- 12.5 -> 13.1 us at 1 024 robots, fp32 10.6 -> 11.2
  // (profiles/r05g_ab_rollout_*.log): measured, not kept.  0 (default): phase 1 on the integrator wavefront behind the joint rows
// the integrator's state stores without their `if (live)` (integrate.hip.hpp, UNGUARD)
  // (round 5) this tick's tau, f (QP wavefronts) and h (rnea role) for the integrator ALSO in LDS, the tick's first barrier ordering LDS only (the global
  // stores drain until barrier B) and phase 2 reading them there instead of through L2.  Measured (profiles/r05f_ab_rollout_reslds_*.log, us per tick at
  // 1 024 robots, off -> on): fp32 10.57 -> 10.01, fp64 12.51 -> 12.80 (128 robots 12.30 -> 12.67, planner in the loop 15.15 -> 15.6) -- round 4 had seen the
  // same sign for fp64.  1 (default): fp32 kernels only; 0: never; 2: both scalar types (A/B)
  constexpr bool RES_LDS = true || (sizeof(T) == 4);
  __shared__ T fact_sh[true ? INT_FACT_WORDS * 64 : 1];   // the integrator's phase 1 -> phase 2 hand-over (integrate.hip.hpp, PHASE)
  __shared__ T st_sh[SIMG_WORDS * 16];             // the workgroup's states
  __shared__ T ref_sh[(TRACK) ? 24 * 16 : 1];         // (planner in the loop) this tick's references: planner role -> rnea role
  __shared__ T plan_sh[(TRACK) ? PLAN_WORDS * 16 : 1];   // ... and the plans of the workgroup's states
  {
    for (int i = threadIdx.x; i < SIMG_WORDS * 16; i += blockDim.x) {
      const int comp = i >> 4, slot = i & 15;
      size_t st = (size_t)blockIdx.x * SPW + (slot < SPW ? slot : 0);
      st = st < a.N ? st : a.N - 1;
      st_sh[i] = comp < SIMG_V ? a.q[(size_t)comp * a.N + st] : a.v[(size_t)(comp - SIMG_V) * a.N + st];
    }
    if constexpr (TRACK) {
      for (int i = threadIdx.x; i < PLAN_WORDS * 16; i += blockDim.x) {
        const int comp = i >> 4, slot = i & 15;
        size_t st = (size_t)blockIdx.x * SPW + (slot < SPW ? slot : 0);
        st = st < a.N ? st : a.N - 1;
        plan_sh[i] = ra.plan[(size_t)comp * a.N + st];
      }
    }
    if constexpr (!RES_LDS) __syncthreads();
  }
  __shared__ T res_sh[RES_LDS ? (RES_WORDS + 18) * 16 : 1];   // (+ 18 rows: the external torques of the workgroup's states, parked once)
  T* const res_img = RES_LDS ? res_sh : nullptr;
  if constexpr (RES_LDS) {
    for (int i = threadIdx.x; i < 18 * 16; i += blockDim.x) {
      const int comp = i >> 4, slot = i & 15;
      size_t st = (size_t)blockIdx.x * SPW + (slot < SPW ? slot : 0);
      st = st < a.N ? st : a.N - 1;
      res_sh[(RES_WORDS + comp) * 16 + slot] = ia.tau_ext ? ia.tau_ext[(size_t)comp * a.N + st] : (T)0;
    }
    if constexpr (OBSERVER) {   // tau_prev, f_prev of the first tick: the caller's; of every later tick: what the QP left in these rows
      for (int i = threadIdx.x; i < 24 * 16; i += blockDim.x) {
        const int comp = i >> 4, slot = i & 15;
        size_t st = (size_t)blockIdx.x * SPW + (slot < SPW ? slot : 0);
        st = st < a.N ? st : a.N - 1;
        res_sh[i] = comp < 12 ? (a.tau_prev ? a.tau_prev[(size_t)comp * a.N + st] : (T)0) : (a.f_prev ? a.f_prev[(size_t)(comp - 12) * a.N + st] : (T)0);
      }
    }
    __syncthreads();
  }
  auto barrier_A = [] __device__() {
    if constexpr (RES_LDS) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // my LDS writes (lgkmcnt only): the global stores keep draining
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    } else __syncthreads();
  };
  if constexpr (WARM) {
    if (threadIdx.x < 16) {
      const size_t sq = (size_t)blockIdx.x * SPW + threadIdx.x;
      aset_sh[threadIdx.x] = (qa.aset_in && sq < a.N && (int)threadIdx.x < SPW) ? qa.aset_in[sq] : 0;
    }
    __syncthreads();
  }
  for (int t = 0; t < horizon; ++t) {
    // The batch size is laundered through an empty asm once per tick: every per-lane address in the role bodies derives
    // from it, so none of that (tick-invariant) address arithmetic is hoisted out of the horizon loop -- hoisted, it
    // occupied ~250 registers for the whole kernel and spilled 1-2 kB per lane.
    unsigned long long n_tick = a.N;
    asm volatile("" : "+s"(n_tick) : : "memory");
    SweepArgs<T> at = a;
    QpArgs<T> qat = qa;
    IntegrateArgs<T> iat = ia;
    at.N = qat.N = iat.N = (size_t)n_tick;
    at.simg = st_sh; iat.simg = st_sh; at.resimg = res_sh; at.refimg = ref_sh;
// 0: every tick stores its q, v (A/B)
    iat.skip_state = (1 && t < horizon - 1) ? 1 : 0;   // q, v of the LAST tick are what the caller finds (the roles read the LDS image)
// 0: every tick stores its M / Jc / pf (A/B)
    at.skip_mats = (1 && t < horizon - 1) ? 1 : 0;   // M, Jc, pf of the LAST tick are what the caller finds in its buffers (as with per-tick launches)
    if (t > 0) at.skip_consts = 1;   // the structural zeros / ones of M, Jc were written by tick 0 of THIS launch into the same buffers (the mass_jac role's
                                     // ~55 store instructions per tick sit in front of the integrator's factorisation: wbc_api.cpp, rollout_persistent)
#ifdef WBC_FUSED_STAMP   // diagnostic build: the last tick's role timestamps go out through the pf output
    double* const rstamp = (t == horizon - 1) ? (double*)a.pf : nullptr;
    const unsigned rstampN = (unsigned)n_tick;
    at.pf = nullptr;
#define RSTAMP(slot) do { if (rstamp) WBC_FSTAMP_S(rstamp, rstampN, slot, SPW); } while (0)
    if (wave == 0) RSTAMP(0);
#else
#define RSTAMP(slot) do {} while (0)
#endif
    auto planner_role = [&]() __attribute__((always_inline)) {   // this tick's references
      if constexpr (TRACK) {
        RefArgs<T> rt = ra;
        rt.N = (size_t)n_tick;
        rt.simg = st_sh; rt.refimg = ref_sh; rt.planimg = plan_sh;
        rt.skip_out = (t < horizon - 1) ? 1 : 0;   // (the caller finds the LAST tick's references in its w_des / vdot_des buffers)
        rt.t = (T)t * prm.dt + ra.t;
        rt.com = com0 ? com0 + (size_t)t * 6 * (size_t)n_tick : nullptr;
        com_reference_body<T, true, SPW, true>(model, G, rt, cst);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // w_des, vdot_des are in L2 ...
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&rready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ... then the flag
      }
    };
    auto joint_rows_role = [&]() __attribute__((always_inline)) {
      if constexpr (OBSERVER && FUSED_OBS_WAVES == 2) {
        // the JOINT rows of the observer update (rhat_joint, which the QP needs only in its torque map); wave 6 is left with the base rows,
        // whose rhat_base the QP's b waits for
        observer_body<T, 64, REXT, 2, SPW>(model, prm, at, cst, wsl);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    };
    if (wave == W_RNEA) {
      int* const rflag = &rready;
      const int rneed = t + 1;
      int* const gflag = &gready;
      // (4-state workgroups: the bias and the acceleration recursion side by side in the lanes -- RS_LANE2, device_types.hpp; -DWBC_RO_LANE2=0: one after the other)
      constexpr int RNEA_MODE = (SPW == 4 ? (RS_STEP | RS_H | RS_LANE2) : (RS_STEP | RS_H)) | (NOJC ? RS_NOJC : 0) | ((TRACK) ? RS_REFIMG : 0);
      auto wait_refs = [rflag, rneed] __device__() {
        if constexpr (TRACK) {
          while (__hip_atomic_load(rflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < rneed) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
      };
      auto geom_out = [gflag] __device__() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(gflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      // (16-state workgroups) tau_partial is handed to the QP BEFORE the base rows of h are summed, rotated and written: only phase 2 of the integrator,
      // behind the torque map, needs those -- it waits for `hready`.  Measured (profiles/r05zz_ab_rollout_taup_first.log): 2 048 rollouts 12.33 -> 12.11 us
      // per tick; the 4-state workgroups LOSE with it (8.84 -> 8.96, planner in the loop 11.39 -> 11.51) and keep the one flag behind the whole body
      // (-DWBC_RO_TAUP_FIRST=2: both; 0: neither)
      constexpr bool TAUP_FIRST = ((SPW == 16));
      if constexpr (TAUP_FIRST) {
        int* const finflag = &ready;
        auto taup_out = [finflag] __device__() {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
          if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(finflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        rnea_step_body<T, RNEA_MODE, 64, REXT, SPW>(model, prm, at, cst, wsl, wait_refs, geom_out, res_img ? res_img + RES_H * 16 : nullptr, taup_out);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&hready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
      rnea_step_body<T, RNEA_MODE, 64, REXT, SPW>(model, prm, at, cst, wsl, wait_refs, geom_out, res_img ? res_img + RES_H * 16 : nullptr);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      RSTAMP(3);   // (WBC_RO_STAMP_ALT) rnea: done
    } else if (wave == W_MJ) {
      int* const mflag = &mready;
      int* const fflag = &fready;
      T* const factp = fact_sh;
      T* const handp = mj_hand;
      const IntegrateArgs<T> ia1 = iat;
      auto publish = [=] __device__() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // the hand-over image is in LDS ...
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(mflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ... then the flag
        RSTAMP(1);   // (WBC_RO_STAMP_ALT) mass_jac: image published
        // phase 1 of the integrator, on my own image (my own LDS words: program order of one lane); the factors are complete when this wavefront
        // reaches the tick barrier, behind which the integrator wavefront reads them
        integrate_body<T, SPW, IntegrateNoWait, 1, true, true, false, IntegrateNoWait, true>(model, ia1, IntegrateNoWait(), handp, nullptr, factp);
           // the factors are in LDS: wavefront 0 may start phase 2
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
          if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(fflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        
      };
      // (the M / Jc / pf stores to HBM come BEHIND the flag, from the image, and only in the launch's last tick: nothing in this kernel reads them)
      mass_jac_body<T, 64, REXT, SPW, true, decltype(publish)>(model, at, cst, zidx_s, mj_hand, publish);
    } else if (OBSERVER && wave == W_OBS) {
      if constexpr (OBSERVER) {
        int* const ack = &rpack;
        const int ack_need = QP_WAVES * (t + 1);
        const bool ack_on = qa.rprev != nullptr;   // (null: the QP waits for rhat and reads nothing this role writes)
        auto wait_ack = [ack, ack_need, ack_on] __device__() {
          if constexpr (SPEC_ORDER) { if (ack_on) { while (__hip_atomic_load(ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < ack_need) __builtin_amdgcn_s_sleep(1); } }
        };
        if constexpr (OBS_ONE) {   // one pass over both sets of rows; rhat_base is handed to the QP as soon as it exists, the joint rows count as a finisher
          int* const oflag = &oready;
          int* const jflag = &ready;
          auto rows_out = [oflag, jflag] __device__(int stage) {   // 0: rhat_base is in the image, 1: rhat_joint is
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(stage == 0 ? oflag : jflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          };
          observer_body<T, 64, REXT, 0, SPW, decltype(wait_ack), decltype(rows_out)>(model, prm, at, cst, wsl, wait_ack, rows_out);
          RSTAMP(10);
        } else {
        if constexpr (FUSED_OBS_WAVES == 2) observer_body<T, 64, REXT, 1, SPW, decltype(wait_ack)>(model, prm, at, cst, wsl, wait_ack);   // base rows
        else observer_body<T, 64, REXT, 0, SPW, decltype(wait_ack)>(model, prm, at, cst, wsl, wait_ack);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&oready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        RSTAMP(10);
        }
        if constexpr (FUSED_OBS_WAVES == 2 && !OBS_ONE && !OBS_FIFTH) joint_rows_role();   // (-DWBC_RO_MERGE_OBS=2) ... then the joint rows, which the torque map needs ~3 us later
      }
    } else if (OBS_FIFTH && wave == W_JOINT) {
      joint_rows_role();
    } else {
      constexpr int NFIN = (OBSERVER && (FUSED_OBS_WAVES == 2 || OBS_ONE)) ? 2 : 1;   // rnea role (+ the observer's joint rows, run by the integrator wavefront)
#ifdef WBC_FUSED_STAMP
      QpSync sy{&gready, &oready, &ready, 2 * t + 1, 2 * t + 2, t + 1, NFIN * (t + 1), rstamp, rstampN};
#else
      QpSync sy{&gready, &oready, &ready, 2 * t + 1, 2 * t + 2, t + 1, NFIN * (t + 1)};
#endif
      if constexpr (SPEC_ORDER) { if (qa.rprev) sy.rp_ack = &rpack; }
      sy.res = res_img;
// 0: every tick stores its tau, f, status, iters (A/B)
      if constexpr (NOJC) { sy.hand = mj_hand; sy.hand_flag = &mready; sy.need_hand = t + 1; }
      sy.skip_out = 1 && t < horizon - 1;   // (the LAST tick's are what the caller finds, as with per-tick launches)
      if constexpr (PLAN_WAVE >= 0) { if (wave == PLAN_WAVE) planner_role(); }
      if constexpr (WARM) {
        qat.aset_out = (t == horizon - 1) ? qa.aset_out : nullptr;   // the set goes out once, behind the last tick
        if (wave * 4 < SPW) qp_body<T, true, OBSERVER, SPW, false, 4, QpNoIdle, false, 2>(prm, qat, jmap, wsl, &sy, QpWho{0, false}, QpNoIdle(), &aset_sh[(threadIdx.x & 255) >> 4]);
      } else
      if (wave * 4 < SPW) qp_body<T, true, OBSERVER, SPW>(prm, qat, jmap, wsl, &sy);   // (SPW = 4: QP wavefront 0 only)
      if constexpr (QP_WAVES > 1) {   // (16 states: four QP wavefronts fill the result image; phase 2 runs behind all of them)
        if (wave * 4 < SPW) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
          if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&qdone, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
              if (wave == 0) {   // phase 2 of the integrator, on the wavefront that has just written tau and f to the LDS image (its own LDS traffic: program order)
          if constexpr (QP_WAVES > 1) { while (__hip_atomic_load(&qdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < QP_WAVES * (t + 1)) __builtin_amdgcn_s_sleep(1); }
          while (__hip_atomic_load(&fready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < t + 1) __builtin_amdgcn_s_sleep(1);   // M's blocks and the factors
          if constexpr (((SPW == 16))) { while (__hip_atomic_load(&hready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < t + 1) __builtin_amdgcn_s_sleep(1); }   // h
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
          iat.tau_traj = traj0 ? traj0 + (size_t)t * 12 * (size_t)n_tick : nullptr;
#ifdef WBC_FUSED_STAMP
          iat.istamp = rstamp; iat.istampN = rstampN;
          RSTAMP(7);   // phase 2 starts (the factors are there)
#endif
// 1: the tick's barrier in FRONT of this wavefront's stores (integrate.hip.hpp, after_state: measured, not kept)
                      integrate_body<T, SPW, IntegrateNoWait, 2, true, true, true, IntegrateNoWait, true>(model, iat, IntegrateNoWait(), mj_hand, res_img, fact_sh);
            RSTAMP(8);
          
        }
      
    }
    __syncthreads(); continue;   // the tick's only barrier: the new state (LDS image), tau, f (memory: the next tick's observer reads them)
    barrier_A();       // barrier A: tau, f (waves 0..3), h (wave 4) are visible to the integrator (round 5: in LDS)
    __syncthreads();   // barrier B: q, v of the next tick -- and this tick's tau, f, h in memory (the next tick's observer role reads tau, f as tau_prev, f_prev)
  }
}

}  // namespace wbc

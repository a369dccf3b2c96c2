// tile_tick_kernel (round 6): the WHOLE tick of a batch beyond the fused size as ONE launch of big workgroups -- one per CU in a round.
// A workgroup owns `tile` = 16 W NS consecutive states.  Its first NS wavefronts run the observer-free dynamics sweep (dyn_sweep_body: M, h, Jc, the
// step workspace); with the observer on the next NS run the momentum-observer update (observer_park_body: the new observer state, rhat -> workspace)
// -- the two roles of sweep_obs_kernel (observer.hip.hpp), here side by side in one workgroup.  Behind ONE barrier the workgroup's NWQ wavefronts (the role
// wavefronts plus, for small tiles, helpers that idle through the roles) are the staged QP tile of exactly those states (qp_stile_body,
// qp_kernels.hip.hpp): the image is read back from memory this CU has just written (L2), predictor, finisher, groups, results row by row.
// Why (configs[3]'s shard, 32 768 fp32 states, observer on): as two launches the tick is sweep_obs 19.4 us + staged tiles 17.3 us by the dispatch events,
// of which 16.5 and 15.2 us lie between the first workgroup's entry and the last one's exit (tools/so_stamp.py, tools/tile_stamp.py) -- every launch
// pays its ramp and the drain of its slowest workgroup with the rest of the device idle, and the second one a cold read of what the first wrote.
// One launch pays them once; a CU goes on to its QPs the moment ITS states are through the sweep.
// The roles' LDS (ONE constant table for the workgroup, per-wavefront parking lots) and the QP stage's (image, solver tables) are overlaid: one dynamic block.
// Behind both lie the hand-over rows: tau_partial (sweep wavefronts) and rhat (observer wavefronts) go to the QP stage's stage-in through LDS, not through the
// memory workspace -- the roles end without those stores, the stage-in starts without waiting for them (32 768 fp32 states: 30.8 -> 30.0 us per tick, 262 144:
// 243 -> 229 us; fp64 16 384 observer off: 34.8 -> 32.5 us; HBM traffic by PMC 95.4 -> 88.6 MB per tick: profiles/r06j_ab_tile_tick_lds_handover.log).
#pragma once
#include "dyn_sweep.hip.hpp"
#include "observer.hip.hpp"
#include "qp_kernels.hip.hpp"

namespace wbc {

// LDS of a workgroup: [constant table, shared by the roles | NS sweep objects | NS observer objects] overlaid with the QP stage's block, and BEHIND both the
// hand-over rows (tau_partial from the sweep wavefronts, rhat from the observer ones): written by the roles, read by the QP stage's stage-in.
template <class T> constexpr size_t tile_tick_cst_bytes() { return (sizeof(T) * CST_WORDS + 15) / 16 * 16; }
template <class T, int W, int NS, int NWQ, bool OBS>
constexpr size_t tile_tick_hand_off() {
  constexpr int MODE = SW_MATS | SW_STEP | SW_NOB;
  constexpr size_t roles = tile_tick_cst_bytes<T>() + (size_t)NS * (sizeof(SweepLds<T, MODE, 64, W, true>) + (OBS ? sizeof(ObsLds<T, 64, W, true>) : 0));
  constexpr size_t qp = stile_lds_bytes(16 * W * NS, sizeof(T), NWQ);
  return ((roles > qp ? roles : qp) + 15) / 16 * 16;
}
template <class T, int W, int NS, int NWQ, bool OBS>
constexpr size_t tile_tick_lds_bytes() {
  constexpr int MODE = SW_MATS | SW_STEP | SW_NOB;
  constexpr size_t roles = tile_tick_cst_bytes<T>() + (size_t)NS * (sizeof(SweepLds<T, MODE, 64, W, true>) + (OBS ? sizeof(ObsLds<T, 64, W, true>) : 0));
  constexpr size_t qp = stile_lds_bytes(16 * W * NS, sizeof(T), NWQ);
  return tile_tick_hand_off<T, W, NS, NWQ, OBS>() + (size_t)(OBS ? HAND_ROWS : HAND_RHAT) * (16 * W * NS) * sizeof(T);   // (observer off: the tau_partial rows only)
}

template <class T, int W, int NS, int NWQ, bool OBS>
__global__ __launch_bounds__(64 * NWQ, 2) void tile_tick_kernel(const DevModel<T>* __restrict__ model, DevParams<T> prm, SweepArgs<T> a, QpArgs<T> qa, QpJidx jmap) {
  constexpr int MODE = SW_MATS | SW_STEP | SW_NOB;
  constexpr int TILE = 16 * W * NS, CH = (TILE + 63) / 64, NR = OBS ? 2 * NS : NS;
  static_assert(NWQ >= NR && 64 * NWQ >= 4 * TILE, "every role wavefront joins the QP stage; the predictor needs one thread per foot and state");
  using SwL = SweepLds<T, MODE, 64, W, true>;
  using ObL = ObsLds<T, 64, W, true>;
  extern __shared__ __attribute__((aligned(16))) unsigned char tt_dyn[];
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  constexpr size_t CSTB = tile_tick_cst_bytes<T>();
  T* const hand = (T*)(tt_dyn + tile_tick_hand_off<T, W, NS, NWQ, OBS>());
  RoleShare<T> xr;
  xr.cst = (T*)tt_dyn; xr.hand = hand; xr.hs = TILE; xr.stage_threads = 64 * NR;
  if (wave < (unsigned)NS) {
    xr.col0 = (int)wave * 16 * W;
    dyn_sweep_body<T, MODE, 64, W, true>(model, prm, a, *(SwL*)(tt_dyn + CSTB + (size_t)wave * sizeof(SwL)), blockIdx.x * NS + wave, xr);
  } else if (OBS && wave < (unsigned)NR) {
    xr.col0 = (int)(wave - NS) * 16 * W;
    observer_park_body<T, 64, W, true>(model, prm, a, *(ObL*)(tt_dyn + CSTB + (size_t)NS * sizeof(SwL) + (size_t)(wave - NS) * sizeof(ObL)), blockIdx.x * NS + (wave - NS), xr);
  } else __syncthreads();   // (helper wavefronts: the one barrier every role body has behind its table staging)
  // What the roles stored (workspace, Jc) is read by OTHER wavefronts of this workgroup below.  __syncthreads() is a workgroup-scope release / acquire: the
  // stores have left the wavefronts, and the wavefronts of a workgroup share their CU's vector L1 (write-through), so they see them.  (An AGENT-scope
  // release here writes back the XCD's whole L2 -- 256 times per launch: the tick took 79 us instead of 37.)
  __syncthreads();
  qp_stile_body<T, OBS, NWQ, CH, true, true>(prm, qa, jmap, TILE, blockIdx.x, tt_dyn, wave, hand);
}

}  // namespace wbc

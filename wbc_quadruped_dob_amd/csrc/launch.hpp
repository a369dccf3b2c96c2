// Host-callable launchers of the gfx950 kernels.  Each kernel family is its own translation unit (k_*.hip), compiled
// once per scalar type (-DWBC_SCALAR=double|float), so the library builds in parallel and a change to one kernel
// recompiles only the units that contain it.  The host side (wbc_api.hip, wbc_multi.cpp) sees nothing but these
// declarations and the plain-data argument structs of device_types.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"
#include "qp_general_args.hpp"

namespace wbc {

// Batch sizes at which a launcher changes kernel variant.  They live here so that the host side's tick planner (wbc_plan_tick,
// wbc_api.cpp) reports exactly what the launchers do.
// 256-thread workgroups (one constant table for four wavefronts) from two full rounds of 8 waves per CU on
constexpr size_t BIG_GRID_THREADS = (size_t)256 * 8 * 64 * 2;
// fp32 sweep, even N: two states per lane as packed pairs.  Below this the batch does not fill the SIMDs with one state per lane
// either, and the shorter dependent chain per state of the unpacked form wins
constexpr long long WBC_PACK2_MIN_STATES = 32768;
// fp32 tiles of 64 ... 128 states from this batch size on run the leaner dense fp32 body (four workgroups per CU)
constexpr long long WBC_F32_DENSE_TILE_MIN = 65537;

// staged tiles (qp_stile_kernel, fp32 solvers): one workgroup of twelve wavefronts per CU holds ceil(N / 256) states, up to 192 (three 64-column chunks:
// image 80 kB + twelve wavefronts' solver tables 70 kB of the CU's 160 kB of LDS) -- i.e. one round of workgroups up to 49 152 states
constexpr int STILE_MAX_TILE = 192;
constexpr size_t STILE_MAX_STATES = (size_t)STILE_MAX_TILE * 256;
constexpr long long WBC_STILE_MIN_F32 = 16384;

// stream of the launch + (optionally) the events that receive the dispatch's own start / stop timestamps
struct LaunchCtx {
  hipStream_t st = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  int f32_pack2 = 0;   // wbc_solver_options.f32_pack2 (dyn_sweep launches only)
};

// dyn_sweep_kernel<T, MODE>: MODE = SW_MATS | SW_STEP | SW_OBS bits (device_types.hpp); workgroup size chosen from N
template <class T> hipError_t k_dyn_sweep(const LaunchCtx& L, int mode, const DevModel<T>* model, const DevParams<T>& prm, const SweepArgs<T>& a);
// rnea_step_kernel<T, MODE>: CRBA-free front half of ticks whose caller passes no M/h/Jc buffers; MODE = RS_STEP [| RS_OBS] [| RS_PF]
template <class T> hipError_t k_rnea_step(const LaunchCtx& L, int mode, const DevModel<T>* model, const DevParams<T>& prm, const SweepArgs<T>& a);
// observer_kernel<T>: the momentum-observer update as its own kernel (large observer-on batches, second stream)
template <class T> hipError_t k_observer(const LaunchCtx& L, const DevModel<T>* model, const DevParams<T>& prm, const SweepArgs<T>& a);
// sweep_obs_kernel<T, W>: k_observer and the observer-free k_dyn_sweep(SW_MATS | SW_STEP | SW_NOB) as the two roles of ONE launch (mid-size observer-on batches)
template <class T> hipError_t k_sweep_obs(const LaunchCtx& L, const DevModel<T>* model, const DevParams<T>& prm, const SweepArgs<T>& a);
// tile_tick_kernel<T, W, NS>: the whole tick of a mid-size observer-on fp32 batch (even N, M / h / Jc outputs) as one launch of `states`-state workgroups
// (64 | 96 | 128: NS = 2 | 3 | 4 packed sweep wavefronts + as many observer wavefronts): sweep | observer roles, then the staged QP tile of the same
// states (tile_tick.hip.hpp).  The host picks the smallest size that makes ONE round of workgroups on the 256 CUs.
constexpr int TILE_TICK_STATES = 128;
// Beyond one round (N > 32 768) 64-state workgroups again, two resident per CU: they drift apart over the rounds, so that one's QP stage (latency-bound) shares the SIMDs with
// the other's roles (issue-bound) -- 49 152 states: 895 -> 960 M steps/s, 262 144: 1 051 -> 1 088; a tie at 65 536 (profiles/r06h_ab_tile_tick_states.log)
inline int tile_tick_states(size_t N) { return N <= 64 * 256 ? 64 : (N <= 96 * 256 ? 96 : (N <= 128 * 256 ? 128 : 64)); }   // (32-state workgroups for <= 8 192 states: measured, no faster -- the tick's floor is ~22 us; r06o)
// fp64, observer off: NS = 2 ... 7 sweep wavefronts of 16 states (a CU's LDS holds seven wavefronts' parking lots), again the smallest one-round size
inline int tile_tick_states_f64(size_t N) { const size_t ns = (N + 4095) / 4096; return 16 * (int)(ns < 2 ? 2 : (ns > 7 ? 7 : ns)); }
constexpr long long WBC_TILE_TICK_MIN_F64 = 8193;
// fp64, observer on: NS = 2 ... 4 sweep + as many observer wavefronts of 16 states -- one round of workgroups holds 64 x 256 states, larger batches take several
inline int tile_tick_states_f64_obs(size_t N) { const size_t ns = (N + 4095) / 4096; return 16 * (int)(ns < 2 ? 2 : (ns > 4 ? 4 : ns)); }
constexpr long long WBC_TILE_TICK_MAX_F64_OBS = 196608;
constexpr long long WBC_TILE_TICK_MIN = 8194;
template <class T> hipError_t k_tile_prepare();   // raises the dynamic-LDS limit of the tile_tick kernels (once per process and device)
template <class T> hipError_t k_qp_prepare();     // ... of the staged QP tile kernels
template <class T> hipError_t k_tile_tick(const LaunchCtx& L, bool observer, int states, const DevModel<T>* model, const DevParams<T>& prm, const SweepArgs<T>& a, const QpArgs<T>& qa, const QpJidx& jmap);
// GRF QP + torque map; rhat = the observer estimate arrives through the workspace (k_observer ran).
// tile = 0: qp_group16_kernel, one wavefront per workgroup, four consecutive states per wavefront;
// tile = 64 | 128 | 256 | 512: qp_tile_kernel, workgroups of four wavefronts deal a tile of that many states by predicted work
// list (optional): solve the states list[4 .. 4 + list[0]) instead of the whole batch (qp_list_kernel; the count is reset by the
// NEXT tick's front-half kernel, SweepArgs::qp_todo -- not here)
// warm (tile = 0, no list): qp_group16_kernel<.., WARM>, every state starts from its active set in a.aset_in
// body (tiles): 0 structured body on gathered inputs (qp_tile_kernel), 1 its lean fp32 form (from WBC_F32_DENSE_TILE_MIN states), 2 STAGED tiles --
// qp_stile_kernel, the tile's inputs through LDS, twelve wavefronts per workgroup, tile <= STILE_MAX_TILE (a multiple of 4)
template <class T> hipError_t k_qp(const LaunchCtx& L, bool rhat, int tile, const DevParams<T>& prm, const QpArgs<T>& a, const QpJidx& jmap,
                                  int* list = nullptr, bool warm = false, int body = 0);
// qp_lane_kernel<T, RHAT>: the GRF QP one state per LANE (semismooth Newton on the 6-dimensional residual wrench); states it
// does not finish are appended to todo (todo[0] = count, todo[4 ...] = indices) for k_qp(..., list = todo); the count must be
// zero when this kernel starts: the front-half kernel of the tick empties it (SweepArgs::qp_todo)
template <class T> hipError_t k_qp_lane(const LaunchCtx& L, bool rhat, const DevParams<T>& prm, const QpArgs<T>& a, const QpJidx& jmap, int* todo, bool warm = false);
// fused_tick_kernel<T, OBSERVER, MATS>: the whole tick of a small batch as one launch
template <class T> hipError_t k_fused_tick(const LaunchCtx& L, bool observer, bool mats, const DevModel<T>* model, const DevParams<T>& prm,
                                          const SweepArgs<T>& a, const QpArgs<T>& qa, const QpJidx& jmap, bool warm = false);
// fused_pair_kernel<T>: two 16-state tick workgroups as ONE twelve-wavefront workgroup of 32 states at 168 registers (fp64: the rnea / mass_jac roles spill; fp32: no spill) -- both halves
// resident on a CU together -- for observer-off, cold, M/h/Jc-writing ticks of N >= 64 states
// auto range (wbc_solver_options.fused_pair = 0), measured on MI355X: profiles/r06v_ab_fused_pair.log
// (just above 4 096 states the 16-state workgroups' second round is a handful of workgroups and the pair's tail workgroup costs ~0.9 us: 4 097: 18.7 against 19.7 us, 4 128: 19.0 / 18.8,
//  4 352: 20.0 / 18.5, 5 000: 21.3 / 19.6, 6 000: 24.4 / 20.5, 7 500: 24.4 / 20.3, 8 191: 27.4 / 21.1 -- profiles/r06y3_ab_fused_pair_ragged.log)
constexpr long long WBC_FUSED_PAIR_MIN = 4225, WBC_FUSED_PAIR_MAX = 8192, WBC_FUSED_PAIR_MAX_F32 = 16384;
template <class T> hipError_t k_fused_pair(const LaunchCtx& L, const DevModel<T>* model, const DevParams<T>& prm, const SweepArgs<T>& a, const QpArgs<T>& qa, const QpJidx& jmap);
// rollout_kernel<T, OBSERVER, TRACK, SPW>: `horizon` dependent ticks incl. forward dynamics (and the planner) as one launch
template <class T> hipError_t k_rollout(const LaunchCtx& L, bool observer, bool track, int spw, const DevModel<T>* model, const DevParams<T>& prm,
                                       const SweepArgs<T>& a, const QpArgs<T>& qa, const QpJidx& jmap, const IntegrateArgs<T>& ia, int horizon,
                                       const DevRefParams<T>* G, const RefArgs<T>& ra, bool warm = false);
// qp_general_kernel<T>: dense QPs of run-time size (n <= 36 variables, m <= 64 rows, the first meq of them equalities), one per wavefront
template <class T> hipError_t k_qp_general(const LaunchCtx& L, const QpGeneralArgs<T>& a);
// one thread: *ptr = value, system scope (the completion ticket of the flag-polled single-robot tick)
hipError_t k_flag(hipStream_t st, unsigned* ptr, unsigned value);
// the peer gather of wbc_multi_*: ONE launch copies `bytes` bytes at src to each of the nd <= 64 destinations (this device's or peer-mapped memory)
hipError_t k_gather_push(hipStream_t st, const void* src, void* const* dst, int nd, size_t bytes);
template <class T> hipError_t k_integrate(const LaunchCtx& L, const DevModel<T>* model, const IntegrateArgs<T>& a);
template <class T> hipError_t k_reference(const LaunchCtx& L, const DevModel<T>* model, const DevRefParams<T>* G, const RefArgs<T>& a);

}  // namespace wbc

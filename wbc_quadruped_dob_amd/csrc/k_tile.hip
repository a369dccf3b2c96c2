// tile_tick_kernel launches: sweep (+ observer) roles and the staged QP tile of the same states as one launch of big workgroups (tile_tick.hip.hpp).
#include "k_common.hip.hpp"
#include <type_traits>
#include "tile_tick.hip.hpp"

namespace wbc {

// (templates, so that the branches of the other scalar type are discarded, not instantiated)
template <class T, int W, int NS, int NWQ, bool OBS>
static hipError_t tile_tick_one(const LaunchCtx* L, const DevModel<T>* model, const DevParams<T>* prm, const SweepArgs<T>* a, const QpArgs<T>* qa, const QpJidx* jmap) {
  constexpr size_t smem = tile_tick_lds_bytes<T, W, NS, NWQ, OBS>();
  static_assert(smem <= 160 * 1024, "a CU has 160 kB of LDS");
  constexpr int TILE = 16 * W * NS;
  if (!L) return hipFuncSetAttribute((const void*)tile_tick_kernel<T, W, NS, NWQ, OBS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   // (prepare)
  WBC_KLAUNCH_SMEM(*L, (tile_tick_kernel<T, W, NS, NWQ, OBS>), dim3((unsigned)((a->N + TILE - 1) / TILE)), dim3(64 * NWQ), smem, model, *prm, *a, *qa, *jmap);
  return hipGetLastError();
}
// every instantiation the planner can ask for (tile_tick_states, launch.hpp); L = null: raise its dynamic-LDS limit instead of launching (`states` = 0: all of them)
template <class T>
static hipError_t tile_tick_any(const LaunchCtx* L, bool observer, int states, const DevModel<T>* model, const DevParams<T>* prm, const SweepArgs<T>* a, const QpArgs<T>* qa, const QpJidx* jmap) {
  hipError_t e = hipSuccess;
  bool hit = false;
#define TT_CASE(W_, NS_, NWQ_, OBS_) \
  if (e == hipSuccess && observer == OBS_ && (states == 0 || states == 16 * W_ * NS_)) { hit = true; e = tile_tick_one<T, W_, NS_, NWQ_, OBS_>(L, model, prm, a, qa, jmap); }
  if constexpr (std::is_same<T, float>::value) {
    // fp32: packed roles (32 states per wavefront), observer on: NS sweep + NS observer wavefronts
    TT_CASE(2, 2, 4, true) TT_CASE(2, 3, 6, true) TT_CASE(2, 4, 8, true)
    // fp32, observer off: NS packed sweep wavefronts, helpers for the QP stage
    TT_CASE(2, 2, 4, false) TT_CASE(2, 3, 6, false) TT_CASE(2, 4, 8, false)
  } else {
    // fp64, observer off: NS sweep wavefronts of 16 states; small tiles get helper wavefronts for the QP stage
    TT_CASE(1, 2, 8, false) TT_CASE(1, 3, 8, false) TT_CASE(1, 4, 8, false) TT_CASE(1, 5, 8, false) TT_CASE(1, 6, 8, false) TT_CASE(1, 7, 7, false)
    // fp64, observer on: NS sweep + NS observer wavefronts of 16 states (64 states: the CU's eight wavefronts)
    TT_CASE(1, 2, 8, true) TT_CASE(1, 3, 8, true) TT_CASE(1, 4, 8, true)
  }
#undef TT_CASE
  return hit ? e : hipErrorInvalidValue;
}

template <> hipError_t k_tile_prepare<Scalar>() {
  hipError_t e = tile_tick_any<Scalar>(nullptr, true, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (e == hipErrorInvalidValue) e = hipSuccess;   // (no instantiation with the observer for this scalar type)
  if (e == hipSuccess) { e = tile_tick_any<Scalar>(nullptr, false, 0, nullptr, nullptr, nullptr, nullptr, nullptr); if (e == hipErrorInvalidValue) e = hipSuccess; }
  return e;
}

template <>
hipError_t k_tile_tick<Scalar>(const LaunchCtx& L, bool observer, int states, const DevModel<Scalar>* model, const DevParams<Scalar>& prm, const SweepArgs<Scalar>& a, const QpArgs<Scalar>& qa, const QpJidx& jmap) {
  if (std::is_same<Scalar, float>::value && (a.N & 1) != 0) return hipErrorInvalidValue;   // (packed roles: even batches)
  return tile_tick_any<Scalar>(&L, observer, states, model, &prm, &a, &qa, &jmap);
}

}  // namespace wbc

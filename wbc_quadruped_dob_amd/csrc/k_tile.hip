// tile_tick_kernel launches: sweep | observer roles and the staged QP tile of the same states as one launch, one workgroup per CU (tile_tick.hip.hpp).
#include "k_common.hip.hpp"
#include <type_traits>
#include "tile_tick.hip.hpp"

namespace wbc {

// (templates, so that the branch of the other scalar type is discarded, not instantiated)
template <class T, int W, int NS>
static hipError_t tile_tick_ns(const LaunchCtx& L, const DevModel<T>* model, const DevParams<T>& prm, const SweepArgs<T>& a, const QpArgs<T>& qa, const QpJidx& jmap) {
  constexpr size_t smem = tile_tick_lds_bytes<T, W, NS>();
  constexpr int TILE = 16 * W * NS;
  static bool raised = false;
  if (!raised) {
    const hipError_t e = hipFuncSetAttribute((const void*)tile_tick_kernel<T, W, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    raised = true;
  }
  WBC_KLAUNCH_SMEM(L, (tile_tick_kernel<T, W, NS>), dim3((unsigned)((a.N + TILE - 1) / TILE)), dim3(128 * NS), smem, model, prm, a, qa, jmap);
  return hipGetLastError();
}
template <class T>
static hipError_t tile_tick_launch(const LaunchCtx& L, int states, const DevModel<T>* model, const DevParams<T>& prm, const SweepArgs<T>& a, const QpArgs<T>& qa, const QpJidx& jmap) {
  if constexpr (std::is_same<T, float>::value) {
    if ((a.N & 1) != 0) return hipErrorInvalidValue;
    switch (states) {   // (states per workgroup = 32 per sweep wavefront: the host picks the smallest that makes one round of workgroups, tile_tick_states)
      case 64: return tile_tick_ns<T, 2, 2>(L, model, prm, a, qa, jmap);
      case 96: return tile_tick_ns<T, 2, 3>(L, model, prm, a, qa, jmap);
      case 128: return tile_tick_ns<T, 2, 4>(L, model, prm, a, qa, jmap);
      default: return hipErrorInvalidValue;
    }
  } else return hipErrorInvalidValue;   // (fp32 solvers only: the packed roles and the fp32 image are what fits a CU)
}

// the dynamic-LDS limit of every instantiation, raised once (wbc_solver_create: a first launch inside a stream capture must not have to)
template <class T>
static hipError_t tile_tick_prepare_t() {
  if constexpr (std::is_same<T, float>::value) {
    hipError_t e = hipFuncSetAttribute((const void*)tile_tick_kernel<T, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tile_tick_kernel<T, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tile_tick_kernel<T, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
  } else return hipSuccess;
}
template <> hipError_t k_tile_prepare<Scalar>() { return tile_tick_prepare_t<Scalar>(); }

template <>
hipError_t k_tile_tick<Scalar>(const LaunchCtx& L, int states, const DevModel<Scalar>* model, const DevParams<Scalar>& prm, const SweepArgs<Scalar>& a, const QpArgs<Scalar>& qa, const QpJidx& jmap) {
  return tile_tick_launch<Scalar>(L, states, model, prm, a, qa, jmap);
}

}  // namespace wbc

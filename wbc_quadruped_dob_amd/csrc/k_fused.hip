// fused_tick_kernel launches: the whole tick of a small batch as one launch of wavefront roles (fused_tick.hip.hpp).
#include "k_common.hip.hpp"
#include "fused_tick.hip.hpp"

namespace wbc {

template <>
hipError_t k_fused_tick<Scalar>(const LaunchCtx& L, bool observer, bool mats, const DevModel<Scalar>* model, const DevParams<Scalar>& prm,
                                const SweepArgs<Scalar>& a, const QpArgs<Scalar>& qa, const QpJidx& jmap, bool warm) {
  using T = Scalar;
  const dim3 grid((unsigned)((a.N + 15) / 16));
  constexpr unsigned obs_threads = 384 + 64 * FUSED_OBS_WAVES;
  if (warm) {
    if (observer && mats) WBC_KLAUNCH(L, (fused_tick_kernel<T, true, true, true>), grid, dim3(obs_threads), model, prm, a, qa, jmap);
    else if (observer) WBC_KLAUNCH(L, (fused_tick_kernel<T, true, false, true>), grid, dim3(obs_threads), model, prm, a, qa, jmap);
    else if (mats) WBC_KLAUNCH(L, (fused_tick_kernel<T, false, true, true>), grid, dim3((unsigned)fused_threads<T, false, true, true>()), model, prm, a, qa, jmap);
    else WBC_KLAUNCH(L, (fused_tick_kernel<T, false, false, true>), grid, dim3(384), model, prm, a, qa, jmap);
    return hipGetLastError();
  }
  if (observer && mats) WBC_KLAUNCH(L, (fused_tick_kernel<T, true, true>), grid, dim3(obs_threads), model, prm, a, qa, jmap);
  else if (observer) WBC_KLAUNCH(L, (fused_tick_kernel<T, true, false>), grid, dim3(obs_threads), model, prm, a, qa, jmap);
  else if (mats) WBC_KLAUNCH(L, (fused_tick_kernel<T, false, true>), grid, dim3((unsigned)fused_threads<T, false, true>()), model, prm, a, qa, jmap);
  else WBC_KLAUNCH(L, (fused_tick_kernel<T, false, false>), grid, dim3(384), model, prm, a, qa, jmap);
  return hipGetLastError();
}

// fused_pair_kernel: N >= 64, observer off, M / h / Jc wanted, cold
template <>
hipError_t k_fused_pair<Scalar>(const LaunchCtx& L, const DevModel<Scalar>* model, const DevParams<Scalar>& prm, const SweepArgs<Scalar>& a, const QpArgs<Scalar>& qa,
                                const QpJidx& jmap) {
  using T = Scalar;
  if (a.N < 64) return hipErrorInvalidValue;
  WBC_KLAUNCH(L, (fused_pair_kernel<T>), dim3((unsigned)((a.N + 31) / 32)), dim3((unsigned)FUSED_PAIR_THREADS), model, prm, a, qa, jmap);
  return hipGetLastError();
}

}  // namespace wbc

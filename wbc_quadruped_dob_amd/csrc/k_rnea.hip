// rnea_step_kernel launches: the CRBA-free front half of ticks whose caller wants tau, f only (dyn_split.hip.hpp).
#include "k_common.hip.hpp"
#include "dyn_split.hip.hpp"

namespace wbc {

template <int MODE>
static hipError_t rnea_mode(const LaunchCtx& L, const DevModel<Scalar>* model, const DevParams<Scalar>& prm, const SweepArgs<Scalar>& a) {
  using T = Scalar;
  const size_t threads = a.N * 4;
  // 256-thread workgroups only where four waves' parked state fits the CU twice (one force chain, no observer)
  if constexpr ((MODE & RS_OBS) == 0) {
    if (threads >= BIG_GRID_THREADS) {
      WBC_KLAUNCH(L, (rnea_step_kernel<T, MODE, 256>), dim3((unsigned)((threads + 255) / 256)), dim3(256), model, prm, a);
      return hipGetLastError();
    }
  }
  WBC_KLAUNCH(L, (rnea_step_kernel<T, MODE, 64>), dim3((unsigned)((threads + 63) / 64)), dim3(64), model, prm, a);
  return hipGetLastError();
}

template <>
hipError_t k_rnea_step<Scalar>(const LaunchCtx& L, int mode, const DevModel<Scalar>* model, const DevParams<Scalar>& prm, const SweepArgs<Scalar>& a) {
  switch (mode) {
    case RS_STEP: return rnea_mode<RS_STEP>(L, model, prm, a);
    case RS_STEP | RS_OBS: return rnea_mode<RS_STEP | RS_OBS>(L, model, prm, a);
    case RS_STEP | RS_PF: return rnea_mode<RS_STEP | RS_PF>(L, model, prm, a);
    case RS_STEP | RS_NOB: return rnea_mode<RS_STEP | RS_NOB>(L, model, prm, a);
    case RS_STEP | RS_PF | RS_NOB: return rnea_mode<RS_STEP | RS_PF | RS_NOB>(L, model, prm, a);
    case RS_STEP | RS_OBS | RS_PF: return rnea_mode<RS_STEP | RS_OBS | RS_PF>(L, model, prm, a);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace wbc

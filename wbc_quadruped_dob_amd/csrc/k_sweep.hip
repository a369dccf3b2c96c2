// dyn_sweep_kernel launches (units a1-a6 + the a7/a9 prologue; dyn_sweep.hip.hpp).
#include "k_common.hip.hpp"
#include "dyn_sweep.hip.hpp"

namespace wbc {

template <int MODE>
static hipError_t sweep_mode(const LaunchCtx& L, const DevModel<Scalar>* model, const DevParams<Scalar>& prm, const SweepArgs<Scalar>& a) {
  using T = Scalar;
  const size_t threads = a.N * 4;
  if constexpr ((MODE & SW_OBS) == 0) {  // the observer variants park too much per wave for 256-thread workgroups
    if (threads >= BIG_GRID_THREADS) {
      WBC_KLAUNCH(L, (dyn_sweep_kernel<T, MODE, 256>), dim3((unsigned)((threads + 255) / 256)), dim3(256), model, prm, a);
      return hipGetLastError();
    }
  }
  WBC_KLAUNCH(L, (dyn_sweep_kernel<T, MODE, 64>), dim3((unsigned)((threads + 63) / 64)), dim3(64), model, prm, a);
  return hipGetLastError();
}

template <>
hipError_t k_dyn_sweep<Scalar>(const LaunchCtx& L, int mode, const DevModel<Scalar>* model, const DevParams<Scalar>& prm, const SweepArgs<Scalar>& a) {
  switch (mode) {   // (the combinations the host side launches: wbc_api.cpp)
    case 0: return sweep_mode<0>(L, model, prm, a);                                   // pf only
    case SW_MATS: return sweep_mode<SW_MATS>(L, model, prm, a);
    case SW_OBS: return sweep_mode<SW_OBS>(L, model, prm, a);
    case SW_MATS | SW_OBS: return sweep_mode<SW_MATS | SW_OBS>(L, model, prm, a);
    case SW_MATS | SW_STEP | SW_NOB: return sweep_mode<SW_MATS | SW_STEP | SW_NOB>(L, model, prm, a);
    case SW_MATS | SW_STEP | SW_OBS: return sweep_mode<SW_MATS | SW_STEP | SW_OBS>(L, model, prm, a);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace wbc

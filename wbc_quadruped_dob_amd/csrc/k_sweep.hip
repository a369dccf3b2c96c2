// dyn_sweep_kernel launches (units a1-a6 + the a7/a9 prologue; dyn_sweep.hip.hpp).
#include "k_common.hip.hpp"
#include <type_traits>
#include "dyn_sweep.hip.hpp"
#include "observer.hip.hpp"

namespace wbc {

template <int MODE, int W>
static hipError_t sweep_launch(const LaunchCtx& L, const DevModel<Scalar>* model, const DevParams<Scalar>& prm, const SweepArgs<Scalar>& a) {
  using T = Scalar;
  const size_t threads = ((a.N + W - 1) / W) * 4;   // one lane per leg and per W consecutive states
  if constexpr ((MODE & SW_OBS) == 0) {  // the observer variants park too much per wave for 256-thread workgroups
    if (threads >= BIG_GRID_THREADS) {
      WBC_KLAUNCH(L, (dyn_sweep_kernel<T, MODE, 256, W>), dim3((unsigned)((threads + 255) / 256)), dim3(256), model, prm, a);
      return hipGetLastError();
    }
  }
  WBC_KLAUNCH(L, (dyn_sweep_kernel<T, MODE, 64, W>), dim3((unsigned)((threads + 63) / 64)), dim3(64), model, prm, a);
  return hipGetLastError();
}

// fp32, even N: two states per lane as packed pairs (whole 128-byte lines per 16-lane row, v_pk_* arithmetic, half the
// wavefronts).  Below PACK2_MIN_STATES the batch does not fill the SIMDs with one state per lane either, and the shorter
// dependent chain per state of the unpacked form wins.
template <int MODE>
static hipError_t sweep_mode(const LaunchCtx& L, const DevModel<Scalar>* model, const DevParams<Scalar>& prm, const SweepArgs<Scalar>& a) {
  if constexpr (std::is_same<Scalar, float>::value) {
    if ((a.N & 1) == 0 && L.f32_pack2 >= 0 && (L.f32_pack2 > 0 || a.N >= (size_t)WBC_PACK2_MIN_STATES)) return sweep_launch<MODE, 2>(L, model, prm, a);
  }
  return sweep_launch<MODE, 1>(L, model, prm, a);
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

// sweep_obs_kernel: the observer update + the observer-free sweep as the two roles of one launch (observer.hip.hpp).  fp32, even N: both roles
// with two states per lane (unless f32_pack2 = -1) -- the form exists for batches whose 2 x N / 16 (packed: N / 32) wavefronts fit one round.
template <>
hipError_t k_sweep_obs<Scalar>(const LaunchCtx& L, const DevModel<Scalar>* model, const DevParams<Scalar>& prm, const SweepArgs<Scalar>& a) {
  using T = Scalar;
  if constexpr (std::is_same<Scalar, float>::value) {
    if ((a.N & 1) == 0 && L.f32_pack2 >= 0) {
      const unsigned nsw = (unsigned)((a.N / 2 + 15) / 16);
      WBC_KLAUNCH(L, (sweep_obs_kernel<T, 2>), dim3(2 * nsw), dim3(64), model, prm, a, nsw);
      return hipGetLastError();
    }
  }
  const unsigned nsw = (unsigned)((a.N + 15) / 16);
  WBC_KLAUNCH(L, (sweep_obs_kernel<T, 1>), dim3(2 * nsw), dim3(64), model, prm, a, nsw);
  return hipGetLastError();
}

}  // namespace wbc

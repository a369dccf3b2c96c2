// qp_group16_kernel launches: GRF QP + torque map (units a7-a9; qp_group16.hip.hpp).
#include "k_common.hip.hpp"
#include "qp_kernels.hip.hpp"
#include "qp_lane.hip.hpp"
#include <type_traits>

namespace wbc {

#ifndef WBC_F32_DENSE_TILE_MIN
#define WBC_F32_DENSE_TILE_MIN 49152
#endif
template <int TILE>
static hipError_t qp_tiled(const LaunchCtx& L, bool rhat, const DevParams<Scalar>& prm, const QpArgs<Scalar>& a, const QpJidx& jmap) {
  using T = Scalar;
  const dim3 grid((unsigned)((a.N + TILE - 1) / TILE));
  if constexpr (std::is_same<T, float>::value && TILE == 64) {
    if (a.N >= (size_t)WBC_F32_DENSE_TILE_MIN) {   // more than two tiles per CU: the leaner fp32 body wins on occupancy (qp_kernels.hip.hpp)
      if (rhat) WBC_KLAUNCH(L, (qp_tile_kernel<T, true, TILE, true>), grid, dim3(256), prm, a, jmap);
      else WBC_KLAUNCH(L, (qp_tile_kernel<T, false, TILE, true>), grid, dim3(256), prm, a, jmap);
      return hipGetLastError();
    }
  }
  if (rhat) WBC_KLAUNCH(L, (qp_tile_kernel<T, true, TILE>), grid, dim3(256), prm, a, jmap);
  else WBC_KLAUNCH(L, (qp_tile_kernel<T, false, TILE>), grid, dim3(256), prm, a, jmap);
  return hipGetLastError();
}

template <>
hipError_t k_qp<Scalar>(const LaunchCtx& L, bool rhat, int tile, const DevParams<Scalar>& prm, const QpArgs<Scalar>& a, const QpJidx& jmap, int* list) {
  using T = Scalar;
  if (list) {   // the hand-over list of the per-lane kernel: one wavefront per four listed states, grid-stride
    const dim3 grid((unsigned)((a.N + 31) / 32));
    if (rhat) WBC_KLAUNCH(L, (qp_list_kernel<T, true>), grid, dim3(64), prm, a, jmap, list);
    else WBC_KLAUNCH(L, (qp_list_kernel<T, false>), grid, dim3(64), prm, a, jmap, list);
    return hipGetLastError();
  }
  if (tile == 32) return qp_tiled<32>(L, rhat, prm, a, jmap);
  if (tile == 64) return qp_tiled<64>(L, rhat, prm, a, jmap);
  if (tile == 128) return qp_tiled<128>(L, rhat, prm, a, jmap);
  if (tile == 256) return qp_tiled<256>(L, rhat, prm, a, jmap);
  if (tile == 512) return qp_tiled<512>(L, rhat, prm, a, jmap);
  const dim3 grid((unsigned)((a.N + 3) / 4));   // one wavefront (four QPs) per workgroup
  if (rhat) WBC_KLAUNCH(L, (qp_group16_kernel<T, true>), grid, dim3(64), prm, a, jmap);
  else WBC_KLAUNCH(L, (qp_group16_kernel<T, false>), grid, dim3(64), prm, a, jmap);
  return hipGetLastError();
}

template <>
hipError_t k_qp_lane<Scalar>(const LaunchCtx& L, bool rhat, const DevParams<Scalar>& prm, const QpArgs<Scalar>& a, const QpJidx& jmap, int* todo) {
  using T = Scalar;
  const dim3 grid((unsigned)((a.N + QPL_WG - 1) / QPL_WG));   // one state per lane
  if (rhat) WBC_KLAUNCH(L, (qp_lane_kernel<T, true>), grid, dim3(QPL_WG), prm, a, jmap, todo);
  else WBC_KLAUNCH(L, (qp_lane_kernel<T, false>), grid, dim3(QPL_WG), prm, a, jmap, todo);
  return hipGetLastError();
}

}  // namespace wbc

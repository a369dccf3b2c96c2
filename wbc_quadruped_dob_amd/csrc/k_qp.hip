// qp_group16_kernel launches: GRF QP + torque map (units a7-a9; qp_group16.hip.hpp).
#include "k_common.hip.hpp"
#include "qp_kernels.hip.hpp"
#include "qp_lane.hip.hpp"
#include <type_traits>

namespace wbc {

// staged tiles: NW wavefronts per workgroup, tiles of up to 64 CH states
template <int NW, int CH>
static hipError_t qp_staged(const LaunchCtx& L, bool rhat, int tile, const DevParams<Scalar>& prm, const QpArgs<Scalar>& a, const QpJidx& jmap) {
  using T = Scalar;
  const size_t smem = stile_lds_bytes(tile, sizeof(T), NW);
  static bool raised[2] = {false, false};   // (more than 64 kB of dynamic LDS needs the attribute once per kernel)
  if (!raised[rhat ? 1 : 0]) {
    const hipError_t e = rhat ? hipFuncSetAttribute((const void*)qp_stile_kernel<T, true, NW, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                              : hipFuncSetAttribute((const void*)qp_stile_kernel<T, false, NW, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    raised[rhat ? 1 : 0] = true;
  }
  const dim3 grid((unsigned)((a.N + tile - 1) / tile));
  if (rhat) WBC_KLAUNCH_SMEM(L, (qp_stile_kernel<T, true, NW, CH>), grid, dim3(64 * NW), smem, prm, a, jmap, tile);
  else WBC_KLAUNCH_SMEM(L, (qp_stile_kernel<T, false, NW, CH>), grid, dim3(64 * NW), smem, prm, a, jmap, tile);
  return hipGetLastError();
}

template <int TILE>
static hipError_t qp_tiled(const LaunchCtx& L, bool rhat, const DevParams<Scalar>& prm, const QpArgs<Scalar>& a, const QpJidx& jmap) {
  using T = Scalar;
  const dim3 grid((unsigned)((a.N + TILE - 1) / TILE));
  if constexpr (std::is_same<T, float>::value && TILE >= 64 && TILE <= 128) {
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
hipError_t k_qp<Scalar>(const LaunchCtx& L, bool rhat, int tile, const DevParams<Scalar>& prm, const QpArgs<Scalar>& a, const QpJidx& jmap, int* list, bool warm, int body) {
  using T = Scalar;
  if (warm && !list) {   // dependent ticks: every state starts from its previous active set, so the rows of a wavefront do about equal work -- no dealing by predicted work
    if (tile > 0) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((a.N + 3) / 4));
    if (rhat) WBC_KLAUNCH(L, (qp_group16_kernel<T, true, true>), grid, dim3(64), prm, a, jmap);
    else WBC_KLAUNCH(L, (qp_group16_kernel<T, false, true>), grid, dim3(64), prm, a, jmap);
    return hipGetLastError();
  }
  if (list) {   // the hand-over list of the per-lane kernel: one wavefront per four listed states, grid-stride
    const dim3 grid((unsigned)((a.N + 31) / 32));
    if (warm) {
      if (rhat) WBC_KLAUNCH(L, (qp_list_kernel<T, true, true>), grid, dim3(64), prm, a, jmap, list);
      else WBC_KLAUNCH(L, (qp_list_kernel<T, false, true>), grid, dim3(64), prm, a, jmap, list);
      return hipGetLastError();
    }
    if (rhat) WBC_KLAUNCH(L, (qp_list_kernel<T, true>), grid, dim3(64), prm, a, jmap, list);
    else WBC_KLAUNCH(L, (qp_list_kernel<T, false>), grid, dim3(64), prm, a, jmap, list);
    return hipGetLastError();
  }
  // tile sizes in small steps so that the host can launch ONE round of resident workgroups (wbc_api.cpp): fp64 tiles keep three workgroups on a
  // CU (768 at once) up to 64 states per tile, fp32 tiles two (512) -- hence steps of 4 from 32 to 64 for fp64, of 8 from 64 to 128 for fp32
  if (body == 2) {
    if constexpr (std::is_same<T, float>::value) {
      if (tile <= 0 || tile > STILE_MAX_TILE || tile % 4 != 0) return hipErrorInvalidValue;
      const int ch = (tile + 63) / 64;
      if (ch == 1) return qp_staged<12, 1>(L, rhat, tile, prm, a, jmap);
      if (ch == 2) return qp_staged<12, 2>(L, rhat, tile, prm, a, jmap);
      return qp_staged<12, 3>(L, rhat, tile, prm, a, jmap);
    } else return hipErrorInvalidValue;   // (fp64 solvers: the image of a CU's states does not fit beside the solver tables)
  }
#define WBC_TILE_CASE(n) case n: return qp_tiled<n>(L, rhat, prm, a, jmap);
  if constexpr (std::is_same<T, double>::value) {
    switch (tile) {
      WBC_TILE_CASE(32) WBC_TILE_CASE(36) WBC_TILE_CASE(40) WBC_TILE_CASE(44) WBC_TILE_CASE(48) WBC_TILE_CASE(52) WBC_TILE_CASE(56) WBC_TILE_CASE(60)
      WBC_TILE_CASE(64) WBC_TILE_CASE(128) WBC_TILE_CASE(256) WBC_TILE_CASE(512)
      default: break;
    }
  } else {
    switch (tile) {
      WBC_TILE_CASE(32) WBC_TILE_CASE(36) WBC_TILE_CASE(40) WBC_TILE_CASE(44) WBC_TILE_CASE(48) WBC_TILE_CASE(52) WBC_TILE_CASE(56) WBC_TILE_CASE(60)
      WBC_TILE_CASE(64) WBC_TILE_CASE(72) WBC_TILE_CASE(80) WBC_TILE_CASE(88) WBC_TILE_CASE(96) WBC_TILE_CASE(104) WBC_TILE_CASE(112)
      WBC_TILE_CASE(120) WBC_TILE_CASE(128) WBC_TILE_CASE(256) WBC_TILE_CASE(512)
      default: break;
    }
  }
#undef WBC_TILE_CASE
  if (tile > 0) return hipErrorInvalidValue;   // a size this scalar type has no kernel for (the host validates the option per type)
  const dim3 grid((unsigned)((a.N + 3) / 4));   // one wavefront (four QPs) per workgroup
  if (rhat) WBC_KLAUNCH(L, (qp_group16_kernel<T, true>), grid, dim3(64), prm, a, jmap);
  else WBC_KLAUNCH(L, (qp_group16_kernel<T, false>), grid, dim3(64), prm, a, jmap);
  return hipGetLastError();
}

template <class T>
static hipError_t qp_prepare_t() {
  if constexpr (std::is_same<T, float>::value) {
    hipError_t e = hipSuccess;
    auto up = [&](const void* f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); };
    up((const void*)qp_stile_kernel<T, true, 12, 1>); up((const void*)qp_stile_kernel<T, false, 12, 1>);
    up((const void*)qp_stile_kernel<T, true, 12, 2>); up((const void*)qp_stile_kernel<T, false, 12, 2>);
    up((const void*)qp_stile_kernel<T, true, 12, 3>); up((const void*)qp_stile_kernel<T, false, 12, 3>);
    return e;
  } else return hipSuccess;
}
template <> hipError_t k_qp_prepare<Scalar>() { return qp_prepare_t<Scalar>(); }

template <>
hipError_t k_qp_lane<Scalar>(const LaunchCtx& L, bool rhat, const DevParams<Scalar>& prm, const QpArgs<Scalar>& a, const QpJidx& jmap, int* todo, bool warm) {
  using T = Scalar;
  const dim3 grid((unsigned)((a.N + QPL_WG - 1) / QPL_WG));   // one state per lane
  if (warm) {
    if (rhat) WBC_KLAUNCH(L, (qp_lane_kernel<T, true, true>), grid, dim3(QPL_WG), prm, a, jmap, todo);
    else WBC_KLAUNCH(L, (qp_lane_kernel<T, false, true>), grid, dim3(QPL_WG), prm, a, jmap, todo);
    return hipGetLastError();
  }
  if (rhat) WBC_KLAUNCH(L, (qp_lane_kernel<T, true>), grid, dim3(QPL_WG), prm, a, jmap, todo);
  else WBC_KLAUNCH(L, (qp_lane_kernel<T, false>), grid, dim3(QPL_WG), prm, a, jmap, todo);
  return hipGetLastError();
}

}  // namespace wbc

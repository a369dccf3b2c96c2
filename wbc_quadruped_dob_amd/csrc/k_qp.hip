// qp_group16_kernel launches: GRF QP + torque map (units a7-a9; qp_group16.hip.hpp).
#include "k_common.hip.hpp"
#include "qp_group16.hip.hpp"

namespace wbc {

template <>
hipError_t k_qp<Scalar>(const LaunchCtx& L, bool rhat, const DevParams<Scalar>& prm, const QpArgs<Scalar>& a, const QpJidx& jmap) {
  using T = Scalar;
  const dim3 grid((unsigned)((a.N + 3) / 4));   // one wavefront (four QPs) per workgroup
  if (rhat) WBC_KLAUNCH(L, (qp_group16_kernel<T, true>), grid, dim3(64), prm, a, jmap);
  else WBC_KLAUNCH(L, (qp_group16_kernel<T, false>), grid, dim3(64), prm, a, jmap);
  return hipGetLastError();
}

}  // namespace wbc

// qp_general_kernel launches: dense QPs of run-time size, one per wavefront (qp_general.hip.hpp).
#include "k_common.hip.hpp"
#include "qp_general.hip.hpp"

namespace wbc {

template <>
hipError_t k_qp_general<Scalar>(const LaunchCtx& L, const QpGeneralArgs<Scalar>& a) {
  using T = Scalar;
  const int per_qp = qpg_lds_scalars(a.n, a.m);
  const size_t bytes = (size_t)per_qp * sizeof(T);
  int wpb = (int)(65536 / bytes);   // wavefronts (= QPs) per workgroup: as many as 64 KB of LDS hold, at most four
  wpb = wpb < 1 ? 1 : (wpb > 4 ? 4 : wpb);
  const dim3 grid((unsigned)((a.N + wpb - 1) / wpb));
  if (L.ev_start) hipExtLaunchKernelGGL((qp_general_kernel<T>), grid, dim3(64 * wpb), bytes * wpb, L.st, L.ev_start, L.ev_stop, 0, a, per_qp);
  else hipLaunchKernelGGL((qp_general_kernel<T>), grid, dim3(64 * wpb), bytes * wpb, L.st, a, per_qp);
  return hipGetLastError();
}

}  // namespace wbc

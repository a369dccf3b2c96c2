// The small kernels: stand-alone observer update, forward dynamics + integrator, CoM reference generator.
#include "k_common.hip.hpp"
#include "observer.hip.hpp"
#include "integrate.hip.hpp"
#include "com_ref.hip.hpp"

namespace wbc {

template <>
hipError_t k_observer<Scalar>(const LaunchCtx& L, const DevModel<Scalar>* model, const DevParams<Scalar>& prm, const SweepArgs<Scalar>& a) {
  using T = Scalar;
  if (a.N * 4 >= BIG_GRID_THREADS)
    WBC_KLAUNCH(L, (observer_kernel<T, 256>), dim3((unsigned)((a.N * 4 + 255) / 256)), dim3(256), model, prm, a);
  else
    WBC_KLAUNCH(L, (observer_kernel<T, 64>), dim3((unsigned)((a.N + 15) / 16)), dim3(64), model, prm, a);
  return hipGetLastError();
}

template <>
hipError_t k_integrate<Scalar>(const LaunchCtx& L, const DevModel<Scalar>* model, const IntegrateArgs<Scalar>& a) {
  WBC_KLAUNCH(L, (integrate_kernel<Scalar>), dim3((unsigned)((a.N + 15) / 16)), dim3(64), model, a);
  return hipGetLastError();
}

template <>
hipError_t k_reference<Scalar>(const LaunchCtx& L, const DevModel<Scalar>* model, const DevRefParams<Scalar>* G, const RefArgs<Scalar>& a) {
  WBC_KLAUNCH(L, (com_reference_kernel<Scalar>), dim3((unsigned)((a.N + 15) / 16)), dim3(64), model, G, a);
  return hipGetLastError();
}

#ifdef WBC_SCALAR_IS_DOUBLE   // (defined once: this unit is compiled per scalar type)
__global__ void flag_kernel(unsigned* ptr, unsigned value) {
  __hip_atomic_store(ptr, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t k_flag(hipStream_t st, unsigned* ptr, unsigned value) {
  hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, st, ptr, value);
  return hipGetLastError();
}
#endif

}  // namespace wbc

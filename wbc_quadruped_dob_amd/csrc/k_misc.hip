// The small kernels: stand-alone observer update, forward dynamics + integrator, CoM reference generator.
#include "k_common.hip.hpp"
#include <type_traits>
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

// Consumer-side gather of the torques without RCCL (wbc_multi.cpp): a shard PUSHES its block into every device's buffer -- one read of
// the block, nd coalesced writes, local or through the peer mappings over xGMI -- as one launch instead of nd copies.
struct PushDst { void* p[64]; };
template <class W>
__global__ __launch_bounds__(256) void gather_push_kernel(const W* __restrict__ src, PushDst dst, int nd, size_t words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
    const W v = src[i];
    for (int d = 0; d < nd; ++d) ((W*)dst.p[d])[i] = v;
  }
}
hipError_t k_gather_push(hipStream_t st, const void* src, void* const* dst, int nd, size_t bytes) {
  if (nd < 1 || nd > 64 || (bytes & 3) != 0) return hipErrorInvalidValue;
  PushDst pd;
  bool wide = ((size_t)src & 15) == 0 && (bytes & 15) == 0;
  for (int d = 0; d < nd; ++d) { pd.p[d] = dst[d]; wide = wide && ((size_t)dst[d] & 15) == 0; }
  const size_t words = wide ? bytes / 16 : bytes / 4;
  size_t grid = (words + 255) / 256;
  if (grid > 2048) grid = 2048;
  if (grid == 0) return hipSuccess;
  if (wide) hipLaunchKernelGGL(gather_push_kernel<uint4>, dim3((unsigned)grid), dim3(256), 0, st, (const uint4*)src, pd, nd, words);
  else hipLaunchKernelGGL(gather_push_kernel<unsigned>, dim3((unsigned)grid), dim3(256), 0, st, (const unsigned*)src, pd, nd, words);
  return hipGetLastError();
}
#endif

}  // namespace wbc

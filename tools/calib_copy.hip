// PMC calibration: streaming kernels of KNOWN byte counts in the access widths the WBC kernels use
// (8 B/lane and 4 B/lane component-major rows), so that FETCH_SIZE / WRITE_SIZE can be corrected as
// MI355X_MICROARCH.md (HBM section) prescribes ("calibrate on a known byte count in your own access pattern").
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T> __global__ void calib_copy(const T* __restrict__ in, T* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
template <class T> __global__ void calib_write(T* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (T)i;
}
int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB per buffer: well past the 256 MiB Infinity Cache
  void *a, *b;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    size_t n8 = bytes / 8, n4 = bytes / 4;
    hipLaunchKernelGGL(calib_copy<double>, dim3((n8 + 255) / 256), dim3(256), 0, 0, (const double*)a, (double*)b, n8);
    hipLaunchKernelGGL(calib_copy<float>, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const float*)a, (float*)b, n4);
    hipLaunchKernelGGL(calib_write<double>, dim3((n8 + 255) / 256), dim3(256), 0, 0, (double*)b, n8);
    hipLaunchKernelGGL(calib_write<float>, dim3((n4 + 255) / 256), dim3(256), 0, 0, (float*)b, n4);
  }
  hipDeviceSynchronize();
  std::printf("calib done: each kernel moves %zu bytes per direction\n", bytes);
  return 0;
}

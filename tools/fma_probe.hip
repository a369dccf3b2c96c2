// fp64 / fp32 vector FMA peak on this device (SURVEY.md 8d asks for the FP64 ridge to be re-measured on the box):
// 16 independent FMA chains per lane, 8 waves per SIMD, no memory traffic.  Prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>

template <class T, int CH>
__global__ __launch_bounds__(256) void k_fma(T* out, T a, T b, int iters) {
  T x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = (T)(threadIdx.x + c);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = x[c] * a + b;
  }
  T s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += x[c];
  if (s == (T)123.456) out[0] = s;
}

template <class T> double run(const char* name) {
  T* out; hipMalloc(&out, 64);
  const int iters = 4096, CH = 16, blocks = 256 * 8 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_fma<T, CH><<<blocks, 256>>>(out, (T)0.999, (T)0.001, 16);
  hipDeviceSynchronize();
  double best = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    k_fma<T, CH><<<blocks, 256>>>(out, (T)0.999, (T)0.001, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = 2.0 * CH * (double)iters * blocks * 256 / (ms * 1e-3) / 1e12;
    if (tf > best) best = tf;
  }
  printf("%s scalar-FMA chains: %.1f TFLOP/s\n", name, best);
  hipFree(out);
  return best;
}

int main() {
  run<double>("fp64");
  run<float>("fp32 (non-packed)");
  return 0;
}

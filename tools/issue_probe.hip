// Single-wavefront issue / latency probe for gfx950: what does ONE wavefront that is alone on its SIMD pay per instruction?
// (the QP role of the fused tick and every sweep at N <= 4 096 run exactly like that).  Prints cycles per instruction for
// dependent and independent chains of the instruction kinds the hot loops are made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 2048
template <int CHAINS> __global__ void fma_f64(double* out, long long* cyc, double a, double b) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
  const long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < REP / CHAINS / 4; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], b, a);
  }
  const long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CHAINS> __global__ void fma_f32(float* out, long long* cyc, float a, float b) {
  float x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
  const long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < REP / CHAINS / 4; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fmaf(x[c], b, a);
  }
  const long long t1 = clock64();
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// the same with only the first LANES lanes of the wavefront active (does the SIMD skip the idle quarters of a wave64 operation?)
template <int CHAINS, int LANES> __global__ void fma_f64_masked(double* out, long long* cyc, double a, double b) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
  long long t0 = 0, t1 = 0;
  if (threadIdx.x < LANES) {
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < REP / CHAINS / 4; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], b, a);
    }
    t1 = clock64();
  }
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// dependent chain: dpp mov (32-bit x2) + min f64  (one step of the row argmin)
__global__ void dpp_min(double* out, long long* cyc, double a) {
  double k = a + threadIdx.x;
  const long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < REP / 4; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int lo = __double2loint(k), hi = __double2hiint(k);
      lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
      hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
      double o = __hiloint2double(hi, lo), r;
      asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(k), "v"(o));
      k = r + 1e-9;
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x + blockIdx.x * blockDim.x] = k;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// dependent chain through LDS: write then read (same wave, other lane)
__global__ void lds_rt(double* out, long long* cyc, double a) {
  __shared__ double buf[64];
  double k = a + threadIdx.x;
  const long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < REP / 4; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      buf[threadIdx.x] = k;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      k = buf[(threadIdx.x + 1) & 63] + 1.0;
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x + blockIdx.x * blockDim.x] = k;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// rcp + two Newton steps, dependent
__global__ void rcp_chain(double* out, long long* cyc, double a) {
  double x = a + threadIdx.x;
  const long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < REP / 4; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      double y = __builtin_amdgcn_rcp(x);
      double e = __builtin_fma(-x, y, 1.0); y = __builtin_fma(y, e, y);
      e = __builtin_fma(-x, y, 1.0); y = __builtin_fma(y, e, y);
      x = y + 1.5;
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x + blockIdx.x * blockDim.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// divergent exec region per step: s_and_saveexec / s_cbranch_execz / s_or
__global__ void exec_regions(double* out, long long* cyc, double a, int m) {
  double x = a + threadIdx.x;
  const long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < REP / 4; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (((threadIdx.x + i + u) & m) == 0) { x = __builtin_fma(x, 1.0000001, 0.5); asm volatile("" : "+v"(x)); }
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x + blockIdx.x * blockDim.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double* out; long long* cyc; float* outf;
  hipMalloc(&out, 1 << 20); hipMalloc(&outf, 1 << 20); hipMalloc(&cyc, 4096 * 8);
  std::vector<long long> h(4096);
  auto report = [&](const char* name, int blocks, double per) {
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    long long s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
    printf("%-58s %7.2f cycles per %s\n", name, (double)s / blocks / per, "instruction / step");
  };
  for (int waves = 1; waves <= 2; ++waves) {
    const int blocks = 256 * 4 * waves;   // one (two) wavefront(s) per SIMD
    printf("--- %d wavefront(s) per SIMD (%d single-wave workgroups)\n", waves, blocks);
#define RUN(k, name, per, ...) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, __VA_ARGS__); report(name, blocks, per)
    RUN(fma_f64<1>, "v_fma_f64, 1 dependent chain", REP, out, cyc, 1.0, 0.999);
    RUN(fma_f64<2>, "v_fma_f64, 2 independent chains", REP, out, cyc, 1.0, 0.999);
    RUN(fma_f64<4>, "v_fma_f64, 4 independent chains", REP, out, cyc, 1.0, 0.999);
    RUN(fma_f64<8>, "v_fma_f64, 8 independent chains", REP, out, cyc, 1.0, 0.999);
    RUN((fma_f64_masked<8, 32>), "v_fma_f64, 8 independent chains, 32 of 64 lanes active", REP, out, cyc, 1.0, 0.999);
    RUN((fma_f64_masked<8, 16>), "v_fma_f64, 8 independent chains, 16 of 64 lanes active", REP, out, cyc, 1.0, 0.999);
    RUN((fma_f64_masked<1, 16>), "v_fma_f64, 1 dependent chain, 16 of 64 lanes active", REP, out, cyc, 1.0, 0.999);
    RUN(fma_f32<1>, "v_fma_f32, 1 dependent chain", REP, outf, cyc, 1.0f, 0.999f);
    RUN(fma_f32<4>, "v_fma_f32, 4 independent chains", REP, outf, cyc, 1.0f, 0.999f);
    RUN(fma_f32<8>, "v_fma_f32, 8 independent chains", REP, outf, cyc, 1.0f, 0.999f);
    RUN(dpp_min, "argmin step (2 v_mov_dpp + v_min_f64 + v_add_f64), dependent", REP, out, cyc, 1.0);
    RUN(lds_rt, "LDS write -> fence -> read of another lane + add, dependent", REP, out, cyc, 1.0);
    RUN(rcp_chain, "v_rcp_f64 + 2 Newton steps + add (6 instr), dependent", REP, out, cyc, 1.0);
    RUN(exec_regions, "exec region (saveexec, branch, fma, restore), half the lanes", REP, out, cyc, 1.0, 1);
  }
  return 0;
}

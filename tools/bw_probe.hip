// Bandwidth probes for the access patterns of the WBC kernels (known-good references on the same device,
// cdna_hip_programming.md rule 10): linear copy/write at 8 and 16 B/lane, and the component-major
// "443 rows x 128-byte segments per wave" store pattern of the dynamics sweep with no arithmetic at all.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

__global__ void k_copy8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ void k_write8(double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (double)i;
}
__global__ void k_write16(double2* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = make_double2((double)i, 1.0);
}
__global__ void k_read8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0;
  if (i < n) acc = in[i];
  if (acc == 123.456) out[0] = acc;
}
// sweep-like: lane = (state, leg); each lane writes ROWS/4 rows (leg-strided) of 8 B: 128-byte segments per row
template <int ROWS, int XCD>
__global__ __launch_bounds__(64) void k_soa_write(double* __restrict__ out, unsigned N) {
  unsigned bid = blockIdx.x;
  if (XCD) { const unsigned nb = gridDim.x; bid = (bid % 8) * (nb / 8) + bid / 8; }
  const unsigned gid = bid * 64 + threadIdx.x;
  const unsigned leg = gid & 3, s = gid >> 2;
  if (s >= N) return;
  const double v = (double)gid;
#pragma unroll 8
  for (int r = 0; r < ROWS / 4; ++r) {
    const unsigned comp = 4 * r + leg;
    *(double*)((char*)out + (size_t)((comp * N + s) * 8u)) = v + r;
  }
}
// same rows, but lane order inside the wave is leg-major: lanes 16*leg .. 16*leg+15 hold 16 consecutive states of
// ONE component row -> every 16-lane group writes one contiguous 128-byte line
template <int ROWS>
__global__ __launch_bounds__(64) void k_soa_write_legmajor(double* __restrict__ out, unsigned N) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned leg = lane >> 4;
  const unsigned s = blockIdx.x * 16 + (lane & 15);
  if (s >= N) return;
  const double v = (double)lane;
#pragma unroll 8
  for (int r = 0; r < ROWS / 4; ++r) {
    const unsigned comp = 4 * r + leg;
    *(double*)((char*)out + (size_t)((comp * N + s) * 8u)) = v + r;
  }
}
// the same leg-major lanes, but the rows are tiled: [tile of T states][component][T] -- every tile is one contiguous
// ROWS*T*8-byte region, so the workgroups in flight touch a few regions instead of ROWS streams 8N bytes apart
template <int ROWS>
__global__ __launch_bounds__(64) void k_soa_write_tiled(double* __restrict__ out, unsigned N, unsigned T) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned leg = lane >> 4;
  const unsigned s = blockIdx.x * 16 + (lane & 15);
  if (s >= N) return;
  const unsigned tile = s / T, st = s % T;
  const double v = (double)lane;
  char* base = (char*)out + (size_t)tile * ROWS * T * 8u;
#pragma unroll 8
  for (int r = 0; r < ROWS / 4; ++r) {
    const unsigned comp = 4 * r + leg;
    *(double*)(base + (size_t)((comp * T + st) * 8u)) = v + r;
  }
}
// one component row per store instruction: the 64 lanes hold 64 consecutive states (512 contiguous bytes per instruction)
template <int ROWS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_soa_write_row64(double* __restrict__ out, unsigned N) {
  const unsigned s = blockIdx.x * BLOCK + threadIdx.x;
  if (s >= N) return;
  const double v = (double)threadIdx.x;
#pragma unroll 8
  for (int r = 0; r < ROWS / 4; ++r)
    *(double*)((char*)out + (size_t)((((unsigned)r * 4u + (blockIdx.y & 3u)) * N + s) * 8u)) = v + r;
}
// leg-major lanes as above in 256-thread workgroups (4 wavefronts = 64 consecutive states)
template <int ROWS>
__global__ __launch_bounds__(256) void k_soa_write_legmajor256(double* __restrict__ out, unsigned N) {
  const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned leg = lane >> 4;
  const unsigned s = blockIdx.x * 64 + w * 16 + (lane & 15);
  if (s >= N) return;
  const double v = (double)lane;
#pragma unroll 8
  for (int r = 0; r < ROWS / 4; ++r) {
    const unsigned comp = 4 * r + leg;
    *(double*)((char*)out + (size_t)((comp * N + s) * 8u)) = v + r;
  }
}
// 16 bytes per lane: lane pairs (state s, s+1) merged: 32 lanes x 16 B cover the same 4 x 128 B per instruction pair
template <int ROWS>
__global__ __launch_bounds__(64) void k_soa_write_x4(double* __restrict__ out, unsigned N) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned leg = lane & 3, sl = lane >> 2;       // interleaved order as the sweep kernel
  const unsigned s = blockIdx.x * 16 + (sl & ~1u);     // even state of the pair
  const unsigned odd = sl & 1;
  if (s >= N) return;
  const double v = (double)lane;
#pragma unroll 8
  for (int r = 0; r < ROWS / 8; ++r) {                 // each lane writes half the rows, two states at a time
    const unsigned comp = 8 * r + 2 * leg + odd;
    *(double2*)((char*)out + (size_t)((comp * N + s) * 8u)) = make_double2(v + r, v - r);
  }
}
// The footprint of a whole kernel of this build, reads AND writes, nothing else: leg-major lanes (lane = 16 leg + j, as every dynamics body), 8 bytes per lane
// and row (an fp64 state, or an fp32 PAIR of states: the packed lane type), a lane reads the rows r = leg (mod 4) of `rin` input rows, then writes its share of
// `rout` output rows; what it read is folded into what it writes.  Rows are E elements apart (E = states, or pairs of fp32 states).  One-wavefront workgroups.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_soa_rw(const double* __restrict__ in, double* __restrict__ out, unsigned E, int rin, int rout) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned leg = lane >> 4;
  const unsigned s = (blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 16 + (lane & 15);
  if (s >= E) return;
  double acc = 0;
#pragma unroll 8
  for (int r = (int)leg; r < rin; r += 4) acc += *(const double*)((const char*)in + (size_t)(((unsigned)r * E + s) * 8u));
#pragma unroll 8
  for (int r = (int)leg; r < rout; r += 4) *(double*)((char*)out + (size_t)(((unsigned)r * E + s) * 8u)) = acc + r;
}
template <class F> float timeit(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}
// `bw_probe.bin rw <bytes per scalar: 4 | 8> <input rows> <output rows> <states>`: one line "<GB/s> <us>" for that footprint (bench.py: roofline.frac_of_pattern_ceiling)
static int probe_rw(int scalar, int rin, int rout, unsigned N, bool quiet) {
  const unsigned E = scalar == 4 ? (N + 1) / 2 : N;   // fp32: a lane holds a pair of states
  double *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, (size_t)(rin > 0 ? rin : 1) * E * 8) != hipSuccess || hipMalloc(&b, (size_t)(rout > 0 ? rout : 1) * E * 8) != hipSuccess) return 1;
  hipMemset(a, 0, (size_t)(rin > 0 ? rin : 1) * E * 8);
  // the better of one-wavefront workgroups (small batches: one per CU and more) and 256-thread workgroups (large batches: what the sweep launches there)
  const float t64 = timeit([&] { hipLaunchKernelGGL(k_soa_rw<64>, dim3((E + 15) / 16), dim3(64), 0, 0, a, b, E, rin, rout); }, 50);
  const float t256 = timeit([&] { hipLaunchKernelGGL(k_soa_rw<256>, dim3((E + 63) / 64), dim3(256), 0, 0, a, b, E, rin, rout); }, 50);
  const float t = t64 < t256 ? t64 : t256;
  const double bytes = (double)(rin + rout) * N * scalar;
  if (quiet) printf("%.1f %.3f\n", bytes / t / 1e6, t * 1e3);
  else printf("rw pattern %3d in + %3d out rows x %6u %s states : %.1f us  %.0f GB/s\n", rin, rout, N, scalar == 4 ? "fp32" : "fp64", t * 1e3, bytes / t / 1e6);
  hipFree(a); hipFree(b);
  return 0;
}
int main(int argc, char** argv) {
  if (argc == 6 && std::string(argv[1]) == "rw") return probe_rw(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), (unsigned)atol(argv[5]), true);
  // the footprints this build's rooflines are argued on (DESIGN.md sections 4, 7, 9): reads + writes of the whole kernel
  probe_rw(4, 138, 465, 32768, false);    // tile tick, fp32, observer on: in 78 + 60, out M, h, Jc 405 + tau, f 24 + observer state 36 = 603 words
  probe_rw(4, 138, 465, 262144, false);
  probe_rw(4, 37, 406, 32768, false);     // the dynamics stage alone: 443 words
  probe_rw(4, 97, 511, 32768, false);     // sweep_obs as launched (PMC: 608 words incl. the step workspace)
  probe_rw(8, 78, 429, 4096, false);      // one-launch tick, fp64, observer off: 507 words
  probe_rw(8, 138, 465, 4096, false);     // ... observer on: 603
  probe_rw(8, 37, 406, 262144, false);    // dyn_sweep, fp64, 262 144 states: 443
  probe_rw(8, 78, 429, 28672, false);     // tile tick, fp64, observer off, one round of workgroups
  const size_t bytes = (size_t)1 << 30;
  double *a, *b;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  size_t n8 = bytes / 8, n16 = bytes / 16;
  float t;
  t = timeit([&] { hipLaunchKernelGGL(k_copy8, dim3((n8 + 255) / 256), dim3(256), 0, 0, a, b, n8); }, 10);
  printf("copy  8B/lane 1GiB->1GiB : %.1f us  %.0f GB/s (r+w)\n", t * 1e3, 2 * bytes / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL(k_read8, dim3((n8 + 255) / 256), dim3(256), 0, 0, a, b, n8); }, 10);
  printf("read  8B/lane 1GiB       : %.1f us  %.0f GB/s\n", t * 1e3, bytes / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL(k_write8, dim3((n8 + 255) / 256), dim3(256), 0, 0, b, n8); }, 10);
  printf("write 8B/lane 1GiB       : %.1f us  %.0f GB/s\n", t * 1e3, bytes / t / 1e6);
  t = timeit([&] { hipLaunchKernelGGL(k_write16, dim3((n16 + 255) / 256), dim3(256), 0, 0, (double2*)b, n16); }, 10);
  printf("write 16B/lane 1GiB      : %.1f us  %.0f GB/s\n", t * 1e3, bytes / t / 1e6);
  for (unsigned N : {4096u, 32768u, 262144u}) {
    const size_t wb = (size_t)444 * N * 8;
    unsigned blocks = (N * 4 + 63) / 64;
    t = timeit([&] { hipLaunchKernelGGL((k_soa_write<444, 0>), dim3(blocks), dim3(64), 0, 0, b, N); }, 20);
    printf("SoA 444 rows N=%6u      : %.1f us  %.0f GB/s\n", N, t * 1e3, wb / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL((k_soa_write<444, 1>), dim3(blocks), dim3(64), 0, 0, b, N); }, 20);
    printf("SoA 444 rows N=%6u xcd  : %.1f us  %.0f GB/s\n", N, t * 1e3, wb / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL((k_soa_write_legmajor<444>), dim3((N + 15) / 16), dim3(64), 0, 0, b, N); }, 20);
    printf("SoA 444 rows N=%6u legmajor lanes : %.1f us  %.0f GB/s\n", N, t * 1e3, wb / t / 1e6);
    for (unsigned T : {64u, 1024u, 4096u, 16384u}) {
      if (T > N) continue;
      t = timeit([&] { hipLaunchKernelGGL((k_soa_write_tiled<444>), dim3((N + 15) / 16), dim3(64), 0, 0, b, N, T); }, 20);
      printf("SoA 444 rows N=%6u legmajor lanes, tiles of %5u states : %.1f us  %.0f GB/s\n", N, T, t * 1e3, wb / t / 1e6);
    }
    t = timeit([&] { hipLaunchKernelGGL((k_soa_write_row64<444, 64>), dim3((N + 63) / 64, 4), dim3(64), 0, 0, b, N); }, 20);
    printf("SoA 444 rows N=%6u one row x 64 states per instruction, 64-thread wg : %.1f us  %.0f GB/s\n", N, t * 1e3, wb / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL((k_soa_write_row64<444, 256>), dim3((N + 255) / 256, 4), dim3(256), 0, 0, b, N); }, 20);
    printf("SoA 444 rows N=%6u one row x 64 states per instruction, 256-thread wg: %.1f us  %.0f GB/s\n", N, t * 1e3, wb / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL((k_soa_write_legmajor256<444>), dim3((N + 63) / 64), dim3(256), 0, 0, b, N); }, 20);
    printf("SoA 444 rows N=%6u legmajor lanes, 256-thread wg : %.1f us  %.0f GB/s\n", N, t * 1e3, wb / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL((k_soa_write_x4<440>), dim3((N + 15) / 16), dim3(64), 0, 0, b, N); }, 20);
    printf("SoA 440 rows N=%6u 16B/lane pairs : %.1f us  %.0f GB/s\n", N, t * 1e3, (size_t)440 * N * 8 / t / 1e6);
  }
  return 0;
}

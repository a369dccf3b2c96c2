// A/B that BASELINE.json's north_star and SURVEY.md section 7 ask for: the batched small products inside the CRBA
// mass-matrix assembly on the MFMA matrix cores against the per-lane FMA form the sweep uses.
//
// The unit of work is the 3x3 congruence  C = E S E^T  (E a joint rotation, S a symmetric rotational inertia): the CRBA
// return sweep does one per joint (composite inertia into the parent frame), i.e. 3 per leg, 12 per state; the base-leg
// 6x3 blocks are products of the same shape.  In the sweep a LANE owns (leg, state) and holds E and S in its own
// registers, so the congruence is 45 FMA/MUL instructions per lane with no cross-lane traffic.
//
// v_mfma_f64_4x4x4_4b_f64 multiplies FOUR independent 4x4 blocks per instruction (one A, one B and one D element per
// lane): a 3x3 product padded to 4x4x4 uses 27 of its 64 multiply-adds.  Three variants are timed, all on 64 states per
// wavefront per step, data in registers, 8 wavefronts per SIMD-group worth of independent work:
//   fma          per-lane congruence as in dyn_sweep.hip.hpp (lane = one state)
//   mfma         two MFMAs per four states (T = E S, C = T E^T) with the operands ALREADY in the MFMA lane layout -- the
//                matrix cores' best case, unreachable for the sweep (its operands are produced per lane)
//   mfma+relayout the same, operands produced per lane as in the sweep: through LDS into the MFMA layout and back
// Prints congruences per second for each and checks all three against a host computation.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/mfma_probe.bin && tools/mfma_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// ---- per-lane form: in [15][n] component-major (E 9, S xx xy xz yy yz zz), out [6][n]
__device__ __forceinline__ void congr_lane(const double* E, const double* S, double* C) {
  // T = E S (S symmetric), C = T E^T (symmetric: 6 entries)
  double T[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    T[3 * i + 0] = E[3 * i] * S[0] + E[3 * i + 1] * S[1] + E[3 * i + 2] * S[2];
    T[3 * i + 1] = E[3 * i] * S[1] + E[3 * i + 1] * S[3] + E[3 * i + 2] * S[4];
    T[3 * i + 2] = E[3 * i] * S[2] + E[3 * i + 1] * S[4] + E[3 * i + 2] * S[5];
  }
  int o = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j) C[o++] = T[3 * i] * E[3 * j] + T[3 * i + 1] * E[3 * j + 1] + T[3 * i + 2] * E[3 * j + 2];
}

__global__ __launch_bounds__(256) void k_fma(const double* __restrict__ in, double* __restrict__ out, int n, int reps) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  double E[9], S[6], acc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 9; ++c) E[c] = in[(size_t)c * n + s];
#pragma unroll
  for (int c = 0; c < 6; ++c) S[c] = in[(size_t)(9 + c) * n + s];
  for (int r = 0; r < reps; ++r) {
    double C[6];
    congr_lane(E, S, C);
#pragma unroll
    for (int c = 0; c < 6; ++c) { acc[c] += C[c]; S[c] = S[c] * 0.999 + 1e-3 * C[c]; }   // the next congruence depends on this one, as in the tree sweep
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) out[(size_t)c * n + s] = acc[c];
}

// ---- MFMA form.  Lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 (probed with one-hot operands; the check in main pins
// it): block = (lane / 4) % 4;  A[i][k] in lane 16 k + 4 block + i;  B[k][j] in lane 16 k + 4 block + j;  D[i][j] in lane
// 16 i + 4 block + j.  D therefore has the layout of a B operand: products chain without a transposition when the
// intermediate is the RIGHT factor.  One wavefront step = 16 quads of states (64 states), one operand element per lane.
__device__ __forceinline__ double mfma444(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// operands pre-arranged in the MFMA layout in memory: inA [16 quads][64 lanes] = E[lane % 4][lane / 16] -- which is at
// once "E as A operand" (A[i][k], i = lane % 4) and "E^T as B operand" (B[k][j] = E[j][k], j = lane % 4) -- and inS = S as A
// operand.  C = E (S E^T): the first MFMA forms S E^T, whose D layout is the B layout the second MFMA (A = E) needs.
__global__ __launch_bounds__(256) void k_mfma(const double* __restrict__ inA, const double* __restrict__ inS, double* __restrict__ out, int nwaves, int reps) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= nwaves) return;
  double ea[16], sb[16], acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) { ea[q] = inA[((size_t)wave * 16 + q) * 64 + lane]; sb[q] = inS[((size_t)wave * 16 + q) * 64 + lane]; acc[q] = 0; }
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const double t = mfma444(sb[q], ea[q], 0.0);     // S E^T   (A = S, B = E^T; result in the D = B layout)
      const double c = mfma444(ea[q], t, 0.0);         // C = E (S E^T)
      acc[q] += c;
      sb[q] = sb[q] * 0.999 + 1e-3 * c;                // dependent chain like the per-lane variant (C is symmetric)
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) out[((size_t)wave * 16 + q) * 64 + lane] = acc[q];
}

// operands produced PER LANE (lane = state), as in the sweep: stage E, S through LDS into the MFMA layout, multiply, and
// bring C back to the owning lane (the sweep continues with per-lane work on it)
__global__ __launch_bounds__(256) void k_mfma_relayout(const double* __restrict__ in, double* __restrict__ out, int n, int reps) {
  __shared__ double lds[4][64 * 17];   // per wave: [state in wave][16 + 1 pad]
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (s >= n) return;
  double E[9], S[6], acc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 9; ++c) E[c] = in[(size_t)c * n + s];
#pragma unroll
  for (int c = 0; c < 6; ++c) S[c] = in[(size_t)(9 + c) * n + s];
  double* my = &lds[w][lane * 17];
  const int blk = (lane >> 2) & 3, ii = lane & 3, kk = lane >> 4;
  for (int r = 0; r < reps; ++r) {
    // E as A operand (= E^T as B operand): element [i][k] of state 4q + blk lives in lane 16 k + 4 blk + i
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) my[4 * i + k] = E[3 * i + k];
    my[3] = my[7] = my[11] = my[12] = my[13] = my[14] = my[15] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    double ea[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) ea[q] = lds[w][(4 * q + blk) * 17 + 4 * ii + kk];
    __builtin_amdgcn_wave_barrier();
    const int sidx[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) my[4 * i + k] = S[sidx[3 * i + k]];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    double cq[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const double sa = lds[w][(4 * q + blk) * 17 + 4 * ii + kk];     // S as A operand: S[i][k], i = lane % 4, k = lane / 16
      const double t = mfma444(sa, ea[q], 0.0);
      cq[q] = mfma444(ea[q], t, 0.0);
    }
    __builtin_amdgcn_wave_barrier();
    // C back to the owning lanes: D[i][j] of state 4q + blk sits in lane 16 i + 4 blk + j
#pragma unroll
    for (int q = 0; q < 16; ++q) lds[w][(4 * q + blk) * 17 + 4 * kk + ii] = cq[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    const double C[6] = {my[0], my[1], my[2], my[5], my[6], my[10]};
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 6; ++c) { acc[c] += C[c]; S[c] = S[c] * 0.999 + 1e-3 * C[c]; }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) out[(size_t)c * n + s] = acc[c];
}

static void host_ref(const std::vector<double>& in, int n, int s, int reps, double* acc) {
  double E[9], S[6];
  for (int c = 0; c < 9; ++c) E[c] = in[(size_t)c * n + s];
  for (int c = 0; c < 6; ++c) S[c] = in[(size_t)(9 + c) * n + s];
  for (int c = 0; c < 6; ++c) acc[c] = 0;
  for (int r = 0; r < reps; ++r) {
    const double Sm[9] = {S[0], S[1], S[2], S[1], S[3], S[4], S[2], S[4], S[5]};
    double T[9], C[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { T[3 * i + j] = 0; for (int k = 0; k < 3; ++k) T[3 * i + j] += E[3 * i + k] * Sm[3 * k + j]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { C[3 * i + j] = 0; for (int k = 0; k < 3; ++k) C[3 * i + j] += T[3 * i + k] * E[3 * j + k]; }
    const double Cs[6] = {C[0], C[1], C[2], C[4], C[5], C[8]};
    for (int c = 0; c < 6; ++c) { acc[c] += Cs[c]; S[c] = S[c] * 0.999 + 1e-3 * Cs[c]; }
  }
}

int main() {
  const int n = 256 * 8 * 64 * 4;   // states: 8 waves per SIMD-quad of every CU, four rounds
  const int reps = 64, check_reps = 3;
  std::vector<double> in((size_t)15 * n);
  srand(7);
  for (int s = 0; s < n; ++s) {   // a rotation-like E (not exactly orthogonal: irrelevant here) and an SPD-like S
    for (int c = 0; c < 9; ++c) in[(size_t)c * n + s] = (rand() / (double)RAND_MAX - 0.5) * 1.2;
    for (int c = 0; c < 6; ++c) in[(size_t)(9 + c) * n + s] = (rand() / (double)RAND_MAX) * ((c == 0 || c == 3 || c == 5) ? 1.0 : 0.1);
  }
  // MFMA-layout images: quad q of wave w holds states 64 w + 4 q + blk
  const int nwaves = n / 64;
  std::vector<double> inA((size_t)n * 16), inS((size_t)n * 16);
  const int sidx[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
  for (int w = 0; w < nwaves; ++w)
    for (int q = 0; q < 16; ++q)
      for (int lane = 0; lane < 64; ++lane) {
        const int blk = (lane >> 2) & 3, lo = lane & 3, hi = lane >> 4, s = 64 * w + 4 * q + blk;
        const size_t o = ((size_t)w * 16 + q) * 64 + lane;
        inA[o] = (lo < 3 && hi < 3) ? in[(size_t)(3 * lo + hi) * n + s] : 0.0;                 // E[lo][hi]
        inS[o] = (lo < 3 && hi < 3) ? in[(size_t)(9 + sidx[3 * lo + hi]) * n + s] : 0.0;        // S[lo][hi]
      }
  double *dIn, *dA, *dS, *dOut, *dOutM;
  HK(hipMalloc(&dIn, in.size() * 8)); HK(hipMalloc(&dA, inA.size() * 8)); HK(hipMalloc(&dS, inS.size() * 8));
  HK(hipMalloc(&dOut, (size_t)6 * n * 8)); HK(hipMalloc(&dOutM, (size_t)16 * n * 8));
  HK(hipMemcpy(dIn, in.data(), in.size() * 8, hipMemcpyHostToDevice));
  HK(hipMemcpy(dA, inA.data(), inA.size() * 8, hipMemcpyHostToDevice));
  HK(hipMemcpy(dS, inS.data(), inS.size() * 8, hipMemcpyHostToDevice));
  // ---- correctness (also pins the lane layout assumed above)
  std::vector<double> o1((size_t)6 * n), o2((size_t)16 * n), o3((size_t)6 * n);
  k_fma<<<n / 256, 256>>>(dIn, dOut, n, check_reps); HK(hipMemcpy(o1.data(), dOut, o1.size() * 8, hipMemcpyDeviceToHost));
  k_mfma<<<n / 256, 256>>>(dA, dS, dOutM, nwaves, check_reps); HK(hipMemcpy(o2.data(), dOutM, o2.size() * 8, hipMemcpyDeviceToHost));
  k_mfma_relayout<<<n / 256, 256>>>(dIn, dOut, n, check_reps); HK(hipMemcpy(o3.data(), dOut, o3.size() * 8, hipMemcpyDeviceToHost));
  double e1 = 0, e2 = 0, e3 = 0;
  const int cidx[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
  for (int s = 0; s < n; s += 997) {
    double ref[6];
    host_ref(in, n, s, check_reps, ref);
    const int w = s / 64, q = (s % 64) / 4, blk = s % 4;
    for (int c = 0; c < 6; ++c) {
      e1 = fmax(e1, fabs(o1[(size_t)c * n + s] - ref[c]));
      e3 = fmax(e3, fabs(o3[(size_t)c * n + s] - ref[c]));
      // k_mfma accumulates C in the D layout: C[i][j] at lane 16 i + 4 blk + j
      const int lane = 16 * cidx[c][0] + 4 * blk + cidx[c][1];
      e2 = fmax(e2, fabs(o2[((size_t)w * 16 + q) * 64 + lane] - ref[c]));
    }
  }
  std::printf("max |error| vs host: fma %.2e  mfma %.2e  mfma+relayout %.2e\n", e1, e2, e3);
  if (!(e1 < 1e-9 && e2 < 1e-9 && e3 < 1e-9)) { std::printf("MISMATCH (lane layout assumption wrong?)\n"); return 2; }
  // ---- timing
  hipEvent_t a, b; HK(hipEventCreate(&a)); HK(hipEventCreate(&b));
  auto timeit = [&](auto&& launch) {
    launch(); HK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) { HK(hipEventRecord(a)); launch(); HK(hipEventRecord(b)); HK(hipEventSynchronize(b)); float ms; HK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return (double)best * 1e-3;
  };
  const double total = (double)n * reps;
  const double t1 = timeit([&] { k_fma<<<n / 256, 256>>>(dIn, dOut, n, reps); });
  const double t2 = timeit([&] { k_mfma<<<n / 256, 256>>>(dA, dS, dOutM, nwaves, reps); });
  const double t3 = timeit([&] { k_mfma_relayout<<<n / 256, 256>>>(dIn, dOut, n, reps); });
  std::printf("3x3 congruences C = E S E^T, fp64, %d states x %d dependent steps\n", n, reps);
  std::printf("  per-lane FMA (the sweep's form)              %8.3f ms  %7.1f G congruences/s  1.00x\n", t1 * 1e3, total / t1 / 1e9);
  std::printf("  MFMA 4x4x4, operands already in MFMA layout  %8.3f ms  %7.1f G congruences/s  %.2fx\n", t2 * 1e3, total / t2 / 1e9, t1 / t2);
  std::printf("  MFMA 4x4x4 + LDS re-layout in and out        %8.3f ms  %7.1f G congruences/s  %.2fx\n", t3 * 1e3, total / t3 / 1e9, t1 / t3);
  return 0;
}

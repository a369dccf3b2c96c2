// What lets a second workgroup onto a CU that still holds part of the first?  Six-wavefront workgroups like the fused tick's:
// wavefronts 4,5 leave after T/2, wavefronts 0..3 after T (spinning on the 100 MHz clock), LDS size and register count
// as parameters.  With 512 workgroups on 256 CUs the kernel takes 2T if the second round waits for whole workgroups and
// ~1.5T if it moves in as soon as six wave slots are free.
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ char dyn_lds[];
template <int REGS>
__global__ __launch_bounds__(384) void spin(unsigned ticks, int* sink, int stagger) {
  if (REGS > 200) asm volatile("v_mov_b32 v231, 0" ::: "v231");
  else if (REGS > 128) asm volatile("v_mov_b32 v163, 0" ::: "v163");   // 164 -> 168 registers: three wavefronts per SIMD
  else asm volatile("v_mov_b32 v120, 0" ::: "v120");
  const unsigned w = threadIdx.x >> 6;
  dyn_lds[threadIdx.x] = (char)w;
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  // STAGGER: the four long wavefronts leave at 0.6 / 0.65 / 0.8 / 1.0 T like QP wavefronts with different iteration counts
  const unsigned lim = w >= 4 ? ticks / 2 : (stagger ? (w == 0 ? ticks * 6 / 10 : w == 1 ? ticks * 65 / 100 : w == 2 ? ticks * 8 / 10 : ticks) : ticks);
  while (wall_clock64() - r0 < lim) { __builtin_amdgcn_s_sleep(1); }
  if (t0 == 12345 && sink) sink[0] = dyn_lds[5];
}
template <int REGS> float run(unsigned grid, unsigned lds, unsigned ticks, int stagger = 0) {
  hipFuncSetAttribute((const void*)spin<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(spin<REGS>, dim3(grid), dim3(384), lds, 0, ticks, (int*)nullptr, stagger);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin<REGS>, dim3(grid), dim3(384), lds, 0, ticks, (int*)nullptr, stagger);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 20 * 1e3f;
}
int main() {
  const unsigned ticks = 2000;   // 20 us
  for (unsigned lds : {16384u, 40960u, 49152u, 65536u, 73728u, 78848u, 80272u, 81920u, 98192u}) {
    printf("LDS %6u B: 232 VGPRs: 256 wg %.1f us, 512 wg %.1f us | 120 VGPRs: 256 wg %.1f us, 512 wg %.1f us\n", lds,
           run<232>(256, lds, ticks), run<232>(512, lds, ticks), run<120>(256, lds, ticks), run<120>(512, lds, ticks));
  }
  for (unsigned lds : {16384u, 80272u, 98192u})
    printf("staggered, LDS %6u B: 232 VGPRs: 256 wg %.1f us, 512 wg %.1f us | 120 VGPRs: 256 wg %.1f us, 512 wg %.1f us\n", lds,
           run<232>(256, lds, ticks, 1), run<232>(512, lds, ticks, 1), run<120>(256, lds, ticks, 1), run<120>(512, lds, ticks, 1));
  // round 6: six-wavefront workgroups at 168 registers (three wavefronts per SIMD): do TWO of them share a CU (12 wavefronts = 3 + 3 + 3 + 3)?
  for (unsigned lds : {16384u, 49152u, 73728u})
    printf("LDS %6u B: 168 VGPRs: 256 wg %.1f us, 512 wg %.1f us\n", lds, run<168>(256, lds, ticks), run<168>(512, lds, ticks));
  return 0;
}

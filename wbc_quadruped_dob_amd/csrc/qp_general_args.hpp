// Plain-data arguments of qp_general_kernel (qp_general.hip.hpp): shared by the kernel unit and the host side.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace wbc {

constexpr int QPG_MAXN = 36, QPG_MAXM = 64;   // (a constraint per lane: 64 is the wavefront)

template <class T> struct QpGeneralArgs {
  size_t N;
  int n, m, meq, max_iter;
  T tol;
  const T* H; const T* g; const T* C; const T* d;   // [N][n*n], [N][n], [N][m*n], [N][m]  (problem-major)
  T* x; T* lambda;                                   // [N][n], [N][m] (lambda may be null)
  int* status; int* iters;                           // [N] (iters may be null)
};

__host__ __device__ inline int qpg_ld(int n) { return n | 1; }
// LDS scalars per QP
__host__ __device__ inline int qpg_lds_scalars(int n, int m) { const int ld = qpg_ld(n); return 2 * n * ld + m * ld + 7 * (n + 1) + (m + 1); }

}  // namespace wbc

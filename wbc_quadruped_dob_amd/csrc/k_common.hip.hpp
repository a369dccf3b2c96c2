// Shared by the kernel translation units: the one launch helper.  Inside an instrumented span the dispatch carries
// the span's start/stop events (hipExtLaunchKernelGGL), so the measured time is the dispatch's own begin/end
// timestamps -- what rocprofv3 reports -- and not an event pair recorded around it.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "launch.hpp"

#ifndef WBC_SCALAR
#error "compile the kernel units with -DWBC_SCALAR=double or -DWBC_SCALAR=float"
#endif

#define WBC_KLAUNCH(L, kern, grid, block, ...)                                                                  \
  do {                                                                                                          \
    if ((L).ev_start) hipExtLaunchKernelGGL(kern, grid, block, 0, (L).st, (L).ev_start, (L).ev_stop, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, grid, block, 0, (L).st, __VA_ARGS__);                                         \
  } while (0)

#define WBC_KLAUNCH_SMEM(L, kern, grid, block, smem, ...)                                                       \
  do {                                                                                                          \
    if ((L).ev_start) hipExtLaunchKernelGGL(kern, grid, block, smem, (L).st, (L).ev_start, (L).ev_stop, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, grid, block, smem, (L).st, __VA_ARGS__);                                      \
  } while (0)

namespace wbc {
using Scalar = WBC_SCALAR;
}  // namespace wbc

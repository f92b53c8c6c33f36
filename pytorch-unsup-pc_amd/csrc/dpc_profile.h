// Opt-in per-kernel timing of the fused path (off by default; bench.py switches it on for a short eager pass).
// When on, every fused-path launch is bracketed by two hipEvents recorded on the launch stream; the caller
// synchronises the stream and then reads (name, milliseconds) pairs.  Not usable during graph capture.
#pragma once
#include <hip/hip_runtime.h>

void dpc_prof_before(const char* name, hipStream_t st);
void dpc_prof_after(hipStream_t st);

#ifdef DPC_LAUNCH_TWICE
// timing experiment: every kernel is launched twice back to back (idempotent kernels only); the second launch finds its
// code in the instruction cache, its record carries the suffix "#2"
#define DPC_LAUNCH(name, kernel, grid, block, lds, st, ...)          \
  do {                                                               \
    dpc_prof_before(name, st);                                       \
    hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);   \
    dpc_prof_after(st);                                              \
    dpc_prof_before(name "#2", st);                                  \
    hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);   \
    dpc_prof_after(st);                                              \
  } while (0)
#else
#define DPC_LAUNCH(name, kernel, grid, block, lds, st, ...)          \
  do {                                                               \
    dpc_prof_before(name, st);                                       \
    hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);   \
    dpc_prof_after(st);                                              \
  } while (0)
#endif

// Opt-in per-kernel timing of the fused path (off by default; bench.py switches it on for a short eager pass).
// When on, every fused-path launch is bracketed by two hipEvents recorded on the launch stream; the caller
// synchronises the stream and then reads (name, milliseconds) pairs.  Not usable during graph capture.
#pragma once
#include <hip/hip_runtime.h>

void dpc_prof_before(const char* name, hipStream_t st);
void dpc_prof_after(hipStream_t st);

#define DPC_LAUNCH(name, kernel, grid, block, lds, st, ...)          \
  do {                                                               \
    dpc_prof_before(name, st);                                       \
    hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);   \
    dpc_prof_after(st);                                              \
  } while (0)

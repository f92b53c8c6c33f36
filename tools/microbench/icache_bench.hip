// Straight-line vs looped VALU code: does a long fully-unrolled kernel pay for instruction fetch?
// Every wave executes NF dependent-free v_fma_f32 (8 accumulators); variant A is fully unrolled (8 B per instruction,
// NF*8 bytes of code), variant B loops over a 64-instruction body.  grid = 512 x 256 threads (2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NF>
__global__ __launch_bounds__(256) void k_straight(float* out, float a, float b) {
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
#pragma unroll
  for (int i = 0; i < NF; ++i) acc[i & 7] = __builtin_fmaf(acc[i & 7], a, b);
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  if (s == 123.456f) out[0] = s;
}
template <int NF>
__global__ __launch_bounds__(256) void k_loop(float* out, float a, float b) {
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
#pragma unroll 1
  for (int it = 0; it < NF / 64; ++it) {
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i & 7] = __builtin_fmaf(acc[i & 7], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  if (s == 123.456f) out[0] = s;
}
__global__ void k_other(float* out) { if (out[0] == 123.f) out[1] = 1.f; }
template <class F>
float time_it(F launch, int reps, bool interleave, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  hipDeviceSynchronize();
  float tot = 0;
  for (int i = 0; i < reps; ++i) {
    if (interleave) hipLaunchKernelGGL(k_other, dim3(1024), dim3(256), 0, 0, d);
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
  }
  return tot / reps * 1e3f;
}
int main() {
  float* d; hipMalloc(&d, 1024); hipMemset(d, 0, 1024);
  const dim3 g(512), b(256);
#define RUN(NF)                                                                                              \
  {                                                                                                          \
    float ts = time_it([&] { hipLaunchKernelGGL(k_straight<NF>, g, b, 0, 0, d, 1.0001f, 0.5f); }, 50, false, d); \
    float tl = time_it([&] { hipLaunchKernelGGL(k_loop<NF>, g, b, 0, 0, d, 1.0001f, 0.5f); }, 50, false, d);     \
    float ti = time_it([&] { hipLaunchKernelGGL(k_straight<NF>, g, b, 0, 0, d, 1.0001f, 0.5f); }, 50, true, d);  \
    printf("NF=%5d code=%6d B  straight %.2f us  loop %.2f us  straight-after-other-kernel %.2f us   (ideal 2 cyc/instr x 2 waves @2.4GHz: %.2f us)\n", \
           NF, NF * 8, ts, tl, ti, NF * 2.0 * 2 / 2400.0);                                                    \
  }
  RUN(256) RUN(1024) RUN(2048) RUN(4096) RUN(8192)
  return 0;
}

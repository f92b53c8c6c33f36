// Effective shader clock inside a short kernel: s_memtime (shader cycles) against s_memrealtime (100 MHz), for
// launches separated by idle gaps and for launches issued back to back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ __launch_bounds__(256) void k_spin(unsigned long long* out, int iters, float a, float b) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i & 7] = __builtin_fmaf(acc[i & 7], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (s == 123.456f) out[0] = 0;
}
int main() {
  unsigned long long *d, h[2 * 512];
  hipMalloc(&d, sizeof(h));
  auto report = [&](const char* what) {
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < 512; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
    printf("%-44s kernel body %.2f us, %.0f shader cycles -> %.0f MHz\n", what, real / 512 / 100.0, cyc / 512, cyc / real * 100.0);
  };
  for (int iters : {16, 64, 256}) {
    printf("iters=%d (%d fma per wave, 2 waves/SIMD)\n", iters, iters * 64);
    for (int rep = 0; rep < 3; ++rep) {
      usleep(20000);
      hipLaunchKernelGGL(k_spin, dim3(512), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
      hipDeviceSynchronize();
      report("  isolated launch after 20 ms idle:");
    }
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_spin, dim3(512), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    report("  last of 2000 back-to-back launches:");
  }
  return 0;
}

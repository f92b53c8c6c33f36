// Instruction fetch on gfx950: cycles per VALU wave-instruction per SIMD for a loop body of BODY independent v_fma_f32
// (8 accumulators), 4 waves per SIMD, versus the body size.  Straight-line code measured 4.1 cycles, a 64-instruction loop
// 2.4 (icache_bench.hip): how large may a loop body be before it is fetch-bound like straight-line code?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BODY>
__global__ __launch_bounds__(256) void k_loop(float* out, float a, float b, int iters) {
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < BODY; ++i) acc[i & 7] = __builtin_fmaf(acc[i & 7], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  if (s == 123.456f) out[0] = s;
}
template <int BODY>
void run(float* d) {
  const int total = 1 << 20;  // instructions per wave and launch (rounded down to whole bodies)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto t = [&](int iters) {
    for (int i = 0; i < 3; ++i) k_loop<BODY><<<1024, 256>>>(d, 1.0001f, 0.5f, iters);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k_loop<BODY><<<1024, 256>>>(d, 1.0001f, 0.5f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 100.f;  // us per launch
  };
  const float t1 = t(total / BODY), t2 = t(2 * (total / BODY));
  printf("body %5d instr (%6d B): %.2f cycles per wave-instruction per SIMD (4 waves/SIMD, 2.4 GHz assumed)\n", BODY, BODY * 8,
         (t2 - t1) * 2400.0 / ((double)(total / BODY) * BODY * 4.0));
}
int main() {
  float* d; hipMalloc(&d, 1024); hipMemset(d, 0, 1024);
  run<64>(d); run<1024>(d); run<1280>(d); run<1536>(d); run<1792>(d); run<2048>(d); run<4096>(d);
  return 0;
}

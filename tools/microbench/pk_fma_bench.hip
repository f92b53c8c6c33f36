// Is the 4.1-cycle issue rate of straight-line VALU code on gfx950 a per-INSTRUCTION cost?  Then v_pk_fma_f32 (two fp32
// FMAs per instruction) doubles the arithmetic of issue-bound straight-line code.  Straight-line streams of NF v_fma_f32 vs
// NF v_pk_fma_f32, 8 independent accumulators, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int NF>
__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b) {
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
__global__ __launch_bounds__(256) void k_pk(float* out, float a, float b) {
  f2 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f2{(float)(threadIdx.x + i), (float)(threadIdx.x - i)};
  const f2 aa = f2{a, a * 1.5f}, bb = f2{b, b + 1.f};
#pragma unroll
  for (int i = 0; i < NF; ++i) acc[i & 7] = __builtin_elementwise_fma(acc[i & 7], aa, bb);
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  if (s == 123.456f) out[0] = s;
}
template <class F> float t_us(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e0); for (int i = 0; i < 20; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 50.f;
}
int main() {
  float* d; hipMalloc(&d, 1024); hipMemset(d, 0, 1024);
  const float f1 = t_us([&] { k_fma<1024><<<1024, 256>>>(d, 1.0001f, 0.5f); }), f2_ = t_us([&] { k_fma<3072><<<1024, 256>>>(d, 1.0001f, 0.5f); });
  const float p1 = t_us([&] { k_pk<1024><<<1024, 256>>>(d, 1.0001f, 0.5f); }), p2 = t_us([&] { k_pk<3072><<<1024, 256>>>(d, 1.0001f, 0.5f); });
  printf("straight-line v_fma_f32   : %.2f cycles per instruction per SIMD\n", (f2_ - f1) * 2400.0 / (2048 * 4.0));
  printf("straight-line v_pk_fma_f32: %.2f cycles per instruction per SIMD (two FMAs each)\n", (p2 - p1) * 2400.0 / (2048 * 4.0));
  return 0;
}

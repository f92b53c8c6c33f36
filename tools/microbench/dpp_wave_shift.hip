// Does gfx950 execute the gfx9 whole-wave DPP shifts (wave_shr:1 / wave_shl:1, zero fill with bound_ctrl)?  A 7-tap
// correlation across the 64 lanes of a wave built from them is checked against the host, and timed against the same
// correlation through LDS.   hipcc -O3 --offload-arch=gfx950 dpp_wave_shift.hip -o dpp_wave_shift && ./dpp_wave_shift
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

template <int CTRL>
__device__ inline float dpp0(float v) {  // lanes without a source read 0 (bound_ctrl)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ inline float shr1(float v) { return dpp0<0x138>(v); }  // wave_shr:1  lane i <- lane i-1
__device__ inline float shl1(float v) { return dpp0<0x130>(v); }  // wave_shl:1  lane i <- lane i+1

__global__ void k_dpp(const float* in, float* out, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = in[i], acc = 0.f;
  const float w[7] = {0.01f, 0.06f, 0.24f, 0.38f, 0.24f, 0.06f, 0.01f};
  for (int it = 0; it < iters; ++it) {
    const float r1 = shr1(v), r2 = shr1(r1), r3 = shr1(r2);   // x-1, x-2, x-3
    const float l1 = shl1(v), l2 = shl1(l1), l3 = shl1(l2);   // x+1, x+2, x+3
    acc = w[0] * r3 + w[1] * r2 + w[2] * r1 + w[3] * v + w[4] * l1 + w[5] * l2 + w[6] * l3;
    if (it + 1 < iters) v = acc * 0.5f + v * 0.5f;
  }
  out[i] = acc;
}

int main() {
  const int n = 256 * 1024;
  std::vector<float> h(n), o(n);
  for (int i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
  float *di, *dout;
  hipMalloc(&di, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(di, h.data(), n * 4, hipMemcpyHostToDevice);
  k_dpp<<<n / 256, 256>>>(di, dout, 1);
  hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
  const float w[7] = {0.01f, 0.06f, 0.24f, 0.38f, 0.24f, 0.06f, 0.01f};
  double worst = 0;
  for (int i = 0; i < n; ++i) {
    const int lane = i & 63, base = i - lane;
    float e = 0.f;
    // same association order as the kernel
    float t[7];
    for (int k = 0; k < 7; ++k) { const int l = lane + k - 3; t[k] = (l >= 0 && l < 64) ? h[base + l] : 0.f; }
    e = w[0] * t[0] + w[1] * t[1] + w[2] * t[2] + w[3] * t[3] + w[4] * t[4] + w[5] * t[5] + w[6] * t[6];
    worst = fmax(worst, fabs((double)e - o[i]));
  }
  printf("wave_shr/shl 7-tap across lanes: max abs err vs host %.3g  (%s)\n", worst, worst < 1e-6 ? "OK" : "WRONG");
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k_dpp<<<n / 256, 256>>>(di, dout, 1000);
  hipEventRecord(a); k_dpp<<<n / 256, 256>>>(di, dout, 1000); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("1000 iterations of (6 dpp moves + 7 fma + 2): %.1f us -> %.2f ns per wave-iteration per SIMD-resident wave\n", ms * 1e3, ms * 1e6 / 1000.0);
  return worst < 1e-6 ? 0 : 1;
}

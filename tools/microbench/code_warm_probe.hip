// Probe (MI355X): what does a kernel of ~18 KB of straight-line code pay for instruction fetch when another kernel has
// streamed data through the L2s since its last launch, and does touching its CODE BYTES with ordinary loads (from a kernel
// that runs just before it) buy that back?
//   A   256 workgroups x 1024 threads, NF dependent-free FMAs, fully unrolled (8 B each); it publishes its own first and
//       last program counter (s_getpc_b64) so that others know where its code lives
//   B   streams 2 x 64 MB (read + write): evicts the 8 x 4 MB L2s
//   C   touches [pc_first, pc_last] of A with 128-byte-strided loads from workgroups on every XCD
// Timed with events around A only:  A after A | A after B | A after B, C | A after (B that touches at its end)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__device__ inline unsigned long long getpc() {
  unsigned long long pc;
  asm volatile("s_getpc_b64 %0" : "=s"(pc));
  return pc;
}

template <int NF>
__global__ __launch_bounds__(1024) void k_code(float* out, float a, float b, unsigned long long* tab) {
  if (blockIdx.x == 0 && threadIdx.x == 0) tab[0] = getpc();
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
#pragma unroll
  for (int i = 0; i < NF; ++i) acc[i & 7] = __builtin_fmaf(acc[i & 7], a, b);
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  if (s == 123.456f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) tab[1] = getpc();
}

// touch the 128-byte lines of [tab[0], tab[1]] from every XCD: workgroup w sits on XCD w % 8
__device__ inline void touch_code(const unsigned long long* tab, float* sink) {
  const unsigned long long lo = tab[0] & ~127ull, hi = tab[1];
  if (lo == 0 || hi <= lo || hi - lo > (1ull << 20)) return;
  const int nlines = (int)((hi - lo + 127) / 128);
  const int m = blockIdx.x / 8;  // m-th workgroup of its XCD
  if (threadIdx.x < 64) {
    const int line = m * 64 + threadIdx.x;
    if (line < nlines) {
      const float v = *reinterpret_cast<const volatile float*>(lo + (unsigned long long)line * 128);
      if (v == 123.456f) sink[1] = v;
    }
  }
}

__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4,
                                                const unsigned long long* tab, float* sink, int touch) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = src[i];
    v.x += 1.f;
    dst[i] = v;
  }
  if (touch) touch_code(tab, sink);
}

__global__ __launch_bounds__(64) void k_touch(const unsigned long long* tab, float* sink) { touch_code(tab, sink); }

int main() {
  float* d;
  CK(hipMalloc(&d, 1024));
  CK(hipMemset(d, 0, 1024));
  unsigned long long* tab;
  CK(hipMalloc(&tab, 64));
  CK(hipMemset(tab, 0, 64));
  const size_t bytes = 64ull << 20;
  float4 *src, *dst;
  CK(hipMalloc(&src, bytes));
  CK(hipMalloc(&dst, bytes));
  CK(hipMemset(src, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  constexpr int NF = 2304;
  auto A = [&] { hipLaunchKernelGGL(k_code<NF>, dim3(256), dim3(1024), 0, 0, d, 1.0001f, 0.5f, tab); };
  auto B = [&](int touch) { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, src, dst, bytes / 16, tab, d, touch); };
  auto C = [&] { hipLaunchKernelGGL(k_touch, dim3(64), dim3(64), 0, 0, tab, d); };
  A();
  CK(hipDeviceSynchronize());
  unsigned long long h[2];
  CK(hipMemcpy(h, tab, 16, hipMemcpyDeviceToHost));
  printf("code of A: %llu bytes between its first and last s_getpc (NF * 8 = %d)\n", h[1] - h[0], NF * 8);
  for (int rep = 0; rep < 3; ++rep) {
    for (int mode = 0; mode < 5; ++mode) {
      std::vector<float> us;
      for (int i = 0; i < 40; ++i) {
        if (mode == 0) A();
        if (mode == 1) B(0);
        if (mode == 2) { B(0); C(); }
        if (mode == 3) B(1);
        if (mode == 4) { B(0); C(); C(); }
        CK(hipEventRecord(e0));
        A();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us.push_back(ms * 1e3f);
      }
      std::sort(us.begin(), us.end());
      const char* names[] = {"A after A", "A after B (stream 128 MB)", "A after B, C (touch A's code)", "A after B that touches at its end",
                             "A after B, C, C"};
      printf("%-36s median %.2f us  min %.2f us\n", names[mode], us[us.size() / 2], us[0]);
    }
  }
  return 0;
}

// Probe (MI355X): do reads that hit the memory-side cache and writes that go to HBM share one budget, or can a kernel that
// does both at the same time finish in the time of the slower of the two?  Buffers of one c2 grid (33.5 MB).
//   fill   write-only (float4 stores)                     read   read-only (float4 loads, sum kept)
//   copy   every thread: load, store, load, store ...      phased every thread: all its loads first, then all its stores
//   mixed  half of the workgroups only read, the other half only write (both streams at the same time)
// Each timed launch follows a fill of the source, so the source sits in the memory-side cache like T does behind k_splat_xl.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_fill(float4* dst, size_t n4, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = float4{v, v, v, v};
}
// the same stream written through (sc1): one dword per lane per store, like the forward slab kernel's row stores
__global__ __launch_bounds__(256) void k_fill_through(float* dst, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __hip_atomic_store(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void k_fill_dword(float* dst, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
// a kernel that computes for `ticks` (100 MHz) and THEN stores its 33.5 MB: how much of the store stream hides under the
// next workgroups' compute when there are two rounds of workgroups, written back or written through?
__global__ __launch_bounds__(1024) void k_compute_then_store(float* dst, size_t per_block, int ticks, int through) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
  float* d = dst + (size_t)blockIdx.x * per_block;
  for (size_t i = threadIdx.x; i < per_block; i += blockDim.x) {
    if (through) __hip_atomic_store(d + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else d[i] = 1.0f;
  }
}
__global__ __launch_bounds__(256) void k_read(const float4* src, size_t n4, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const float4* src, float4* dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = src[i];
    v.x += 1.f;
    dst[i] = v;
  }
}
// 16 float4 per thread: all loads, then all stores (the column kernel's shape: 64 loads, compute, 64 stores)
__global__ __launch_bounds__(256) void k_phased(const float4* src, float4* dst, size_t n4) {
  const size_t base = ((size_t)blockIdx.x * blockDim.x) * 16 + threadIdx.x;
  float4 v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = base + (size_t)k * 256 < n4 ? src[base + (size_t)k * 256] : float4{0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k].x += 1.f;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (base + (size_t)k * 256 < n4) dst[base + (size_t)k * 256] = v[k];
}

// half of the workgroups read all of src, the other half write all of dst: both streams run at the same time
__global__ __launch_bounds__(256) void k_mixed(const float4* src, float4* dst, size_t n4, float* sink) {
  const size_t half = gridDim.x / 2, blk = blockIdx.x / 2;
  if (blockIdx.x & 1) {
    for (size_t i = blk * blockDim.x + threadIdx.x; i < n4; i += half * blockDim.x) dst[i] = float4{3.f, 3.f, 3.f, 3.f};
  } else {
    float acc = 0.f;
    for (size_t i = blk * blockDim.x + threadIdx.x; i < n4; i += half * blockDim.x) {
      const float4 v = src[i];
      acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
  }
}

int main() {
  const size_t bytes = 32ull * 64 * 64 * 64 * 4, n4 = bytes / 16;
  float4 *a, *b;
  float* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_compute_then_store), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  auto timed = [&](const char* name, auto launch) {
    std::vector<float> us;
    for (int i = 0; i < 30; ++i) {
      hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, a, n4, 1.0f);   // the source is fresh in the memory-side cache
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      us.push_back(ms * 1e3f);
    }
    std::sort(us.begin(), us.end());
    printf("%-44s median %.2f us  min %.2f us\n", name, us[us.size() / 2], us[0]);
  };
  for (int rep = 0; rep < 2; ++rep) {
    timed("empty event pair", [&] {});
    timed("write-only 33.5 MB", [&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, b, n4, 2.0f); });
    timed("write-only 33.5 MB, dword stores", [&] { hipLaunchKernelGGL(k_fill_dword, dim3(2048), dim3(256), 0, 0, (float*)b, n4 * 4, 2.0f); });
    timed("write-only 33.5 MB, dword stores written through (sc1)", [&] { hipLaunchKernelGGL(k_fill_through, dim3(2048), dim3(256), 0, 0, (float*)b, n4 * 4, 2.0f); });
    timed("512 workgroups: 4 us of compute, then 64 KB of stores each", [&] { hipLaunchKernelGGL(k_compute_then_store, dim3(512), dim3(1024), 140 * 1024, 0, (float*)b, n4 * 4 / 512, 400, 0); });
    timed("the same, stores written through (sc1)", [&] { hipLaunchKernelGGL(k_compute_then_store, dim3(512), dim3(1024), 140 * 1024, 0, (float*)b, n4 * 4 / 512, 400, 1); });
    timed("read-only 33.5 MB (just written)", [&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, a, n4, sink); });
    timed("copy 33.5 -> 33.5 MB, load/store interleaved", [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, a, b, n4); });
    timed("copy, every thread loads 16 then stores 16", [&] { hipLaunchKernelGGL(k_phased, dim3((unsigned)((n4 + 4095) / 4096)), dim3(256), 0, 0, a, b, n4); });
    timed("half the workgroups read 33.5 MB, half write 33.5 MB", [&] { hipLaunchKernelGGL(k_mixed, dim3(4096), dim3(256), 0, 0, a, b, n4, sink); });
    timed("the same with 2048 workgroups", [&] { hipLaunchKernelGGL(k_mixed, dim3(2048), dim3(256), 0, 0, a, b, n4, sink); });
    timed("copy in place (dst = src)", [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, a, a, n4); });
  }
  return 0;
}

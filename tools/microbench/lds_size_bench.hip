#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k_touch(float* out, int n4) {  // zero-fill n4 float4 of LDS + barrier + read back
  extern __shared__ float lds[];
  float4* p = reinterpret_cast<float4*>(lds);
  for (int i = threadIdx.x; i < n4; i += blockDim.x) p[i] = make_float4(0, 0, 0, 0);
  __syncthreads();
  float acc = 0;
  for (int i = threadIdx.x; i < n4; i += blockDim.x) acc += p[i].x;
  if (acc == 12345.f) out[0] = acc;
}
float timeit(int grid, int thr, size_t lds, int n4, float* o, int reps = 200) {
  hipFuncSetAttribute((const void*)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) k_touch<<<grid, thr, lds>>>(o, n4);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) k_touch<<<grid, thr, lds>>>(o, n4);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / reps;
}
int main() {
  float* o; hipMalloc(&o, 4);
  printf("256 WGs x 1024 threads, touching the first 64 KiB of the allocation:\n");
  for (size_t kb : {64, 80, 96, 112, 128, 132, 136, 144, 152, 156, 160})
    printf("  lds alloc %3zu KiB: %6.2f us/launch\n", kb, timeit(256, 1024, kb * 1024, 4096, o));
  printf("256 WGs x 1024 threads, touching the WHOLE allocation:\n");
  for (size_t kb : {64, 96, 128, 144, 156, 160})
    printf("  lds alloc %3zu KiB: %6.2f us/launch\n", kb, timeit(256, 1024, kb * 1024, (int)(kb * 64), o));
  printf("512 WGs x 512 threads (2/CU), whole allocation:\n");
  for (size_t kb : {32, 64, 72, 80})
    printf("  lds alloc %3zu KiB: %6.2f us/launch\n", kb, timeit(512, 512, kb * 1024, (int)(kb * 64), o));
  return 0;
}

// Known-byte-count kernels to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the two access widths the
// projection kernels use (dword per lane in the column kernels, 16 B per lane in the slab kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_copy_dword(const float* __restrict__ a, float* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void calib_copy_dwordx4(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
  const size_t n = (size_t)1 << 28;  // 1 GiB in, 1 GiB out: well past the 256 MiB Infinity Cache
  float *a, *b;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
  hipMemset(a, 1, n * 4);
  for (int r = 0; r < 3; ++r) {
    calib_copy_dword<<<2048, 256>>>(a, b, n);
    calib_copy_dwordx4<<<2048, 256>>>((const float4*)a, (float4*)b, n / 4);
  }
  hipDeviceSynchronize();
  printf("calibration: each kernel reads %zu bytes and writes %zu bytes\n", n * 4, n * 4);
  return 0;
}

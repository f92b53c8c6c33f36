#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k_empty(float* out) { extern __shared__ float lds[]; if (threadIdx.x == 2000) out[0] = lds[0]; }
__global__ __launch_bounds__(1024) void k_touch(float* out) {  // zero-fill all LDS with b128 stores + one barrier
  extern __shared__ float lds[];
  float4* p = reinterpret_cast<float4*>(lds);
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) p[i] = make_float4(0, 0, 0, 0);
  __syncthreads();
  if (threadIdx.x == 2000) out[0] = lds[5];
}
template <class K> float timeit(K kern, dim3 g, dim3 b, size_t lds, float* o, int reps = 200) {
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) kern<<<g, b, lds>>>(o);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) kern<<<g, b, lds>>>(o);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / reps;
}
int main() {
  float* o; hipMalloc(&o, 4);
  struct { int g, b; size_t lds; } cfg[] = {{256, 1024, 131072}, {512, 1024, 131072}, {256, 1024, 157000}, {256, 1024, 0}, {512, 512, 65536}, {1024, 512, 65536}, {1024, 256, 32768}, {2048, 256, 0}, {4096, 256, 0}, {256, 256, 0}};
  for (auto& c : cfg)
    printf("grid %5d x %4d thr, lds %6zu B: empty %6.2f us/launch   zero-fill+barrier %6.2f us/launch\n", c.g, c.b, c.lds,
           timeit(k_empty, dim3(c.g), dim3(c.b), c.lds, o), c.lds >= 131072 ? timeit(k_touch, dim3(c.g), dim3(c.b), c.lds, o) : 0.f);
  return 0;
}

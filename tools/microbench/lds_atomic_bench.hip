// Microbenchmark: LDS atomic throughput on gfx950 (float add vs uint add vs plain RMW), lane-linear vs random.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
template <int MODE>  // 0 float atomic, 1 uint atomic, 2 plain store, 3 float atomic returning
__global__ __launch_bounds__(1024) void k(const int* __restrict__ idx, int iters, float* out) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int a = idx[(size_t)it * blockDim.x + threadIdx.x];
    const float w = 1.0f + it * 0.001f;
    if (MODE == 0) atomicAdd(&lds[a], w);
    else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&lds[a]), (unsigned)it + 1u);
    else if (MODE == 2) lds[a] = w;
    else if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long*>(lds) + (a >> 1), (unsigned long long)(w * 1099511627776.0f));
    else acc += atomicAdd(&lds[a], w);
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = lds[threadIdx.x] + acc;
}
template <int MODE>
float run(const int* didx, int iters, float* dout) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  k<MODE><<<256, 1024, 131072>>>(didx, iters, dout);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) k<MODE><<<256, 1024, 131072>>>(didx, iters, dout);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5 * 1000.f;
}
int main() {
  const int iters = 64, nt = 1024;
  std::vector<int> lin(iters * nt), rnd(iters * nt), rnd_nc(iters * nt), same(iters * nt), clus(iters*nt);
  srand(1);
  for (int it = 0; it < iters; ++it)
    for (int t = 0; t < nt; ++t) {
      lin[it * nt + t] = (t + it * 1024) % 32768;                       // lane-linear: conflict-free
      rnd[it * nt + t] = rand() % 32768;                                // uniformly random addresses
      rnd_nc[it * nt + t] = ((rand() % 1024) * 32 + (t % 32)) % 32768;  // random rows, lane-own bank: no bank conflict
      same[it * nt + t] = (t / 64) * 7;                                 // whole wave on ONE address
      clus[it * nt + t] = (rand() % 2048) + 8192;                       // random within 2048 addresses (collisions)
    }
  int* d; float* o; hipMalloc(&d, iters * nt * 4); hipMalloc(&o, 256 * 64 * 4);
  const char* names[] = {"lane-linear", "random", "random-no-bank-conflict", "wave-same-address", "clustered-2048"};
  std::vector<int>* sets[] = {&lin, &rnd, &rnd_nc, &same, &clus};
  for (int s = 0; s < 5; ++s) {
    hipMemcpy(d, sets[s]->data(), iters * nt * 4, hipMemcpyHostToDevice);
    float f = run<0>(d, iters, o), u = run<1>(d, iters, o), p = run<2>(d, iters, o), fr = run<4>(d, iters, o);
    // per CU: iters * 16 wave-instructions of 64 lanes
    printf("%-26s float %7.1f us  uint %7.1f us  store %7.1f us  u64 %7.1f us   (cycles/wave-instr @2.4GHz: float %.1f uint %.1f store %.1f)\n",
           names[s], f, u, p, fr, f * 2400 / (iters * 16), u * 2400 / (iters * 16), p * 2400 / (iters * 16));
  }
  return 0;
}

// Probe (MI355X, ROCm 7.2): can two DEPENDENT-LOOKING kernels of one HIP stream overlap when the second is launched with
// hipExtAnyOrderLaunch (AQL packet without the barrier bit), and is dispatch in packet order, so that a consumer kernel
// that spins on tickets published by a producer kernel queued BEFORE it in the same stream can never starve it?
//   test 1  two spin kernels (128 workgroups x 20 us) back to back: plain launch vs any-order launch of the second
//   test 2  producer (1024 workgroups, 1024 threads, 140 KB LDS = 4 rounds of one per CU, 5 us each, one ticket add per
//           workgroup) followed by an any-order consumer (512 x 256 threads) whose workgroups poll the ticket counter
//           (bounded: 20 ms) and record when they saw it complete
//   test 3  the same pair captured into a hipGraph and replayed
// Every spin is bounded by the 100 MHz realtime counter, so nothing here can hang the GPU.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz

__global__ void k_spin(int ticks, unsigned long long* stamps) {
  const unsigned long long t0 = now();
  while (now() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && stamps) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = now();
  }
}

__global__ __launch_bounds__(1024) void k_producer(int ticks, unsigned int* counter, unsigned long long* stamps) {
  extern __shared__ float lds[];
  const unsigned long long t0 = now();
  lds[threadIdx.x] = (float)t0;
  while (now() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = now();
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// out[3*b] = start, out[3*b+1] = tick at which the counter read `target` (0 = gave up), out[3*b+2] = polls
__global__ __launch_bounds__(256) void k_consumer(unsigned int target, const unsigned int* counter, unsigned long long* out,
                                                   int timeout_ticks) {
  __shared__ int ok;
  const unsigned long long t0 = now();
  if (threadIdx.x == 0) {
    unsigned long long polls = 0, seen = 0;
    while (now() - t0 < (unsigned long long)timeout_ticks) {
      ++polls;
      if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
        seen = now();
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
    out[3 * blockIdx.x] = t0;
    out[3 * blockIdx.x + 1] = seen;
    out[3 * blockIdx.x + 2] = polls;
    ok = seen != 0;
  }
  __syncthreads();
  if (!ok) return;
}

static double ms_between(hipEvent_t a, hipEvent_t b) {
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, 1 << 20));
  unsigned int* counter;
  CK(hipMalloc(&counter, 256));

  // ---- test 1 ----
  for (int mode = 0; mode < 2; ++mode) {
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, st, 2000, (unsigned long long*)nullptr);
      if (mode == 0) hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, st, 2000, (unsigned long long*)nullptr);
      else hipExtLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, 2000, (unsigned long long*)nullptr);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      best = std::min(best, ms_between(e0, e1));
    }
    printf("test1 %-10s two 20 us spin kernels: %.1f us\n", mode ? "any-order" : "plain", 1e3 * best);
  }

  // ---- test 2 ----
  const int NP = 1024, NC = 512;
  const size_t lds = 140 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_producer), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  unsigned long long* cons;
  CK(hipMalloc(&cons, NC * 3 * 8));
  std::vector<unsigned long long> hp(2 * NP), hc(3 * NC);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(counter, 0, 4, st));
      CK(hipMemsetAsync(cons, 0, NC * 3 * 8, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(k_producer, dim3(NP), dim3(1024), lds, st, 500, counter, stamps);
      if (mode == 0) hipLaunchKernelGGL(k_consumer, dim3(NC), dim3(256), 0, st, (unsigned)NP, counter, cons, 2000000);
      else hipExtLaunchKernelGGL(k_consumer, dim3(NC), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, (unsigned)NP, counter, cons, 2000000);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(hp.data(), stamps, 2 * NP * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hc.data(), cons, 3 * NC * 8, hipMemcpyDeviceToHost));
      unsigned long long p0 = ~0ull, p1 = 0, c0 = ~0ull, c1 = 0, seen_min = ~0ull, polls = 0;
      int gave_up = 0;
      for (int i = 0; i < NP; ++i) { p0 = std::min(p0, hp[2 * i]); p1 = std::max(p1, hp[2 * i + 1]); }
      for (int i = 0; i < NC; ++i) {
        c0 = std::min(c0, hc[3 * i]);
        if (hc[3 * i + 1] == 0) ++gave_up; else { c1 = std::max(c1, hc[3 * i + 1]); seen_min = std::min(seen_min, hc[3 * i + 1]); }
        polls += hc[3 * i + 2];
      }
      printf("test2 %-10s total %.1f us | producer %.1f us | first consumer started %.1f us after the first producer, "
             "producers done at %.1f, consumers saw it at %.1f..%.1f, gave up %d, polls/WG %.1f\n",
             mode ? "any-order" : "plain", 1e3 * ms_between(e0, e1), (p1 - p0) / 100.0, ((double)c0 - (double)p0) / 100.0,
             (p1 - p0) / 100.0, ((double)seen_min - (double)p0) / 100.0, ((double)c1 - (double)p0) / 100.0, gave_up, (double)polls / NC);
    }
  }

  // ---- test 3: the pair inside a captured graph ----
  {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, st, 2000, (unsigned long long*)nullptr);
    hipExtLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, 2000, (unsigned long long*)nullptr);
    hipError_t e = hipStreamEndCapture(st, &g);
    if (e != hipSuccess) {
      printf("test3 capture of an any-order launch failed: %s\n", hipGetErrorString(e));
    } else {
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      double best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        best = std::min(best, ms_between(e0, e1));
      }
      printf("test3 graph replay of {plain, any-order} spin pair: %.1f us (40 = serialised, 20 = overlapped)\n", 1e3 * best);
    }
  }

  // ---- test 4: host cost of a launch (eager, queue kept busy) ----
  {
    const int n = 2000;
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i)
      hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, (i & 3) ? hipExtAnyOrderLaunch : 0, 0, (unsigned long long*)nullptr);
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(st));
    auto t2 = std::chrono::steady_clock::now();
    printf("test4 host cost per hipExtLaunchKernel: %.2f us (queue drained after %.2f us per launch)\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
  }
  return 0;
}

// Probe (MI355X, ROCm 7.2): what a dependency between two HIP streams costs on the GPU time line, next to a dependent
// kernel boundary inside one stream.  Spin kernels of a fixed length (10 us, 256 workgroups); eager launches.
//   same    : N kernels back to back in one stream                         -> per kernel = 10 us + boundary
//   pingpong: kernel on A, event, B waits, kernel on B, event, A waits ... -> per kernel = 10 us + cross-stream hop
//   forkjoin: per step: A records, B waits; 2 kernels on A and 2 on B concurrently (128 workgroups each); A waits for B
//             -> per step = 20 us + fork/join cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__global__ void k_spin(int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
}

int main() {
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  const int N = 200;
  hipEvent_t ev[2 * N + 2];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  auto wall = [&](auto fn) {
    CK(hipDeviceSynchronize());
    auto a = std::chrono::steady_clock::now();
    fn();
    CK(hipDeviceSynchronize());
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
  };
  for (int rep = 0; rep < 3; ++rep) {
    double same = wall([&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, 1000); });
    double pp = wall([&] {
      for (int i = 0; i < N; ++i) {
        hipStream_t s = (i & 1) ? B : A, o = (i & 1) ? A : B;
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, 1000);
        CK(hipEventRecord(ev[i], s));
        CK(hipStreamWaitEvent(o, ev[i], 0));
      }
    });
    double fj = wall([&] {
      for (int i = 0; i < N / 2; ++i) {
        CK(hipEventRecord(ev[2 * i], A));
        CK(hipStreamWaitEvent(B, ev[2 * i], 0));
        hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, A, 1000);
        hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, B, 1000);
        hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, A, 1000);
        hipLaunchKernelGGL(k_spin, dim3(128), dim3(256), 0, B, 1000);
        CK(hipEventRecord(ev[2 * i + 1], B));
        CK(hipStreamWaitEvent(A, ev[2 * i + 1], 0));
      }
    });
    printf("same stream: %.2f us per 10 us kernel | ping-pong over two streams: %.2f us per kernel | fork/join step of 2x2 kernels: %.2f us per step (20 = free)\n",
           same / N, pp / N, fj / (N / 2));
  }
  return 0;
}

// How much do 4- / 8- / 16-byte per-lane global accesses cost on MI355X for THIS path's shapes?  Grid-sized streams
// (33.5 MB = 32 clouds x 64^3 fp32) read or written plane by plane with lanes on consecutive addresses, 64 accesses per
// lane kept in flight (the ray-march kernel's shape).   hipcc -O3 --offload-arch=gfx950 access_width.hip -o access_width
#include <hip/hip_runtime.h>
#include <cstdio>

template <class V, bool WRITE>
__global__ __launch_bounds__(256) void k_stream(const V* __restrict__ in, V* __restrict__ out, size_t plane_elems) {
  // block handles 256 lanes x 64 planes; plane stride = plane_elems (in units of V)
  const size_t base = (size_t)blockIdx.y * 64 * plane_elems + (size_t)blockIdx.x * 256 + threadIdx.x;
  V v[64];
  if (!WRITE) {
#pragma unroll
    for (int z = 0; z < 64; ++z) v[z] = in[base + z * plane_elems];
    float acc = 0.f;
#pragma unroll
    for (int z = 0; z < 64; ++z) acc += reinterpret_cast<const float*>(&v[z])[0];
    if (acc == 123.456f) reinterpret_cast<float*>(out)[0] = acc;
  } else {
    V x;
    for (unsigned i = 0; i < sizeof(V) / 4; ++i) reinterpret_cast<float*>(&x)[i] = (float)threadIdx.x;
#pragma unroll
    for (int z = 0; z < 64; ++z) out[base + z * plane_elems] = x;
  }
}

template <class V, bool WRITE>
float run(const float* in, float* out, int clouds) {
  const size_t plane_elems = 4096 / (sizeof(V) / 4);   // one 64 x 64 plane
  dim3 grid(plane_elems / 256, clouds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k_stream<V, WRITE><<<grid, 256>>>((const V*)in, (V*)out, plane_elems);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) k_stream<V, WRITE><<<grid, 256>>>((const V*)in, (V*)out, plane_elems);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 20 * 1e3f;
}

int main() {
  const int clouds = 32;
  const size_t bytes = (size_t)clouds * 64 * 4096 * 4;
  float *in, *out;
  hipMalloc(&in, bytes); hipMalloc(&out, bytes);
  hipMemset(in, 0, bytes);
  printf("stream of %.1f MB, 64 accesses per lane in flight\n", bytes / 1e6);
  float t;
  t = run<float, false>(in, out, clouds);  printf("read   4 B/lane: %6.2f us  %5.2f TB/s\n", t, bytes / t / 1e6);
  t = run<float2, false>(in, out, clouds); printf("read   8 B/lane: %6.2f us  %5.2f TB/s\n", t, bytes / t / 1e6);
  t = run<float4, false>(in, out, clouds); printf("read  16 B/lane: %6.2f us  %5.2f TB/s\n", t, bytes / t / 1e6);
  t = run<float, true>(in, out, clouds);   printf("write  4 B/lane: %6.2f us  %5.2f TB/s\n", t, bytes / t / 1e6);
  t = run<float2, true>(in, out, clouds);  printf("write  8 B/lane: %6.2f us  %5.2f TB/s\n", t, bytes / t / 1e6);
  t = run<float4, true>(in, out, clouds);  printf("write 16 B/lane: %6.2f us  %5.2f TB/s\n", t, bytes / t / 1e6);
  return 0;
}

// Offline ISA probe (no GPU needed: hipcc -S) of the store shape that gave wrong values in round 3: the forward slab kernel's
// "four rows -> H pass in registers -> W pass across the lanes (DPP wave shifts) -> 4x4 quad transpose (DPP quad permutes) ->
// ONE 16-byte store per lane", once as an ordinary store (VARIANT 0), once as buffer_store_dwordx4 ... sc1 through the raw
// buffer builtin (VARIANT 1), once as four dword buffer stores sc1 (VARIANT 2), at tap radius RB (the failure showed at RB = 1 only).
//   hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only -DVARIANT=1 -DRB=1 sc1_b128_store_probe.hip -o v1.s
// With -DRUN the file is a host program too: it runs the three variants on one input and compares them lane by lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef RB
#define RB 1
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct Taps { float w[2 * RB + 1]; };

template <int CTRL>
__device__ inline float dpp_zero_fill(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ inline float from_lane_below(float v) { return dpp_zero_fill<0x138>(v); }  // wave_shr:1
__device__ inline float from_lane_above(float v) { return dpp_zero_fill<0x130>(v); }  // wave_shl:1
__device__ inline void quad_transpose(float (&r)[4], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll
  for (int j = 0; j < 4; j += 2) {
    const float got = dpp_zero_fill<0xb1>(b0 ? r[j] : r[j + 1]);   // quad_perm:[1,0,3,2]
    r[j + 1] = b0 ? r[j + 1] : got;
    r[j] = b0 ? got : r[j];
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float got = dpp_zero_fill<0x4e>(b1 ? r[j] : r[j + 2]);   // quad_perm:[2,3,0,1]
    r[j + 2] = b1 ? r[j + 2] : got;
    r[j] = b1 ? got : r[j];
  }
}
__device__ inline float wpass_lanes(float v, const Taps& taps) {
  float acc = taps.w[RB] * v;
  float lo = v, hi = v;
#pragma unroll
  for (int k = 1; k <= RB; ++k) {
    lo = from_lane_below(lo);
    hi = from_lane_above(hi);
    acc = fmaf(taps.w[RB - k], lo, acc);
    acc = fmaf(taps.w[RB + k], hi, acc);
  }
  return acc;
}

constexpr int SEG = 16, G = 64;
// one wave = one x row of 64 lanes; thread owns SEG rows at its x; in: [planes][G + 2 RB][G] (rows padded), out: [planes][G][G]
template <int V>
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ in, float* __restrict__ out, Taps taps, int planes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;      // 4 waves = 4 row segments of one plane
  const int zz = blockIdx.x, y0 = wave * SEG;
  if (zz >= planes) return;
  float v[SEG + 2 * RB];
#pragma unroll
  for (int i = 0; i < SEG + 2 * RB; ++i) v[i] = in[((size_t)zz * (G + 2 * RB) + y0 + i) * G + lane];
  float* Tout = out + (((size_t)zz * G) + y0 + (lane & 3)) * G + (lane & ~3);
  const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(out, 0, planes * G * G * 4, 0x00020000);
  const int voff = (int)((((size_t)zz * G) + y0 + (lane & 3)) * G + (lane & ~3)) * 4;
#pragma unroll
  for (int j = 0; j < SEG; j += 4) {
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float h = 0.f;
#pragma unroll
      for (int tp = 0; tp < 2 * RB + 1; ++tp) h = fmaf(taps.w[tp], v[j + e + tp], h);
      o[e] = wpass_lanes(h, taps);
    }
    quad_transpose(o, lane);
    if (V == 0) {
      *reinterpret_cast<f32x4*>(Tout + (size_t)j * G) = f32x4{o[0], o[1], o[2], o[3]};
    } else if (V == 1) {
      const u32x4 d4 = u32x4{__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
      __builtin_amdgcn_raw_buffer_store_b128(d4, dst, voff + j * G * 4, 0, 16);   // aux 16 = sc1
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[e]), dst, voff + j * G * 4 + 4 * e, 0, 16);
    }
  }
}
template __global__ void k_probe<VARIANT>(const float*, float*, Taps, int);

#ifdef RUN
int main() {
  const int planes = 64;
  const size_t nin = (size_t)planes * (G + 2 * RB) * G, nout = (size_t)planes * G * G;
  float *hin = new float[nin], *h[3];
  for (size_t i = 0; i < nin; ++i) hin[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
  float *din, *dout;
  hipMalloc(&din, nin * 4); hipMalloc(&dout, nout * 4);
  hipMemcpy(din, hin, nin * 4, hipMemcpyHostToDevice);
  Taps t;
  for (int i = 0; i < 2 * RB + 1; ++i) t.w[i] = 1.0f / (1 + (i - RB) * (i - RB));
  for (int v = 0; v < 3; ++v) {
    h[v] = new float[nout];
    hipMemset(dout, 0, nout * 4);
    if (v == 0) hipLaunchKernelGGL(k_probe<0>, dim3(planes), dim3(256), 0, 0, din, dout, t, planes);
    if (v == 1) hipLaunchKernelGGL(k_probe<1>, dim3(planes), dim3(256), 0, 0, din, dout, t, planes);
    if (v == 2) hipLaunchKernelGGL(k_probe<2>, dim3(planes), dim3(256), 0, 0, din, dout, t, planes);
    hipDeviceSynchronize();
    hipMemcpy(h[v], dout, nout * 4, hipMemcpyDeviceToHost);
  }
  for (int v = 1; v < 3; ++v) {
    size_t bad = 0; int lanes[64] = {0};
    for (size_t i = 0; i < nout; ++i) if (h[v][i] != h[0][i]) { ++bad; ++lanes[i % 64]; }
    printf("RB=%d variant %d vs ordinary store: %zu of %zu values differ; by x:", RB, v, bad, nout);
    for (int x = 0; x < 64; ++x) if (lanes[x]) printf(" %d:%d", x, lanes[x]);
    printf("\n");
  }
  return 0;
}
#endif

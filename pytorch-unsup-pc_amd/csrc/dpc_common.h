// Device-side building blocks shared by the fused and the stage-level kernels (gfx950 / CDNA4 only).
//
// Layout reminders: grid voxel (z,y,x) of cloud b at ((b*D+z)*H+y)*W+x; point clouds [B,N,3]; wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dpc_render.h"

#define DPC_WAVE 64

// Tap weights of one 1-D kernel, centred in a compile-time radius bucket RB (zero padded), passed BY VALUE
// as a kernel argument so that the statically indexed weights live in SGPRs.
template <int RB>
struct TapsT {
  float w[2 * RB + 1];
};

// Runtime-length taps for the generic (slow-path) kernels.
struct TapsDyn {
  float w[DPC_MAX_TAPS];
  int n;  // number of taps (odd) or 0
};

// Tap VALUES from device memory (DpcParams.dev_taps_*: a captured HIP graph under a sigma schedule, include/dpc_render.h).
// `host` came with the launch and fixed the radius bucket RB; when `dev` is given, every tap of the n-tap kernel that falls
// inside the compiled window replaces it (uniform addresses: scalar loads, the weights stay in SGPRs).  flip: the adjoint
// (correlation with the reversed kernel).
template <int RB>
__device__ inline TapsT<RB> resolve_taps(const TapsT<RB>& host, const float* __restrict__ dev, int n, bool flip) {
  if (dev == nullptr || n <= 0) return host;
  TapsT<RB> t;
  const int c = (n - 1) / 2;
#pragma unroll
  for (int i = 0; i < 2 * RB + 1; ++i) {
    const int o = i - RB, src = c + (flip ? -o : o);
    float w = 0.f;
    if (src >= 0 && src < n) w = dev[src];
    t.w[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w)));
  }
  return t;
}

__device__ inline TapsDyn resolve_taps_dyn(const TapsDyn& host, const float* __restrict__ dev, bool flip) {
  if (dev == nullptr || host.n <= 0) return host;
  TapsDyn t;
  t.n = host.n;
  for (int i = 0; i < DPC_MAX_TAPS; ++i) t.w[i] = i < host.n ? dev[flip ? host.n - 1 - i : i] : 0.f;
  return t;
}

// ------------------------------------------------------------------------------------------------------
// Camera: p -> (z, y, x) = (p'_0, f p'_1 / (p'_0 + d), f p'_2 / (p'_0 + d)),  p' = R(q/|q|) p + t
// (dpc/util/point_cloud_to.py:135-148,169-177; dpc/util/quaternion.py:110-132)
// ------------------------------------------------------------------------------------------------------
struct Camera {
  float r[9];        // row-major rotation of the normalised quaternion
  float qw, qx, qy, qz;  // normalised quaternion
  float inv_norm;    // 1/|q|
  float tx, ty, tz;  // translation in the reference's (index 0,1,2) order, 0 when absent
  float f, d;
};

// The camera inputs as they sit in memory: loaded early (a dependent global read), turned into a Camera where needed.
struct CameraRaw {
  float q[4], t[3], f;
};

__device__ inline CameraRaw load_camera_raw(const DpcParams& P, const float* __restrict__ q, const float* __restrict__ t,
                                            const float* __restrict__ f, int b) {
  CameraRaw r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.q[i] = q[4 * b + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) r.t[i] = t ? t[3 * b + i] : 0.f;
  r.f = f ? f[b] : P.focal_length;
  return r;
}

// Backward-side camera (the forward uses the reference-exact CameraRef below).
__device__ inline Camera make_camera(const DpcParams& P, const CameraRaw& raw) {
  Camera c;
  // normalise in double once per thread: keeps R orthogonal to ~1e-16 before the single rounding to fp32
  // (an all-fp32 variant measured no faster and cost a third of the d(q) parity margin)
  double w = raw.q[0], x = raw.q[1], y = raw.q[2], z = raw.q[3];
  double n = sqrt(w * w + x * x + y * y + z * z);
  double in = 1.0 / n;
  w *= in; x *= in; y *= in; z *= in;
  c.qw = (float)w; c.qx = (float)x; c.qy = (float)y; c.qz = (float)z;
  c.inv_norm = (float)in;
  c.r[0] = (float)(1.0 - 2.0 * (y * y + z * z));
  c.r[1] = (float)(2.0 * (x * y - w * z));
  c.r[2] = (float)(2.0 * (x * z + w * y));
  c.r[3] = (float)(2.0 * (x * y + w * z));
  c.r[4] = (float)(1.0 - 2.0 * (x * x + z * z));
  c.r[5] = (float)(2.0 * (y * z - w * x));
  c.r[6] = (float)(2.0 * (x * z - w * y));
  c.r[7] = (float)(2.0 * (y * z + w * x));
  c.r[8] = (float)(1.0 - 2.0 * (x * x + y * y));
  c.tx = raw.t[0]; c.ty = raw.t[1]; c.tz = raw.t[2];
  c.f = raw.f;
  c.d = P.camera_distance;
  return c;
}

__device__ inline Camera load_camera(const DpcParams& P, const float* __restrict__ q, const float* __restrict__ t,
                                     const float* __restrict__ f, int b) {
  return make_camera(P, load_camera_raw(P, q, t, f, b));
}

struct Projected {
  float Z, Y, X;     // output coordinates (z,y,x order of the reference)
  float p0, p1, p2;  // rotated + translated point p''
  float inv;         // 1 / (p''_0 + d)
};

__device__ inline Projected project_point(const Camera& c, float px, float py, float pz) {
  Projected o;
  const float rot0 = fmaf(c.r[0], px, fmaf(c.r[1], py, c.r[2] * pz));
  o.p0 = rot0 + c.tx;
  o.p1 = fmaf(c.r[3], px, fmaf(c.r[4], py, c.r[5] * pz)) + c.ty;
  o.p2 = fmaf(c.r[6], px, fmaf(c.r[7], py, c.r[8] * pz)) + c.tz;
  o.inv = 1.0f / (o.p0 + c.d);
  o.X = c.f * o.p2 * o.inv;
  o.Y = c.f * o.p1 * o.inv;
  o.Z = rot0;  // the reference adds t_0 and d, then subtracts both again (:145,172-175)
  return o;
}

// Accumulators of the transform backward for one cloud: M = sum p G^T (9), dt (3), df (1).
struct CamGrad {
  float m[9];
  float dt[3];
  float df;
};

__device__ inline void camgrad_zero(CamGrad& g) {
#pragma unroll
  for (int i = 0; i < 9; ++i) g.m[i] = 0.f;
  g.dt[0] = g.dt[1] = g.dt[2] = 0.f;
  g.df = 0.f;
}

// Backward of project_point: (dZ,dY,dX) -> dp (returned through dpx..), accumulates M, dt, df.
__device__ inline void project_point_bwd(const Camera& c, const Projected& o, float px, float py, float pz, float dZ,
                                         float dY, float dX, float& dpx, float& dpy, float& dpz, CamGrad& g) {
  const float g2 = c.f * o.inv * dX;                      // d/dp''_2
  const float g1 = c.f * o.inv * dY;                      // d/dp''_1
  const float persp = -(o.X * dX + o.Y * dY) * o.inv;     // through 1/(p''_0 + d)
  const float g0 = dZ + persp;                            // d/dp''_0 (w.r.t. the rotated point)
  g.df += (o.p2 * dX + o.p1 * dY) * o.inv;
  g.dt[0] += persp;                                       // z output does not depend on t_0 (+t_0 - t_0)
  g.dt[1] += g1;
  g.dt[2] += g2;
  dpx = fmaf(c.r[0], g0, fmaf(c.r[3], g1, c.r[6] * g2));  // R^T G
  dpy = fmaf(c.r[1], g0, fmaf(c.r[4], g1, c.r[7] * g2));
  dpz = fmaf(c.r[2], g0, fmaf(c.r[5], g1, c.r[8] * g2));
  g.m[0] += px * g0; g.m[1] += px * g1; g.m[2] += px * g2;
  g.m[3] += py * g0; g.m[4] += py * g1; g.m[5] += py * g2;
  g.m[6] += pz * g0; g.m[7] += pz * g1; g.m[8] += pz * g2;
}

// From M = sum_points p G^T to the gradient w.r.t. the UNNORMALISED quaternion.  With qn = (w, v):
//   p' = (w^2 - |v|^2) p + 2 (v.p) v + 2 w (v x p)      (vector part of qn (0,p) qn*)
//   dL/dw = 2 w tr(M) + 2 v.c ,  dL/dv = -2 tr(M) v + 2 (M + M^T) v + 2 w c ,  c = sum p x G
// followed by the Jacobian of q -> q/|q| (the norm is not detached, dpc/util/quaternion.py:119-121).
__device__ inline void quaternion_grad(const Camera& c, const float* m, float* dq) {
  const float w = c.qw, vx = c.qx, vy = c.qy, vz = c.qz;
  const float tr = m[0] + m[4] + m[8];
  const float cx = m[5] - m[7], cy = m[6] - m[2], cz = m[1] - m[3];
  const float sx = 2.f * m[0] * vx + (m[1] + m[3]) * vy + (m[2] + m[6]) * vz;  // ((M + M^T) v)_x
  const float sy = (m[3] + m[1]) * vx + 2.f * m[4] * vy + (m[5] + m[7]) * vz;
  const float sz = (m[6] + m[2]) * vx + (m[7] + m[5]) * vy + 2.f * m[8] * vz;
  const float gw = 2.f * (w * tr + vx * cx + vy * cy + vz * cz);
  const float gx = 2.f * (-tr * vx + sx + w * cx);
  const float gy = 2.f * (-tr * vy + sy + w * cy);
  const float gz = 2.f * (-tr * vz + sz + w * cz);
  const float dot = w * gw + vx * gx + vy * gy + vz * gz;
  dq[0] = (gw - w * dot) * c.inv_norm;
  dq[1] = (gx - vx * dot) * c.inv_norm;
  dq[2] = (gy - vy * dot) * c.inv_norm;
  dq[3] = (gz - vz * dot) * c.inv_norm;
}

// ------------------------------------------------------------------------------------------------------
// Reference-exact camera, used once per point by the locate / transform kernels.
//
// The reference's hard thresholds (|c| <= 1/2, clamp at eps) sit on quantities such as g - floor(g) that are
// decided by the last bits of the transformed coordinate, so the transform is reproduced OPERATION BY
// OPERATION, in the precision each reference op runs in (verified bit-for-bit against the reference):
//   quaternion.py:119-121  |q| and q/|q| in fp32 (sequential sum of squares, sqrt, divide)
//   quaternion.py:79-85    first Hamilton product qn * (0,p): fp32, every product and sum rounded
//   quaternion.py:91-92    conjugate multiplies by a float64 constant => second product runs in fp64
//   point_cloud_to.py:137-177  translation, +d, *f, /z', -d, -t0 in fp64, in that order
// No FMA contraction anywhere in these functions (torch executes each op as its own rounded kernel).
// ------------------------------------------------------------------------------------------------------
struct CameraRef {
  float w, x, y, z;        // qn = q/|q| in fp32
  double cw, cx, cy, cz;   // conjugate, fp64
  double tx, ty, tz;
  double f, d;
  bool has_t;
};

__device__ inline CameraRef load_camera_ref(const DpcParams& P, const float* __restrict__ q,
                                            const float* __restrict__ t, const float* __restrict__ f, int b) {
#pragma clang fp contract(off)
  CameraRef c;
  const float q0 = q[4 * b + 0], q1 = q[4 * b + 1], q2 = q[4 * b + 2], q3 = q[4 * b + 3];
  float acc = 0.f;
  acc = acc + q0 * q0;
  acc = acc + q1 * q1;
  acc = acc + q2 * q2;
  acc = acc + q3 * q3;
  const float n = sqrtf(acc);
  c.w = q0 / n; c.x = q1 / n; c.y = q2 / n; c.z = q3 / n;
  c.cw = (double)c.w; c.cx = -(double)c.x; c.cy = -(double)c.y; c.cz = -(double)c.z;
  c.has_t = t != nullptr;
  c.tx = t ? (double)t[3 * b + 0] : 0.0;
  c.ty = t ? (double)t[3 * b + 1] : 0.0;
  c.tz = t ? (double)t[3 * b + 2] : 0.0;
  c.f = f ? (double)f[b] : (double)P.focal_length;
  c.d = (double)P.camera_distance;
  return c;
}

__device__ inline void project_point_ref(const CameraRef& c, float x2, float y2, float z2, double& Z, double& Y,
                                         double& X) {
#pragma clang fp contract(off)
  const float w2 = 0.0f;  // vector3d_to_quaternion pads a zero in front
  const float w1 = c.w, x1 = c.x, y1 = c.y, z1 = c.z;
  const float aw = ((w1 * w2 - x1 * x2) - y1 * y2) - z1 * z2;
  const float ax = ((w1 * x2 + x1 * w2) + y1 * z2) - z1 * y2;
  const float ay = ((w1 * y2 + y1 * w2) + z1 * x2) - x1 * z2;
  const float az = ((w1 * z2 + z1 * w2) + x1 * y2) - y1 * x2;
  const double dw = (double)aw, dx = (double)ax, dy = (double)ay, dz = (double)az;
  double p0 = ((dw * c.cx + dx * c.cw) + dy * c.cz) - dz * c.cy;
  double p1 = ((dw * c.cy + dy * c.cw) + dz * c.cx) - dx * c.cz;
  double p2 = ((dw * c.cz + dz * c.cw) + dx * c.cy) - dy * c.cx;
  if (c.has_t) {
    p0 = p0 + c.tx; p1 = p1 + c.ty; p2 = p2 + c.tz;
  }
  double zs = p0 + c.d;
  const double xs = (p2 * c.f) / zs;
  const double ys = (p1 * c.f) / zs;
  zs = zs - c.d;
  if (c.has_t) zs = zs - c.tx;
  Z = zs; Y = ys; X = xs;
}

// ------------------------------------------------------------------------------------------------------
// Trilinear cell of a transformed point (dpc/util/point_cloud_to.py:26-40) and its 16-byte record.
//   code  = iz<<20 | iy<<10 | ix, or -1 for a point outside [-1/2,1/2]^3
//   tz,ty,tx encode the fractional part r of each grid coordinate as t = r (r < 1/2) or r - 1 (otherwise),
//   so that BOTH interpolation weights r and 1-r are recovered with fp32 relative precision.
// ------------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) PointRec {
  int code;
  float tz, ty, tx;
};

struct Cell {
  int iz, iy, ix;
  float wz[2], wy[2], wx[2];
  bool valid;
};

__device__ inline float frac_encode(double g, int& cell) {
#pragma clang fp contract(off)
  const double fl = floor(g);
  cell = (int)fl;
  const double r = g - fl;
  return (float)(r < 0.5 ? r : r - 1.0);
}

__device__ inline PointRec make_record(double Z, double Y, double X, int D, int H, int W) {
#pragma clang fp contract(off)
  PointRec rec;
  const bool valid = (Z >= -0.5) && (Z <= 0.5) && (Y >= -0.5) && (Y <= 0.5) && (X >= -0.5) && (X <= 0.5);
  int iz, iy, ix;
  rec.tz = frac_encode((Z + 0.5) * (double)(D - 1), iz);
  rec.ty = frac_encode((Y + 0.5) * (double)(H - 1), iy);
  rec.tx = frac_encode((X + 0.5) * (double)(W - 1), ix);
  rec.code = valid ? ((iz << 20) | (iy << 10) | ix) : -1;
  if (!valid) rec.tz = rec.ty = rec.tx = 0.f;
  return rec;
}

__device__ inline void frac_decode(float t, float (&w)[2]) {
  w[1] = t >= 0.f ? t : 1.0f + t;
  w[0] = t >= 0.f ? 1.0f - t : -t;
}

__device__ inline Cell cell_from_record(const PointRec& rec) {
  Cell c;
  c.valid = rec.code >= 0;
  c.iz = rec.code >> 20;
  c.iy = (rec.code >> 10) & 1023;
  c.ix = rec.code & 1023;
  frac_decode(rec.tz, c.wz);
  frac_decode(rec.ty, c.wy);
  frac_decode(rec.tx, c.wx);
  return c;
}

__device__ inline PointRec load_record(const PointRec* __restrict__ recs, size_t i) {
  const int4 v = *reinterpret_cast<const int4*>(recs + i);  // one 16-byte load
  PointRec r;
  r.code = v.x;
  r.tz = __int_as_float(v.y);
  r.ty = __int_as_float(v.z);
  r.tx = __int_as_float(v.w);
  return r;
}

// ------------------------------------------------------------------------------------------------------
// Sliding-window 1-D correlation of one line segment held in registers.
//   v[i] = line[start + i - RB] (zero outside [0,len)),  out(j) = sum_t w[t] v[j+t],  j = 0..L-1
// All indices are compile-time after unrolling, so v[] stays in VGPRs and w[] in SGPRs.
// ------------------------------------------------------------------------------------------------------
template <int RB, int L, bool CLAMP1>
__device__ inline void window_load(float (&v)[L + 2 * RB], const float* base, int stride, int start, int len) {
#pragma unroll
  for (int i = 0; i < L + 2 * RB; ++i) {
    const int pos = start + i - RB;
    float x = (pos >= 0 && pos < len) ? base[pos * stride] : 0.f;
    if (CLAMP1) x = fminf(x, 1.0f);
    v[i] = x;
  }
}

// Order in which the ADJOINT passes add the taps of a 1-D kernel: from the edges inwards, the centre last
// (0, 2RB, 1, 2RB-1, ..., RB).  A Gaussian's terms grow towards the centre and gradient windows are signed: adding the small
// terms first keeps the partial sums, and with them every rounding of the chain, small (the d(points) error at 21 live taps sat
// at 0.80 of the parity rule with the taps added left to right).  Same FMAs, same count.
template <int RB>
__host__ __device__ constexpr int tap_edge_first(int i) { return i < 2 * RB ? ((i & 1) ? 2 * RB - (i >> 1) : (i >> 1)) : RB; }

template <int RB, int L>
__device__ inline float window_dot(const float (&v)[L + 2 * RB], const TapsT<RB>& taps, int j) {
  float acc = 0.f;
#pragma unroll
  for (int t = 0; t < 2 * RB + 1; ++t) acc = fmaf(taps.w[t], v[j + t], acc);
  return acc;
}

// Zero fill as a kernel.  The library never uses hipMemsetAsync: with memset nodes for the fixed-point accumulators the captured
// 300-kernel training step drifted 7 % from the eager one within 50 replays (round 3), and with this kernel it does not.  A graph
// of nothing but memset -> kernel -> copy re-zeroes correctly on every replay (round 4, tools/microbench/memset_node_probe.py,
// 4 KB .. 32 MB, through torch and through hipMemsetAsync on the capturing stream), so what went wrong inside the big graph
// was never isolated; the kernel costs nothing and stays.
static __global__ __launch_bounds__(256) void k_zero_words(unsigned int* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
inline bool zero_words_async(void* p, size_t words, hipStream_t st) {
  if (words == 0) return true;
  const unsigned blocks = (unsigned)((words + 255) / 256 < 2048 ? (words + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_zero_words, dim3(blocks), dim3(256), 0, st, static_cast<unsigned int*>(p), words);
  return hipGetLastError() == hipSuccess;
}

// ------------------------------------------------------------------------------------------------------
// Reductions
// ------------------------------------------------------------------------------------------------------
// Wave-wide sum with DPP-modified VALU adds (no LDS traffic, unlike __shfl_xor = ds_bpermute): butterfly inside
// quads, rotations inside the 16-lane rows, then row_bcast:15 / row_bcast:31 carry row totals upward; lane 63
// ends with the total, which v_readlane broadcasts.  gfx9-family DPP controls (row_bcast exists on gfx950).
template <int CTRL>
__device__ inline float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

__device__ inline float wave_sum(float v) {
  v += dpp_move<0xb1>(v);   // quad_perm:[1,0,3,2]
  v += dpp_move<0x4e>(v);   // quad_perm:[2,3,0,1]
  v += dpp_move<0x124>(v);  // row_ror:4
  v += dpp_move<0x128>(v);  // row_ror:8   -> every lane holds its row's sum
  v += dpp_move<0x142>(v);  // row_bcast:15 (lanes without a source add 0)
  v += dpp_move<0x143>(v);  // row_bcast:31
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Inclusive prefix sum over the 64 lanes of a wave, on DPP-modified adds (six VALU instructions; six __shfl_up are six
// ds_bpermute round trips through the LDS crossbar): Kogge-Stone inside the 16-lane rows (row_shr, lanes without a source add
// 0), then row_bcast:15 hands rows 1 and 3 their left neighbour's total and row_bcast:31 hands rows 2 and 3 the total of the
// lower half.
template <int CTRL, int ROW_MASK>
__device__ inline int dpp_move_int(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ inline int wave_inclusive_scan(int v) {
  v += dpp_move_int<0x111, 0xf>(v);  // row_shr:1
  v += dpp_move_int<0x112, 0xf>(v);  // row_shr:2
  v += dpp_move_int<0x114, 0xf>(v);  // row_shr:4
  v += dpp_move_int<0x118, 0xf>(v);  // row_shr:8   -> inclusive inside every row
  v += dpp_move_int<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_move_int<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

// Sum NV per-thread values over the block; result valid in thread 0.  red must hold NV * (blockDim/64) floats.
// Wave shuffles, then lanes 0..NV-1 of wave 0 each add one value's per-wave partials (independent LDS reads
// that pipeline), then thread 0 collects the NV results with shuffles -- no serial chain of LDS latencies.
template <int NV>
__device__ inline void block_sum(float (&vals)[NV], float* red) {
  static_assert(NV <= DPC_WAVE, "one lane per value");
  const int lane = threadIdx.x & (DPC_WAVE - 1), wave = threadIdx.x / DPC_WAVE, nw = (blockDim.x + DPC_WAVE - 1) / DPC_WAVE;
#pragma unroll
  for (int i = 0; i < NV; ++i) vals[i] = wave_sum(vals[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave * NV + i] = vals[i];
  }
  __syncthreads();
  if (wave == 0) {
    float s = 0.f;
    if (lane < NV)
      for (int k = 0; k < nw; ++k) s += red[k * NV + lane];
#pragma unroll
    for (int i = 0; i < NV; ++i) vals[i] = __shfl(s, i, DPC_WAVE);
  }
}

// Fixed-order fp64 block sum of the 13 camera-gradient accumulators and d(q) in fp64: see the notes on the camera
// gradient in dpc_kernels.h.  scratch: camgrad_scratch_bytes(nthr) of LDS nobody else is using.
__host__ __device__ constexpr size_t camgrad_scratch_bytes(int nthr) { return (size_t)13 * (nthr / DPC_WAVE) * sizeof(float); }

// nthr a multiple of 64.  Per wave a DPP butterfly (a fixed tree, fp32: the per-point values are fp32 and their rounding
// dominates the error, measured); the wave totals go through LDS and lanes 0..12 of the first wave add them in wave order
// in fp64.  The order of every addition is fixed by the thread ids alone.  Returns value `tid`'s total in threads 0..12.
// (A version that parked all 13 x nthr values in LDS and summed them in fp64 measured 0.7 us slower per workgroup and no
// more accurate.)
// `nvals` (block-uniform) = 9 when neither a translation nor a focal length takes a gradient: the last four accumulators
// (dt, df) are then not reduced at all (their totals read 0).
__device__ inline double block_sum13_fixed(const float (&vals)[13], float* scratch, int tid, int nthr, int nvals = 13) {
  const int lane = tid & (DPC_WAVE - 1), wave = tid / DPC_WAVE, nw = nthr / DPC_WAVE;
  float ws[13];
#pragma unroll
  for (int i = 0; i < 9; ++i) ws[i] = wave_sum(vals[i]);
  if (nvals > 9) {
#pragma unroll
    for (int i = 9; i < 13; ++i) ws[i] = wave_sum(vals[i]);
  } else {
#pragma unroll
    for (int i = 9; i < 13; ++i) ws[i] = 0.f;
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 13; ++i) scratch[wave * 13 + i] = ws[i];
  }
  __syncthreads();
  // (all sixteen wave totals requested at once and then added in wave order, instead of this loop of dependent LDS round
  // trips, measured 0.3-0.4 us SLOWER per step: profiles/r03_ab_runs.txt)
  double tot = 0.0;
  if (tid < 13)
    for (int k = 0; k < nw; ++k) tot += (double)scratch[k * 13 + tid];
  return tot;
}

// d(q) from the moment matrix in fp64 (same algebra as quaternion_grad)
__device__ inline void quaternion_grad_f64(const float* q_raw, const double* m, double* dq) {
  double w = q_raw[0], vx = q_raw[1], vy = q_raw[2], vz = q_raw[3];
  const double inv_norm = 1.0 / sqrt(w * w + vx * vx + vy * vy + vz * vz);
  w *= inv_norm; vx *= inv_norm; vy *= inv_norm; vz *= inv_norm;
  const double tr = m[0] + m[4] + m[8];
  const double cx = m[5] - m[7], cy = m[6] - m[2], cz = m[1] - m[3];
  const double sx = 2.0 * m[0] * vx + (m[1] + m[3]) * vy + (m[2] + m[6]) * vz;
  const double sy = (m[3] + m[1]) * vx + 2.0 * m[4] * vy + (m[5] + m[7]) * vz;
  const double sz = (m[6] + m[2]) * vx + (m[7] + m[5]) * vy + 2.0 * m[8] * vz;
  const double gw = 2.0 * (w * tr + vx * cx + vy * cy + vz * cz);
  const double gx = 2.0 * (-tr * vx + sx + w * cx);
  const double gy = 2.0 * (-tr * vy + sy + w * cy);
  const double gz = 2.0 * (-tr * vz + sz + w * cz);
  const double dot = w * gw + vx * gx + vy * gy + vz * gz;
  dq[0] = (gw - w * dot) * inv_norm;
  dq[1] = (gx - vx * dot) * inv_norm;
  dq[2] = (gy - vy * dot) * inv_norm;
  dq[3] = (gz - vz * dot) * inv_norm;
}


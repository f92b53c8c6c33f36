// Stage-level entry points: one per reference function, for the sub-stage Python API
// (pc_perspective_transform, pointcloud2voxels3d_fast backward, smoothen_voxels3d, drc_*) and as an
// independent on-device cross-check of the fused path (dpc_entry.hip).  These favour generality (any grid
// size, any tap count up to DPC_MAX_TAPS) over fusion; the hot path does not go through them.
#include <math.h>

#include "dpc_common.h"

namespace {

constexpr int kThreads = 256;

int validate(const DpcParams* p) {
  if (p == nullptr) return DPC_ERR_NULL;
  if (p->B < 0 || p->N < 0 || p->D < 1 || p->H < 1 || p->W < 1) return DPC_ERR_SHAPE;
  if (p->D > 1024 || p->H > 1024 || p->W > 1024 || p->B > 65535) return DPC_ERR_SHAPE;  // 10-bit cell indices
  if (p->point_replicas < 0 || p->point_replicas > 1 || p->point_index != nullptr) return DPC_ERR_SHAPE;  // shared / indexed point sets: fused entry points only
  for (int taps : {p->taps_xy, p->taps_z})
    if (taps < 0 || taps > DPC_MAX_TAPS || (taps > 0 && taps % 2 == 0)) return DPC_ERR_TAPS;
  if ((p->dev_taps_xy != nullptr && p->taps_xy < 1) || (p->dev_taps_z != nullptr && p->taps_z < 1)) return DPC_ERR_TAPS;
  return DPC_OK;
}

int launch_ok() { return hipGetLastError() == hipSuccess ? DPC_OK : DPC_ERR_LAUNCH; }

// ------------------------------------------------------------------------------------------------------
// pc_perspective_transform                              dpc/util/point_cloud_to.py:118-178
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_transform_fwd(DpcParams P, const float* __restrict__ pc,
                                                            const float* __restrict__ q, const float* __restrict__ t,
                                                            const float* __restrict__ f, float* __restrict__ out) {
  const int b = blockIdx.y;
  const CameraRef cam = load_camera_ref(P, q, t, f, b);  // the reference's exact op sequence, rounded once on store
  const float* cloud = pc + (size_t)b * P.N * 3;
  float* o3 = out + (size_t)b * P.N * 3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.N; i += gridDim.x * blockDim.x) {
    double Z, Y, X;
    project_point_ref(cam, cloud[3 * i], cloud[3 * i + 1], cloud[3 * i + 2], Z, Y, X);
    o3[3 * i + 0] = (float)Z; o3[3 * i + 1] = (float)Y; o3[3 * i + 2] = (float)X;
  }
}

__global__ __launch_bounds__(kThreads) void k_transform_bwd(DpcParams P, const float* __restrict__ pc,
                                                            const float* __restrict__ q, const float* __restrict__ t,
                                                            const float* __restrict__ f,
                                                            const float* __restrict__ dout, float* __restrict__ dpc,
                                                            float* __restrict__ dsmall) {
  // one workgroup per cloud: the 13 camera-gradient sums are taken in a fixed order, in fp64 (no atomics)
  __shared__ float scratch[13 * (kThreads / DPC_WAVE)];
  const int b = blockIdx.x;
  const CameraRaw raw = load_camera_raw(P, q, t, f, b);
  const Camera cam = make_camera(P, raw);
  const float* cloud = pc + (size_t)b * P.N * 3;
  const float* g3 = dout + (size_t)b * P.N * 3;
  float* d3 = dpc + (size_t)b * P.N * 3;
  CamGrad g;
  camgrad_zero(g);
  for (int i = threadIdx.x; i < P.N; i += blockDim.x) {
    const float px = cloud[3 * i], py = cloud[3 * i + 1], pz = cloud[3 * i + 2];
    const Projected o = project_point(cam, px, py, pz);
    float dx, dy, dz;
    project_point_bwd(cam, o, px, py, pz, g3[3 * i], g3[3 * i + 1], g3[3 * i + 2], dx, dy, dz, g);
    d3[3 * i] = dx; d3[3 * i + 1] = dy; d3[3 * i + 2] = dz;
  }
  float vals[13];
#pragma unroll
  for (int i = 0; i < 9; ++i) vals[i] = g.m[i];
  vals[9] = g.dt[0]; vals[10] = g.dt[1]; vals[11] = g.dt[2]; vals[12] = g.df;
  const double tot = block_sum13_fixed(vals, scratch, threadIdx.x, kThreads);
  if (threadIdx.x < DPC_WAVE) {
    double m[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) m[i] = __shfl(tot, i, DPC_WAVE);
    if (threadIdx.x == 0) {
      double dq[4];
      quaternion_grad_f64(raw.q, m, dq);
      for (int i = 0; i < 4; ++i) dsmall[(size_t)DPC_COL_DQ * P.B + (size_t)b * 4 + i] = (float)dq[i];
      for (int i = 0; i < 3; ++i) dsmall[(size_t)DPC_COL_DT * P.B + (size_t)b * 3 + i] = (float)m[9 + i];
      dsmall[(size_t)DPC_COL_DF * P.B + b] = (float)m[12];
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// backward of pointcloud2voxels3d_fast: gather the 8 corners from a grid in global memory
// ------------------------------------------------------------------------------------------------------
template <class TIN>
__global__ __launch_bounds__(kThreads) void k_splat_bwd(DpcParams P, const TIN* __restrict__ tr,
                                                        const float* __restrict__ dvox, float* __restrict__ dtr) {
  const int D = P.D, H = P.H, W = P.W;
  const size_t total = (size_t)P.B * P.N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / P.N);
    const Cell c = cell_from_record(make_record((double)tr[3 * i], (double)tr[3 * i + 1], (double)tr[3 * i + 2], D, H, W));
    float dZ = 0.f, dY = 0.f, dX = 0.f;
    if (c.valid) {
      const float* gb = dvox + (size_t)b * D * H * W;
      float cv[2][2][2];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const bool ok = (c.iz + k < D) && (c.iy + j < H) && (c.ix + e < W);
            cv[k][j][e] = ok ? gb[((size_t)(c.iz + k) * H + c.iy + j) * W + c.ix + e] : 0.f;
          }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          dZ += (cv[1][a][e] - cv[0][a][e]) * c.wy[a] * c.wx[e];
          dY += (cv[a][1][e] - cv[a][0][e]) * c.wz[a] * c.wx[e];
          dX += (cv[a][e][1] - cv[a][e][0]) * c.wz[a] * c.wy[e];
        }
      dZ *= (float)(D - 1); dY *= (float)(H - 1); dX *= (float)(W - 1);
    }
    dtr[3 * i] = dZ; dtr[3 * i + 1] = dY; dtr[3 * i + 2] = dX;
  }
}

// ------------------------------------------------------------------------------------------------------
// smoothen_voxels3d: one zero-padded 1-D correlation along an axis of [B*D?, ...]   point_cloud_to.py:90-103
//   element (o, i, r): outer index o, position i along the axis (length len, stride `inner`), inner index r
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_conv_axis(const float* __restrict__ in, float* __restrict__ out,
                                                        size_t total, int len, int inner, TapsDyn taps_arg,
                                                        const float* __restrict__ dev_taps, int flip) {
  const TapsDyn taps = resolve_taps_dyn(taps_arg, dev_taps, flip != 0);   // DpcParams.dev_taps_*: values from device memory
  const int R = (taps.n - 1) / 2;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)((e / inner) % len);
    float acc = 0.f;
    for (int k = 0; k < taps.n; ++k) {
      const int p = i + k - R;
      if (p >= 0 && p < len) acc = fmaf(taps.w[k], in[e + (ptrdiff_t)(k - R) * inner], acc);
    }
    out[e] = acc;
  }
}

TapsDyn dyn_taps(const float* k, int n, bool flip) {
  TapsDyn t;
  t.n = n;
  for (int i = 0; i < DPC_MAX_TAPS; ++i) t.w[i] = 0.f;
  for (int i = 0; i < n; ++i) t.w[i] = k[flip ? n - 1 - i : i];
  return t;
}

// ------------------------------------------------------------------------------------------------------
// DRC, general form                                       dpc/util/drc.py:48-129,145-160
//   p_0 = e^eps y_0, p_k = y_k A_k (0<k<D), p_D = e^eps A_D, A_k = prod_{j<k} (1-y_j), y = clamp(v, eps, 1-eps)
// ------------------------------------------------------------------------------------------------------
__device__ inline float depth_psi(const DpcParams& P, int k) {
  return k < P.D ? (float)((double)k / (double)P.D - 0.5 + (double)P.camera_distance) : P.max_depth;
}

__global__ __launch_bounds__(kThreads) void k_drc_fwd(DpcParams P, const float* __restrict__ vox,
                                                      float* __restrict__ proj, float* __restrict__ probs,
                                                      float* __restrict__ depth) {
  const int HW = P.H * P.W, D = P.D;
  const size_t rays = (size_t)P.B * HW;
  const float eps = P.clip_val, hi = (float)(1.0 - (double)P.clip_val);
  const double e_eps = exp((double)P.clip_val);
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rays; r += (size_t)gridDim.x * blockDim.x) {
    const size_t b = r / HW, pix = r - b * HW;
    const float* col = vox + b * D * HW + pix;
    double A = 1.0, sum = 0.0, dsum = 0.0;
    for (int k = 0; k < D; ++k) {
      const float y = fminf(fmaxf(col[(size_t)k * HW], eps), hi);
      const double pk = (k == 0 ? e_eps : 1.0) * (double)y * A;
      if (probs) probs[(size_t)k * rays + r] = (float)pk;
      sum += pk;
      dsum += pk * (double)depth_psi(P, k);
      A *= 1.0 - (double)y;
    }
    const double pD = e_eps * A;
    if (probs) probs[(size_t)D * rays + r] = (float)pD;
    if (proj) proj[r] = (float)sum;
    if (depth) depth[r] = (float)(dsum + pD * (double)P.max_depth);
  }
}

// dL/dy_m = gp_m E_m A_m - (sum_{k>m} gp_k p_k) / (1 - y_m),  gp_k = dprobs_k + dproj [k<D] + ddepth psi_k.
// Pass 1 parks the prefix products A_m in dvox (fp32), pass 2 walks the ray backwards with the suffix sum.
__global__ __launch_bounds__(kThreads) void k_drc_bwd(DpcParams P, const float* __restrict__ vox,
                                                      const float* __restrict__ dproj,
                                                      const float* __restrict__ dprobs,
                                                      const float* __restrict__ ddepth, float* __restrict__ dvox) {
  const int HW = P.H * P.W, D = P.D;
  const size_t rays = (size_t)P.B * HW;
  const float eps = P.clip_val, hi = (float)(1.0 - (double)P.clip_val);
  const double e_eps = exp((double)P.clip_val);
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rays; r += (size_t)gridDim.x * blockDim.x) {
    const size_t b = r / HW, pix = r - b * HW;
    const float* col = vox + b * D * HW + pix;
    float* dcol = dvox + b * D * HW + pix;
    const double gproj = dproj ? (double)dproj[r] : 0.0;
    const double gdepth = ddepth ? (double)ddepth[r] : 0.0;
    double A = 1.0;
    for (int k = 0; k < D; ++k) {
      dcol[(size_t)k * HW] = (float)A;
      A *= 1.0 - (double)fminf(fmaxf(col[(size_t)k * HW], eps), hi);
    }
    const double gD = (dprobs ? (double)dprobs[(size_t)D * rays + r] : 0.0) + gdepth * (double)P.max_depth;
    double suffix = gD * e_eps * A;  // sum_{k>m} gp_k p_k
    for (int m = D - 1; m >= 0; --m) {
      const float v = col[(size_t)m * HW];
      const float y = fminf(fmaxf(v, eps), hi);
      const double Am = (double)dcol[(size_t)m * HW];
      const double E = m == 0 ? e_eps : 1.0;
      const double gp = (dprobs ? (double)dprobs[(size_t)m * rays + r] : 0.0) + gproj + gdepth * (double)depth_psi(P, m);
      const double dy = gp * E * Am - suffix / (1.0 - (double)y);
      dcol[(size_t)m * HW] = (v >= eps && v <= hi) ? (float)dy : 0.f;
      suffix += gp * E * (double)y * Am;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Silhouette loss of the caller (dpc/models/model_pc_to.py:339-385,410-440), fused with its gradient:
//   K == 1: loss = sum (gt - pred)^2 / S
//   K  > 1: per sample pick the candidate with the smallest sum of squared differences (argmin over K),
//           loss = sum over winners (gt - pred)^2 / S; losing candidates get zero gradient.
// gt [S, n_pix] (already pooled to the silhouette size), pred [S*K, n_pix].  One block per sample.
//   out: loss_part [S] (this sample's winning sum / S), winner [S] (int32), dpred [S*K, n_pix] (d loss / d pred)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_silhouette_loss(const float* __restrict__ gt, const float* __restrict__ pred,
                                                              int K, int n_pix, float inv_S,
                                                              float* __restrict__ loss_part, int* __restrict__ winner,
                                                              float* __restrict__ dpred) {
  __shared__ float red[kThreads / DPC_WAVE];
  __shared__ float best_val;
  __shared__ int best_k;
  const int smp = blockIdx.x;
  const float* g = gt + (size_t)smp * n_pix;
  if (threadIdx.x == 0) { best_val = 0.f; best_k = 0; }
  for (int k = 0; k < K; ++k) {
    const float* p = pred + ((size_t)smp * K + k) * n_pix;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_pix; i += blockDim.x) {
      const float d = g[i] - p[i];
      acc = fmaf(d, d, acc);
    }
    acc = wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot = 0.f;
      for (int w = 0; w < (int)(blockDim.x / DPC_WAVE); ++w) tot += red[w];
      if (k == 0 || tot < best_val) { best_val = tot; best_k = k; }  // first minimum wins, like torch.argmin
    }
  }
  __syncthreads();
  const int win = best_k;
  if (threadIdx.x == 0) {
    loss_part[smp] = best_val * inv_S;
    winner[smp] = win;
  }
  for (int k = 0; k < K; ++k) {
    const float* p = pred + ((size_t)smp * K + k) * n_pix;
    float* d = dpred + ((size_t)smp * K + k) * n_pix;
    const float scale = k == win ? 2.f * inv_S : 0.f;
    for (int i = threadIdx.x; i < n_pix; i += blockDim.x) d[i] = scale * (p[i] - g[i]);
  }
}

// ------------------------------------------------------------------------------------------------------
// pc_point_dropout's random choice (dpc/util/point_cloud_to.py:269-295: np.random.choice(N, n, replace=False) per cloud),
// drawn on the device: every point of a cloud gets a 32-bit key hashed from (seed, cloud, point); the cloud keeps the n
// points with the smallest keys (ties: lower index first) = a uniformly random n-subset.  One workgroup per cloud: a
// four-pass radix select finds the n-th smallest key (keys are recomputed, never stored), then an ordered compaction
// writes the kept indices in ascending order.  seed[2] lives on the device (the caller draws it from torch's generator,
// which is what makes the draw differ from replay to replay of a captured graph).
// ------------------------------------------------------------------------------------------------------
constexpr int kDropThreads = 1024;

__device__ inline uint32_t dropout_key(uint64_t s0, uint64_t s1, int cloud, int i) {
  uint64_t x = s0 ^ ((uint64_t)(uint32_t)cloud * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)i * 0xD1B54A32D192ED03ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;   // splitmix64 finaliser, twice, the second seed word in between
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  x += s1;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}

// n_cap: slots per output row; n_live (device, may be NULL): how many of them to fill in this launch
__global__ __launch_bounds__(kDropThreads) void k_dropout_indices(int N, int n_cap, const int32_t* __restrict__ n_live,
                                                                  const int64_t* __restrict__ seed,
                                                                  int32_t* __restrict__ out) {
  const int n = n_live != nullptr ? min(max(*n_live, 0), n_cap) : n_cap;
  __shared__ int hist[256];
  __shared__ uint32_t sel_prefix;
  __shared__ int sel_remaining;
  __shared__ int wave_less[kDropThreads / DPC_WAVE], wave_equal[kDropThreads / DPC_WAVE];
  __shared__ int base_out, base_equal;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (DPC_WAVE - 1), wave = tid / DPC_WAVE;
  const uint64_t s0 = (uint64_t)seed[0], s1 = (uint64_t)seed[1];
  if (tid == 0) { sel_prefix = 0u; sel_remaining = n; }
  // radix select, most significant byte first: after pass p the top 8(p+1) bits of the n-th smallest key are known
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = sel_prefix;
    const uint32_t high = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < N; i += kDropThreads) {
      const uint32_t k = dropout_key(s0, s1, b, i);
      if ((k & high) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int remaining = sel_remaining, d = 0;
      while (d < 255 && hist[d] < remaining) remaining -= hist[d++];
      sel_prefix = prefix | ((uint32_t)d << shift);
      sel_remaining = remaining;
    }
    __syncthreads();
  }
  const uint32_t T = sel_prefix;          // the n-th smallest key
  const int take_equal = sel_remaining;   // how many of the keys == T are kept (>= 1 when n > 0)
  if (tid == 0) { base_out = 0; base_equal = 0; }
  __syncthreads();
  int32_t* row = out + (size_t)b * n_cap;
  for (int i0 = 0; i0 < N; i0 += kDropThreads) {
    const int i = i0 + tid;
    const uint32_t k = i < N ? dropout_key(s0, s1, b, i) : 0xFFFFFFFFu;
    const bool less = i < N && k < T, equal = i < N && k == T;
    const uint64_t ml = __ballot(less), me = __ballot(equal);
    if (lane == 0) { wave_less[wave] = __popcll(ml); wave_equal[wave] = __popcll(me); }
    __syncthreads();
    int less_before = 0, equal_before = 0, less_all = 0, equal_all = 0;
    for (int w = 0; w < kDropThreads / DPC_WAVE; ++w) {
      if (w < wave) { less_before += wave_less[w]; equal_before += wave_equal[w]; }
      less_all += wave_less[w]; equal_all += wave_equal[w];
    }
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (DPC_WAVE - lane));
    const int my_equal = base_equal + equal_before + __popcll(me & below);   // rank among the keys == T, in index order
    const int kept_equal_before = min(base_equal + equal_before + __popcll(me & below), take_equal) - min(base_equal, take_equal);
    const bool keep = less || (equal && my_equal < take_equal);
    if (keep && n > 0) row[base_out + less_before + __popcll(ml & below) + kept_equal_before] = i;
    __syncthreads();
    if (tid == 0) {
      const int eq_kept = min(base_equal + equal_all, take_equal) - min(base_equal, take_equal);
      base_out += less_all + eq_kept;
      base_equal += equal_all;
    }
    __syncthreads();
  }
}

// Per-step schedule values into device memory (include/dpc_render.h, dpc_schedule_update): they arrive as kernel arguments
struct ScheduleArgs {
  float xy[DPC_MAX_TAPS], z[DPC_MAX_TAPS];
  int nxy, nz, n_live;
};
__global__ __launch_bounds__(64) void k_schedule_update(ScheduleArgs a, float* __restrict__ dxy, float* __restrict__ dz,
                                                        int32_t* __restrict__ dn) {
  const int t = threadIdx.x;
  if (dxy != nullptr && t < a.nxy) dxy[t] = a.xy[t];
  if (dz != nullptr && t < a.nz) dz[t] = a.z[t];
  if (dn != nullptr && t == 0) *dn = a.n_live;
}

int blocks_for(size_t n) { return (int)std::min<size_t>((n + kThreads - 1) / kThreads, 256 * 8); }

}  // namespace

extern "C" {

int dpc_abi_version(void) { return DPC_ABI_VERSION; }

const char* dpc_strerror(int code) {
  switch (code) {
    case DPC_OK: return "ok";
    case DPC_ERR_NULL: return "a required pointer is NULL";
    case DPC_ERR_SHAPE: return "B/N/D/H/W out of range (B <= 65535, N <= 1048575 points per cloud, grid sides 1..1024)";
    case DPC_ERR_TAPS: return "smoothing kernel length must be odd and <= DPC_MAX_TAPS (fused path: effective radius <= 15)";
    case DPC_ERR_LDS: return "an H x W plane does not fit the 160 KiB LDS tile";
    case DPC_ERR_LAUNCH: return "HIP kernel launch failed";
    case DPC_ERR_UNSUPPORTED: return "configuration is a dead branch of the reference";
    default: return "unknown dpc error code";
  }
}

int dpc_transform_fwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f, float* out,
                      void* stream) {
  int rc = validate(p);
  if (rc != DPC_OK) return rc;
  if (p->B == 0 || p->N == 0) return DPC_OK;
  if (!pc || !q || !out) return DPC_ERR_NULL;
  dim3 grid(std::max(1, std::min((p->N + kThreads - 1) / kThreads, 64)), p->B);
  hipLaunchKernelGGL(k_transform_fwd, grid, dim3(kThreads), 0, (hipStream_t)stream, *p, pc, q, t, f, out);
  return launch_ok();
}

int dpc_transform_bwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                      const float* dout, float* dpc, float* dsmall, void* stream) {
  int rc = validate(p);
  if (rc != DPC_OK) return rc;
  if (!pc || !q || !dout || !dpc || !dsmall) return DPC_ERR_NULL;
  if (p->B == 0) return DPC_OK;
  if (!zero_words_async(dsmall, (size_t)p->B * DPC_SMALL_COLS, (hipStream_t)stream))
    return DPC_ERR_LAUNCH;
  if (p->N == 0) return DPC_OK;
  hipLaunchKernelGGL(k_transform_bwd, dim3(p->B), dim3(kThreads), 0, (hipStream_t)stream, *p, pc, q, t, f, dout, dpc, dsmall);
  return launch_ok();
}

int dpc_point_dropout_indices_live(int B, int N, int n, const int32_t* n_live, const int64_t* seed, int32_t* out, void* stream) {
  if (B < 0 || N < 0 || n < 0 || n > N) return DPC_ERR_SHAPE;
  if (B == 0 || n == 0) return DPC_OK;
  if (!seed || !out) return DPC_ERR_NULL;
  hipLaunchKernelGGL(k_dropout_indices, dim3(B), dim3(kDropThreads), 0, (hipStream_t)stream, N, n, n_live, seed, out);
  return launch_ok();
}

int dpc_point_dropout_indices(int B, int N, int n, const int64_t* seed, int32_t* out, void* stream) {
  return dpc_point_dropout_indices_live(B, N, n, nullptr, seed, out, stream);
}

int dpc_schedule_update(const float* host_kern_xy, int taps_xy, const float* host_kern_z, int taps_z, int n_live,
                        float* dev_taps_xy, float* dev_taps_z, int32_t* dev_n_live, void* stream) {
  if (taps_xy < 0 || taps_z < 0 || taps_xy > DPC_MAX_TAPS || taps_z > DPC_MAX_TAPS) return DPC_ERR_TAPS;
  if ((dev_taps_xy && taps_xy > 0 && !host_kern_xy) || (dev_taps_z && taps_z > 0 && !host_kern_z)) return DPC_ERR_NULL;
  if (!dev_taps_xy && !dev_taps_z && !dev_n_live) return DPC_OK;
  ScheduleArgs a;
  a.nxy = dev_taps_xy ? taps_xy : 0;
  a.nz = dev_taps_z ? taps_z : 0;
  a.n_live = n_live;
  for (int i = 0; i < DPC_MAX_TAPS; ++i) {
    a.xy[i] = i < a.nxy ? host_kern_xy[i] : 0.f;
    a.z[i] = i < a.nz ? host_kern_z[i] : 0.f;
  }
  hipLaunchKernelGGL(k_schedule_update, dim3(1), dim3(64), 0, (hipStream_t)stream, a, dev_taps_xy, dev_taps_z, dev_n_live);
  return launch_ok();
}

int dpc_splat_bwd(const DpcParams* p, const void* tr, int tr_is_f64, const float* dvox, float* dtr, void* stream) {
  int rc = validate(p);
  if (rc != DPC_OK) return rc;
  const size_t total = (size_t)p->B * p->N;
  if (total == 0) return DPC_OK;
  if (!tr || !dvox || !dtr) return DPC_ERR_NULL;
  if (tr_is_f64)
    hipLaunchKernelGGL(k_splat_bwd<double>, dim3(blocks_for(total)), dim3(kThreads), 0, (hipStream_t)stream, *p,
                       static_cast<const double*>(tr), dvox, dtr);
  else
    hipLaunchKernelGGL(k_splat_bwd<float>, dim3(blocks_for(total)), dim3(kThreads), 0, (hipStream_t)stream, *p,
                       static_cast<const float*>(tr), dvox, dtr);
  return launch_ok();
}

int dpc_smooth(const DpcParams* p, const float* host_kern_xy, const float* host_kern_z, int transpose, const float* in,
               float* out, float* tmp, void* stream) {
  int rc = validate(p);
  if (rc != DPC_OK) return rc;
  if (!in || !out || !tmp) return DPC_ERR_NULL;
  // taps_xy == 0 / taps_z == 0: that group of axes is left alone (e.g. the D pass alone, on a grid that already went
  // through the W and H passes); at least one group must be given
  const bool do_xy = p->taps_xy > 0, do_z = p->taps_z > 0;
  if ((!do_xy && !do_z) || (do_xy && !host_kern_xy) || (do_z && !host_kern_z)) return DPC_ERR_TAPS;
  const size_t total = (size_t)p->B * p->D * p->H * p->W;
  if (total == 0) return DPC_OK;
  const bool flip = transpose != 0;
  const TapsDyn kxy = do_xy ? dyn_taps(host_kern_xy, p->taps_xy, flip) : TapsDyn{}, kz = do_z ? dyn_taps(host_kern_z, p->taps_z, flip) : TapsDyn{};
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(blocks_for(total)), blk(kThreads);
  // reference order W, H, D (point_cloud_to.py:92-97); the adjoint runs D, H, W with flipped taps
  if (!do_xy) {
    hipLaunchKernelGGL(k_conv_axis, g, blk, 0, st, in, out, total, p->D, p->H * p->W, kz, p->dev_taps_z, (int)flip);
  } else if (!flip) {
    hipLaunchKernelGGL(k_conv_axis, g, blk, 0, st, in, do_z ? out : tmp, total, p->W, 1, kxy, p->dev_taps_xy, (int)flip);
    hipLaunchKernelGGL(k_conv_axis, g, blk, 0, st, (const float*)(do_z ? out : tmp), do_z ? tmp : out, total, p->H, p->W, kxy, p->dev_taps_xy, (int)flip);
    if (do_z) hipLaunchKernelGGL(k_conv_axis, g, blk, 0, st, (const float*)tmp, out, total, p->D, p->H * p->W, kz, p->dev_taps_z, (int)flip);
  } else {
    const float* src = in;
    if (do_z) {
      hipLaunchKernelGGL(k_conv_axis, g, blk, 0, st, in, out, total, p->D, p->H * p->W, kz, p->dev_taps_z, (int)flip);
      src = out;
    }
    hipLaunchKernelGGL(k_conv_axis, g, blk, 0, st, src, tmp, total, p->H, p->W, kxy, p->dev_taps_xy, (int)flip);
    hipLaunchKernelGGL(k_conv_axis, g, blk, 0, st, (const float*)tmp, out, total, p->W, 1, kxy, p->dev_taps_xy, (int)flip);
  }
  return launch_ok();
}

int dpc_silhouette_loss(const float* gt, const float* pred, int S, int K, int n_pix, float* loss_part, int32_t* winner,
                        float* dpred, void* stream) {
  if (S < 0 || K < 1 || n_pix < 1) return DPC_ERR_SHAPE;
  if (S == 0) return DPC_OK;
  if (!gt || !pred || !loss_part || !winner || !dpred) return DPC_ERR_NULL;
  hipLaunchKernelGGL(k_silhouette_loss, dim3(S), dim3(kThreads), 0, (hipStream_t)stream, gt, pred, K, n_pix,
                     1.0f / (float)S, loss_part, winner, dpred);
  return launch_ok();
}

int dpc_drc_fwd(const DpcParams* p, const float* vox, float* proj, float* probs, float* depth, void* stream) {
  int rc = validate(p);
  if (rc != DPC_OK) return rc;
  if (!vox) return DPC_ERR_NULL;
  const size_t rays = (size_t)p->B * p->H * p->W;
  if (rays == 0) return DPC_OK;
  hipLaunchKernelGGL(k_drc_fwd, dim3(blocks_for(rays)), dim3(kThreads), 0, (hipStream_t)stream, *p, vox, proj, probs,
                     depth);
  return launch_ok();
}

int dpc_drc_bwd(const DpcParams* p, const float* vox, const float* dproj, const float* dprobs, const float* ddepth,
                float* dvox, void* stream) {
  int rc = validate(p);
  if (rc != DPC_OK) return rc;
  if (!vox || !dvox) return DPC_ERR_NULL;
  const size_t rays = (size_t)p->B * p->H * p->W;
  if (rays == 0) return DPC_OK;
  hipLaunchKernelGGL(k_drc_bwd, dim3(blocks_for(rays)), dim3(kThreads), 0, (hipStream_t)stream, *p, vox, dproj, dprobs,
                     ddepth, dvox);
  return launch_ok();
}

}  // extern "C"

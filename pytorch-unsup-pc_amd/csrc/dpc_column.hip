// Column kernels: one lane per (y, x) ray, the z column in registers.  D pass + occupancy scale/clamp + DRC silhouette
// (k_zcol_fwd), its backward (k_zcol_bwd), both in one launch for the fused one-candidate loss (k_zcol_fwdbwd), and the
// min-of-K selection (k_loss_finalize).  Design notes: DESIGN.md section 4.
#include "dpc_kernels.h"


DPC_DEBUG_SETTERS(col)

namespace dpck {
namespace {

// The unfused ray march leaves one squared-error partial per (cloud, ray tile); here they are added in tile order (so the
// clouds' sums, the winners and the loss are the same bits on every run), the best pose candidate of every sample is
// picked and the loss formed.  One block.
// (as a device function: k_zcol_bwd's first workgroup runs the same code when the selection is folded into the backward)
__device__ inline void loss_finalize_block(const float* __restrict__ sse_tiles, int ntile, float* __restrict__ sse, int S, int K,
                                           float inv_S, float* __restrict__ loss, int* __restrict__ winner, float* red) {
  // one thread per cloud adds its tiles (independent loads, tile order), then one thread per sample picks the winner
  for (int cloud = threadIdx.x; cloud < S * K; cloud += blockDim.x) {
    float v = 0.f;
    for (int i = 0; i < ntile; ++i) v += sse_tiles[(size_t)cloud * ntile + i];
    sse[cloud] = v;
  }
  __syncthreads();  // sse[] was written by this block: visible to its threads behind the barrier
  float acc = 0.f;
  for (int smp = threadIdx.x; smp < S; smp += blockDim.x) {
    float best = 0.f;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
      const float v = sse[(size_t)smp * K + k];
      if (k == 0 || v < best) { best = v; bk = k; }  // first minimum wins, like torch.argmin
    }
    winner[smp] = bk;
    acc += best;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < 256 / DPC_WAVE; ++i) tot += red[i];
    *loss = tot * inv_S;
  }
}

__global__ __launch_bounds__(256) void k_loss_finalize(const float* __restrict__ sse_tiles, int ntile, float* __restrict__ sse,
                                                       int S, int K, float inv_S, float* __restrict__ loss,
                                                       int* __restrict__ winner) {
  __shared__ float red[256 / DPC_WAVE];
  loss_finalize_block(sse_tiles, ntile, sse, S, K, inv_S, loss, winner, red);
}

// ------------------------------------------------------------------------------------------------------
// Forward 2: D-pass + scale/clamp + DRC silhouette, whole z column in registers.   grid (ceil(HW/256), B)
// ------------------------------------------------------------------------------------------------------
// Epilogue shared by the forward column kernels: silhouette (row flip folded into the index), saved ray
// transmittance, fused loss partial.
__device__ inline void zcol_fwd_epilogue(const DpcParams& P, const RayConst& rc, const Blk& bk, int ray, bool live, double trans,
                                         float y0, float* __restrict__ proj, float* __restrict__ trans_out,
                                         const LossArgs& la) {
  const int HW = P.H * P.W, b = bk.y;
  float sq = 0.f;
  if (live) {
    const int yrow = ray / P.W, x = ray - yrow * P.W;
    const int pix = (P.H - 1 - yrow) * P.W + x;
    float pr = (float)(1.0 - trans + (double)rc.em1 * (double)y0);
    if (!(fabsf(rc.s) <= 3.4e38f)) pr = __builtin_nanf("");  // NaN / infinite scale: NaN silhouette, like the reference's clamps
    proj[(size_t)b * HW + pix] = pr;
    if (trans_out != nullptr) trans_out[(size_t)b * HW + ray] = (float)trans;
    if (la.gt != nullptr) {
      const float d = la.gt[(size_t)(b / la.K) * HW + pix] - pr;
      sq = d * d;
    }
  }
  if (la.gt != nullptr) {  // block-uniform
    __shared__ float red[kColThreads / DPC_WAVE];
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot = 0.f;
      for (int i = 0; i < kColThreads / DPC_WAVE; ++i) tot += red[i];
      la.sse_tiles[(size_t)b * bk.nx + bk.x] = tot;  // no atomics: k_loss_finalize adds the tiles in order
    }
  }
}

template <int DD, int RB>
__global__ __launch_bounds__(kColThreads, (DD <= 64 ? 4 : 2)) void k_zcol_fwd(DpcParams P, RayHost rh, const float* __restrict__ Tbuf,
                                                          const float* __restrict__ s, TapsT<RB> taps_arg,
                                                          float* __restrict__ smoothed, float* __restrict__ proj,
                                                          float* __restrict__ trans_out, LossArgs la) {
  const TapsT<RB> taps = resolve_taps<RB>(taps_arg, P.dev_taps_z, P.taps_z, false);
  const int HW = P.H * P.W;
  const Blk bk = block_coords(P.B);
  const int b = bk.y, ray = bk.x * kColThreads + threadIdx.x;
  const bool live = ray < HW;
  const RayConst rc = ray_const(rh, s, b);
  double trans = 1.0;
  float y0 = 0.f;
  if (live) {
    float* out = smoothed + (size_t)b * DD * HW + ray;
    float c[DD];
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Tbuf + (size_t)b * DD * HW), 0, DD * HW * 4, 0x00020000);
#pragma unroll
    for (int z = 0; z < DD; ++z) c[z] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src, ray * 4, z * HW * 4, 0));  // one lane offset, plane offsets in SGPRs
#pragma unroll
    for (int z = 0; z < DD; ++z) {
      float v2 = 0.f;
#pragma unroll
      for (int k = 0; k < 2 * RB + 1; ++k) {
        const int zz = z + k - RB;
        if (zz >= 0 && zz < DD) v2 = fmaf(taps.w[k], c[zz], v2);
      }
      if (smoothed != nullptr) out[(size_t)z * HW] = v2;
      const float y = drc_clamp(rc, occupancy(rc, v2));
      if (z == 0) y0 = y;
      trans *= 1.0 - (double)y;
    }
  }
  zcol_fwd_epilogue(P, rc, bk, ray, live, trans, y0, proj, trans_out, la);
}

// Forward 2 + Backward 1 in one launch (fused loss, one pose candidate per sample): the gradient arriving at the
// silhouette, 2 (proj - gt) / S * dloss, is linear in the scalar dloss, and this kernel already holds the ray's whole
// column in registers -- so it runs the DRC backward and the adjoint D pass right away for dloss = 1 and writes dT.
// The backward proper is then k_gather_hw alone, which multiplies by the dloss that actually arrives.  Saves a launch
// and a second full read of the W/H grid.  Also zeroes the dq/dt/df accumulators and writes the ds partials.
//
// Written one ray per lane with a forward pass, a recomputed D pass and a branch-free but long DRC backward, this
// kernel was VALU-bound (measured: 6 us of loads, 12 us of arithmetic), so the arithmetic is cut to the bone instead:
// per voxel the forward leaves a single value behind, the clamped occupancy y, with "the clamps acted" (no gradient)
// encoded as y = -inf: then 1 - y = +inf, rcp gives 0, and the backward needs no compare/select and no second D pass.
//   y = med3(s v2, eps, 1-eps)  [= the reference's clamp(clamp(s v2, 0, 1), eps, 1-eps)],  inside <=> y == s v2
//   dL/dv3 = g T / (1 - y) (+ g (e^eps - 1) for the first voxel),  dL/ds = sum y dL/dv3 / s  (inside: v2 = y / s)
// RPL = rays per lane: neighbouring rays x .. x+RPL-1 (RPL divides W, so they share an image row).
#ifndef DPC_ZFB_2WAVE_MAX
#define DPC_ZFB_2WAVE_MAX 128  // deepest column that is compiled for two waves per SIMD (256 VGPRs per lane)
#endif
template <int DD, int RB, int RPL>
__global__ __launch_bounds__(kColThreads, (RPL * DD <= DPC_ZFB_2WAVE_MAX ? 2 : 1))
void k_zcol_fwdbwd(DpcParams P, RayHost rh, const float* __restrict__ Tbuf, const float* __restrict__ s, TapsT<RB> taps_arg,
                   TapsT<RB> taps_adj_arg, float* __restrict__ proj, float* __restrict__ dT, float* __restrict__ ds_part,
                   int n_ds_part, unsigned long long* __restrict__ tickets, SseFormat cf, SseFormat bf, float* __restrict__ dsmall,
                   unsigned int* __restrict__ cg_count, LossArgs la) {
  typedef float vec __attribute__((ext_vector_type(RPL)));
  const TapsT<RB> taps = resolve_taps<RB>(taps_arg, P.dev_taps_z, P.taps_z, false);
  const TapsT<RB> taps_adj = resolve_taps<RB>(taps_adj_arg, P.dev_taps_z, P.taps_z, true);
  const int HW = P.H * P.W;
  const Blk bk = block_coords(P.B);
  const int b = bk.y, ray = RPL * (bk.x * kColThreads + threadIdx.x);
  const bool live = ray < HW;
  const RayConst rc = ray_const(rh, s, b);
  float sq = 0.f, ds_acc = 0.f;
  float y[DD][RPL], g[RPL], gT[RPL];
  if (live) {
    // Column loads and dT stores go through buffer descriptors of this cloud's two grids: ONE 32-bit lane offset (the ray)
    // for all of them, the plane offsets z*HW*4 in SGPRs -- the flat form cost a 64-bit vector add per access (135 of the
    // kernel's 1 640 issue-bound VALU instructions).
    const float* col = Tbuf + (size_t)b * DD * HW + ray;
    float c[DD][RPL];  // T column, then y (encoded), then dL/dv3: each value dies as the next is born
    if constexpr (RPL == 1) {
      const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Tbuf + (size_t)b * DD * HW), 0, DD * HW * 4, 0x00020000);
#pragma unroll
      for (int z = 0; z < DD; ++z) c[z][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src, ray * 4, z * HW * 4, 0));
    } else {
#pragma unroll
      for (int z = 0; z < DD; ++z) {
        const vec v = *reinterpret_cast<const vec*>(col + (size_t)z * HW);
#pragma unroll
        for (int r = 0; r < RPL; ++r) c[z][r] = v[r];
      }
    }
    if (DPC_ABL(16)) {  // diagnostic: loads only
      float sum = 0.f;
#pragma unroll
      for (int z = 0; z < DD; ++z)
#pragma unroll
        for (int r = 0; r < RPL; ++r) sum += c[z][r];
      if (sum == 123.456f) proj[0] = sum;
      return;
    }
    const float ninf = -__builtin_inff();
    double tr[RPL];
    float yfirst[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) tr[r] = 1.0;
#pragma unroll
    for (int z = 0; z < DD; ++z) {
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        float v2 = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * RB + 1; ++k) {
          const int zz = z + k - RB;
          if (zz >= 0 && zz < DD) v2 = fmaf(taps.w[k], c[zz][r], v2);
        }
        const float x = v2 * rc.s;  // s = 1 when there is no scale input
        const float yc = __builtin_amdgcn_fmed3f(x, rc.eps, rc.hi);
        tr[r] = fma(-(double)yc, tr[r], tr[r]);  // T *= 1 - y, one rounding
        y[z][r] = (yc == x) ? yc : ninf;
        if (z == 0) yfirst[r] = yc;
      }
      if ((z & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    // silhouettes (row flip folded into the index), squared error, d loss / d proj for dloss = 1
    const int yrow = ray / P.W, xcol = ray - yrow * P.W;
    const int pix = (P.H - 1 - yrow) * P.W + xcol;
    vec pr;
#pragma unroll
    for (int r = 0; r < RPL; ++r) pr[r] = (float)(1.0 - tr[r] + (double)rc.em1 * (double)yfirst[r]);
    if (!(fabsf(rc.s) <= 3.4e38f))  // a NaN or infinite occupancy scale: the reference's clamps hand the NaN on, med3 would not
#pragma unroll
      for (int r = 0; r < RPL; ++r) pr[r] = __builtin_nanf("");
    *reinterpret_cast<vec*>(proj + (size_t)b * HW + pix) = pr;
    if (DPC_ABL(17)) return;  // diagnostic: forward only
    const vec gtv = *reinterpret_cast<const vec*>(la.gt + (size_t)b * HW + pix);  // K == 1: sample == cloud
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const float diff = pr[r] - gtv[r];
      sq = fmaf(diff, diff, sq);
      g[r] = 2.0f * la.inv_S * diff;
      gT[r] = g[r] * (float)tr[r];
    }
  }
  // Squared error of the cloud and the loss.  Float atomics from every block onto sse[b] and the one loss word cost
  // 3.4 us here (device-scope atomics execute memory-side, ~7 ns apiece on one address, and the issuing wave's stores
  // queue behind them), and a release/acquire hand-over between blocks costs an L2 write-back per block (the eight
  // XCD L2s are not coherent with each other).  So each block makes ONE relaxed 64-bit atomic add to its cloud's word:
  // the squared error in fixed point plus a block count above it (SseFormat in dpc_kernels.h).  The returned value
  // is looked at only after the backward half; whoever drew the last ticket holds the cloud's complete sum -- exact
  // integer adds, so sse[b] does not depend on the order the blocks arrived in -- and makes the cloud's single add
  // to the loss.
  __shared__ float red[2][kColThreads / DPC_WAVE];
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = sq;
  __syncthreads();
  unsigned long long mine = 0ull, before = 0ull;
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < kColThreads / DPC_WAVE; ++i) tot += red[0][i];
    mine = sse_share(cf, (double)tot, (double)kTileSseCap);
    before = __hip_atomic_fetch_add(tickets + b, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (live) {
    float dsv[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) dsv[r] = 0.f;
    float wadj[2 * RB + 1];  // adjoint taps with the occupancy scale folded in: dT = s * adj(dL/dv3)
#pragma unroll
    for (int k = 0; k < 2 * RB + 1; ++k) wadj[k] = taps_adj.w[k] * rc.s;
    float* out = dT + (size_t)b * DD * HW + ray;
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(dT + (size_t)b * DD * HW, 0, DD * HW * 4, 0x00020000);
#pragma unroll
    for (int z = 0; z < DD + RB; ++z) {
      if (z < DD) {
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          const float yv = y[z][r];
          float e = gT[r] * __builtin_amdgcn_rcpf(1.0f - yv);  // 1 - (-inf) = +inf -> 0
          if (z == 0) e += yv > 0.f ? g[r] * rc.em1 : 0.f;
          dsv[r] = fmaf(__builtin_amdgcn_fmed3f(yv, 0.f, 1.f), e, dsv[r]);  // -inf -> 0
          y[z][r] = e;  // y[z] is dead from here on: its register carries dL/dv3 for the adjoint window
        }
      }
      if (z >= RB) {
        const int zo = z - RB;
        vec acc;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          float a = 0.f;
#pragma unroll
          for (int i = 0; i < 2 * RB + 1; ++i) {
            const int k = tap_edge_first<RB>(i), zz = zo + k - RB;   // adjoint D pass: edges first, centre last (dpc_common.h)
            if (zz >= 0 && zz < DD) a = fmaf(wadj[k], y[zz][r], a);
          }
          acc[r] = a;
        }
        if (!DPC_ABL(18) || acc[0] == 123.456f) {
          if constexpr (RPL == 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, (float)acc[0]), dst, ray * 4, zo * HW * 4, kAuxThrough);
          else *reinterpret_cast<vec*>(out + (size_t)zo * HW) = acc;
        }
      }
      if ((z & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    float dsum = 0.f;
#pragma unroll
    for (int r = 0; r < RPL; ++r) dsum += dsv[r];
    ds_acc = (rc.has_s && rc.s != 0.f) ? dsum / rc.s : 0.f;
  }
  if (DPC_ABL(19)) { if (ds_acc == 123.456f) proj[1] = sq; return; }
  // v2 * dL/dv3 -> this tile's ds partial
  ds_acc = wave_sum(ds_acc);
  if ((threadIdx.x & 63) == 0) red[1][threadIdx.x >> 6] = ds_acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float dst = 0.f;
    for (int i = 0; i < kColThreads / DPC_WAVE; ++i) dst += red[1][i];
    // k_gather_hw sums n_ds_part partials per cloud (one per kColThreads rays); with RPL > 1 this grid has fewer blocks
    for (int i = bk.x; i < n_ds_part; i += bk.nx) ds_part[(size_t)b * n_ds_part + i] = i == bk.x ? dst : 0.f;
    if (sse_complete(cf, before, bk.nx)) {  // every other block of this cloud has added its share
      const double tot = sse_total(cf, before + mine);   // NaN if any tile's share was not a representable number
      la.sse[b] = (float)tot;
      // The batch loss the same way, one level up: the clouds' exact sums go into one more 64-bit word and the cloud that
      // arrives last writes the loss -- integer adds again, so the loss is bit-identical from run to run (float atomics
      // here differed in the last bits with the arrival order).
      const unsigned long long cmine = sse_share(bf, tot, (double)kTileSseCap * bk.nx);
      const unsigned long long cbefore = __hip_atomic_fetch_add(tickets + bk.ny, cmine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (sse_complete(bf, cbefore, bk.ny)) *la.loss_direct = (float)(sse_total(bf, cbefore + cmine) * (double)la.inv_S);
    }
  }
  if (bk.x == 0 && threadIdx.x < DPC_SMALL_COLS) dsmall[(size_t)threadIdx.x * bk.ny + b] = 0.f;  // [col][B]
  if (bk.x == 0 && threadIdx.x == 0) cg_count[b] = 0u;  // k_gather_hw's arrival counter of this cloud
}

// Generic depth / tap count: same arithmetic, column re-read from global (L1/L2 serve the re-reads).
__global__ __launch_bounds__(kColThreads) void k_zcol_fwd_dyn(DpcParams P, RayHost rh, const float* __restrict__ Tbuf,
                                                              const float* __restrict__ s, TapsDyn taps_arg,
                                                              float* __restrict__ smoothed, float* __restrict__ proj,
                                                              float* __restrict__ trans_out, LossArgs la) {
  const TapsDyn taps = resolve_taps_dyn(taps_arg, P.dev_taps_z, false);
  const int HW = P.H * P.W, D = P.D;
  const Blk bk = block_coords(P.B);
  const int b = bk.y, ray = bk.x * kColThreads + threadIdx.x;
  const bool live = ray < HW;
  const RayConst rc = ray_const(rh, s, b);
  double trans = 1.0;
  float y0 = 0.f;
  if (live) {
    const float* col = Tbuf + (size_t)b * D * HW + ray;
    float* out = smoothed + (size_t)b * D * HW + ray;
    const int R = taps.n > 0 ? (taps.n - 1) / 2 : 0;
    for (int z = 0; z < D; ++z) {
      float v2;
      if (taps.n == 0) {
        v2 = col[(size_t)z * HW];
      } else {
        v2 = 0.f;
        for (int k = 0; k < taps.n; ++k) {
          const int zz = z + k - R;
          if (zz >= 0 && zz < D) v2 = fmaf(taps.w[k], col[(size_t)zz * HW], v2);
        }
      }
      if (smoothed != nullptr) out[(size_t)z * HW] = v2;
      const float y = drc_clamp(rc, occupancy(rc, v2));
      if (z == 0) y0 = y;
      trans *= 1.0 - (double)y;
    }
  }
  zcol_fwd_epilogue(P, rc, bk, ray, live, trans, y0, proj, trans_out, la);
}

// ------------------------------------------------------------------------------------------------------
// Backward 1: DRC backward + scale/clamp backward + adjoint D-pass.                grid (ceil(HW/256), B)
//   Also zeroes the dq/dt/df accumulators that k_gather_hw adds into, and writes this tile's ds partial.
// ------------------------------------------------------------------------------------------------------
// Gradient arriving at this ray's silhouette pixel: either read from dproj, or formed from the fused loss.
__device__ inline float ray_grad(const DpcParams& P, const LossArgs& la, const float* __restrict__ dproj,
                                 const float* __restrict__ proj, int b, int ray) {
  const int HW = P.H * P.W;
  const int yrow = ray / P.W, x = ray - yrow * P.W;
  const int pix = (P.H - 1 - yrow) * P.W + x;
  if (la.gt == nullptr) return dproj[(size_t)b * HW + pix];
  const float up = la.dloss ? *la.dloss : 1.0f;
  return 2.0f * la.inv_S * up * (proj[(size_t)b * HW + pix] - la.gt[(size_t)(b / la.K) * HW + pix]);
}

// b: the cloud of this workgroup; the small gradients of clouds [zero_lo, zero_lo + zero_n) are zeroed by part 0
__device__ inline void zcol_bwd_epilogue(float ds_acc, float* ds_part, float* dsmall, unsigned int* cg_count, const Blk& bk,
                                         int b, int B, int zero_lo, int zero_n) {
  __shared__ float red[kColThreads / DPC_WAVE];
  const float w = wave_sum(ds_acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < kColThreads / DPC_WAVE; ++i) tot += red[i];
    ds_part[(size_t)b * bk.nx + bk.x] = tot;
  }
  if (bk.x == 0)
    for (int i = threadIdx.x; i < DPC_SMALL_COLS * zero_n; i += kColThreads)
      dsmall[(size_t)(i / zero_n) * B + zero_lo + i % zero_n] = 0.f;  // [col][B]
  if (bk.x == 0 && threadIdx.x == 0) cg_count[b] = 0u;  // k_gather_hw's arrival counter of this cloud
}

// Reads the grid saved by the forward slab kernel (after clamp + W/H passes), recomputes the forward D-pass in
// registers (cheaper than having the forward write, and this kernel read, a second full grid), then DRC backward,
// scale/clamp backward and the adjoint D-pass.
template <int DD, int RB>
__global__ __launch_bounds__(kColThreads, 2) void k_zcol_bwd(DpcParams P, RayHost rh, const float* __restrict__ Tin,
                                                          const float* __restrict__ s,
                                                          const float* __restrict__ dproj, const float* __restrict__ proj,
                                                          const float* __restrict__ trans_in, TapsT<RB> taps_arg,
                                                          TapsT<RB> taps_adj_arg,
                                                          float* __restrict__ dT, float* __restrict__ ds_part,
                                                          float* __restrict__ dsmall, unsigned int* __restrict__ cg_count,
                                                          const float* __restrict__ dgrid_extra, LossArgs la) {
  const TapsT<RB> taps = resolve_taps<RB>(taps_arg, P.dev_taps_z, P.taps_z, false);
  const TapsT<RB> taps_adj = resolve_taps<RB>(taps_adj_arg, P.dev_taps_z, P.taps_z, true);
  const int HW = P.H * P.W;
  const bool wo = winners_only(la);  // grid over samples: this workgroup works on the winning candidate of sample bk.y
  const Blk bk = block_coords(wo ? P.B / la.K : P.B);
  int win = 0;
  if (la.winner_write != nullptr) {
    // the min-of-K selection folded into this launch: the candidates' squared errors are the ray tiles' partials added in
    // tile order (k_loss_finalize's arithmetic); every workgroup of a sample arrives at the same winner
    __shared__ float cand[kColThreads];
    __shared__ float fin_red[kColThreads / DPC_WAVE];
    __shared__ int s_win;
    const int tid = threadIdx.x;
    for (int k = tid; k < la.K; k += kColThreads) {
      float v = 0.f;
      for (int i = 0; i < la.ntile; ++i) v += la.sse_tiles[((size_t)bk.y * la.K + k) * la.ntile + i];
      if (k < kColThreads) cand[k] = v;
    }
    __syncthreads();
    if (tid == 0) {
      float best = 0.f;
      int bi = 0;
      for (int k = 0; k < la.K; ++k) {
        const float v = cand[k];
        if (k == 0 || v < best) { best = v; bi = k; }
      }
      s_win = bi;
    }
    __syncthreads();
    win = s_win;
    if (blockIdx.x == 0)   // ... and ONE workgroup leaves sse, winner and the loss behind, bit for bit what the separate launch would
      loss_finalize_block(la.sse_tiles, la.ntile, la.sse, bk.ny, la.K, la.inv_S, la.loss_write, la.winner_write, fin_red);
  } else if (wo) {
    win = la.winner[bk.y];
  }
  const int b = wo ? bk.y * la.K + win : bk.y, ray = bk.x * kColThreads + threadIdx.x;
  float ds_acc = 0.f;
  if (ray < HW && (wo || !cloud_loses(la, b))) {
    const RayConst rc = ray_const(rh, s, b);
    float c[DD], d[DD];
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Tin + (size_t)b * DD * HW), 0, DD * HW * 4, 0x00020000);
#pragma unroll
    for (int z = 0; z < DD; ++z) c[z] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src, ray * 4, z * HW * 4, 0));  // one lane offset, plane offsets in SGPRs
    float Tf;
    if (trans_in != nullptr) {
      Tf = trans_in[(size_t)b * HW + ray];  // saved by the forward
    } else {
      double trans = 1.0;
#pragma unroll
      for (int z = 0; z < DD; ++z) {
        float v2 = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * RB + 1; ++k) {
          const int zz = z + k - RB;
          if (zz >= 0 && zz < DD) v2 = fmaf(taps.w[k], c[zz], v2);
        }
        trans *= 1.0 - (double)drc_clamp(rc, occupancy(rc, v2));
        if ((z & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      Tf = (float)trans;
    }
    const float g = ray_grad(P, la, dproj, proj, b, ray);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(dT + (size_t)b * DD * HW, 0, DD * HW * 4, 0x00020000);
    const float* extra = dgrid_extra ? dgrid_extra + (size_t)b * DD * HW + ray : nullptr;  // gradient arriving at grid_wh itself
    // streaming over z: forward taps -> d(v2) -> adjoint taps, RB voxels behind
#pragma unroll
    for (int z = 0; z < DD + RB; ++z) {
      if (z < DD) {
        float v2 = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * RB + 1; ++k) {
          const int zz = z + k - RB;
          if (zz >= 0 && zz < DD) v2 = fmaf(taps.w[k], c[zz], v2);
        }
        float term;
        d[z] = drc_voxel_bwd(rc, v2, g, Tf, z == 0, term);
        ds_acc += term;
      }
      if (z >= RB) {
        const int zo = z - RB;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 2 * RB + 1; ++i) {
          const int k = tap_edge_first<RB>(i), zz = zo + k - RB;   // adjoint D pass: edges first, centre last (dpc_common.h)
          if (zz >= 0 && zz < DD) acc = fmaf(taps_adj.w[k], d[zz], acc);
        }
        if (extra != nullptr) acc += extra[(size_t)zo * HW];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, acc), dst, ray * 4, zo * HW * 4, kAuxThrough);
      }
      // keep the unrolled per-voxel chains from being interleaved across voxels (it would spill the columns)
      if ((z & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  zcol_bwd_epilogue(ds_acc, ds_part, dsmall, cg_count, bk, b, P.B, wo ? bk.y * la.K : b, wo ? la.K : 1);
}

__global__ __launch_bounds__(kColThreads) void k_zcol_bwd_dyn(DpcParams P, RayHost rh, const float* __restrict__ Tin,
                                                              const float* __restrict__ s,
                                                              const float* __restrict__ dproj, const float* __restrict__ proj,
                                                              const float* __restrict__ trans_in, TapsDyn taps_arg,
                                                              TapsDyn taps_adj_arg,
                                                              float* __restrict__ dT, float* __restrict__ ds_part,
                                                              float* __restrict__ dsmall, unsigned int* __restrict__ cg_count,
                                                              const float* __restrict__ dgrid_extra, LossArgs la) {
  const TapsDyn taps = resolve_taps_dyn(taps_arg, P.dev_taps_z, false), taps_adj = resolve_taps_dyn(taps_adj_arg, P.dev_taps_z, true);
  const int HW = P.H * P.W, D = P.D;
  const bool wo = winners_only(la);  // grid over samples: this workgroup works on the winning candidate of sample bk.y
  const Blk bk = block_coords(wo ? P.B / la.K : P.B);
  const int b = wo ? bk.y * la.K + la.winner[bk.y] : bk.y, ray = bk.x * kColThreads + threadIdx.x;
  float ds_acc = 0.f;
  if (ray < HW && !cloud_loses(la, b)) {
    const RayConst rc = ray_const(rh, s, b);
    const float* col = Tin + (size_t)b * D * HW + ray;
    const int R = taps.n > 0 ? (taps.n - 1) / 2 : 0;
    auto v2_at = [&](int z) -> float {  // forward D-pass at depth z
      if (taps.n == 0) return col[(size_t)z * HW];
      float v2 = 0.f;
      for (int k = 0; k < taps.n; ++k) {
        const int zz = z + k - R;
        if (zz >= 0 && zz < D) v2 = fmaf(taps.w[k], col[(size_t)zz * HW], v2);
      }
      return v2;
    };
    float Tf;
    if (trans_in != nullptr) {
      Tf = trans_in[(size_t)b * HW + ray];
    } else {
      double trans = 1.0;
      for (int z = 0; z < D; ++z) trans *= 1.0 - (double)drc_clamp(rc, occupancy(rc, v2_at(z)));
      Tf = (float)trans;
    }
    const float g = ray_grad(P, la, dproj, proj, b, ray);
    float* out = dT + (size_t)b * D * HW + ray;
    const float* extra = dgrid_extra ? dgrid_extra + (size_t)b * D * HW + ray : nullptr;
    for (int z = 0; z < D; ++z) {
      float term;
      const float own = drc_voxel_bwd(rc, v2_at(z), g, Tf, z == 0, term);
      ds_acc += term;
      float acc;
      if (taps_adj.n == 0) {
        acc = own;
      } else {
        acc = 0.f;
        for (int k = 0; k < taps_adj.n; ++k) {
          const int zz = z + k - R;
          if (zz >= 0 && zz < D) {
            float unused;
            acc = fmaf(taps_adj.w[k], drc_voxel_bwd(rc, v2_at(zz), g, Tf, zz == 0, unused), acc);
          }
        }
      }
      if (extra != nullptr) acc += extra[(size_t)z * HW];
      out[(size_t)z * HW] = acc;
    }
  }
  zcol_bwd_epilogue(ds_acc, ds_part, dsmall, cg_count, bk, b, P.B, wo ? bk.y * la.K : b, wo ? la.K : 1);
}

}  // namespace

int launch_zcol_fwdbwd(const DpcParams* p, const float* host_kern_z, const TapPlan& pz, const float* Tbuf, const float* s,
                       float* proj, float* dT, float* ds_part, int ntile, unsigned long long* tickets, float* bwd_dsmall,
                       unsigned int* cg_count, const LossArgs& la, hipStream_t st) {
  int rc = DPC_OK;
  const RayHost rh = ray_host(p);
  constexpr int kRpl = DPC_ZFB_RPL;
  dim3 gpair(((p->H * p->W / kRpl + kColThreads - 1) / kColThreads) * p->B);
  // field layouts of the per-cloud words (contributors: ray tiles) and of the batch word (contributors: clouds)
  const SseFormat cf = sse_format(gpair.x / p->B, kTileSseCap), bf = sse_format(p->B, (double)kTileSseCap * (gpair.x / p->B));
#define DPC_ZFB(RB)                                                                                                \
  {                                                                                                                \
    const TapsT<RB> tzf = make_taps<RB>(host_kern_z, pz, false), tza = make_taps<RB>(host_kern_z, pz, true);       \
    if (p->D == 32) DPC_LAUNCH("k_zcol_fwdbwd", (k_zcol_fwdbwd<32, RB, kRpl>), gpair, dim3(kColThreads), 0, st, *p, rh, Tbuf, s, tzf, tza, proj, dT, ds_part, ntile, tickets, cf, bf, bwd_dsmall, cg_count, la); \
    else if (p->D == 64) DPC_LAUNCH("k_zcol_fwdbwd", (k_zcol_fwdbwd<64, RB, kRpl>), gpair, dim3(kColThreads), 0, st, *p, rh, Tbuf, s, tzf, tza, proj, dT, ds_part, ntile, tickets, cf, bf, bwd_dsmall, cg_count, la); \
    else DPC_LAUNCH("k_zcol_fwdbwd", (k_zcol_fwdbwd<128, RB, kRpl>), gpair, dim3(kColThreads), 0, st, *p, rh, Tbuf, s, tzf, tza, proj, dT, ds_part, ntile, tickets, cf, bf, bwd_dsmall, cg_count, la); \
  }
  DPC_FOR_BUCKET(pz.bucket, DPC_ZFB)
#undef DPC_ZFB
  return rc != DPC_OK ? rc : launch_ok();
}

int launch_zcol_fwd(const DpcParams* p, const float* host_kern_z, const TapPlan& pz, const float* Tbuf, const float* s,
                    float* smoothed, float* proj, float* trans, const LossArgs& la, hipStream_t st) {
  int rc = DPC_OK;
  const RayHost rh = ray_host(p);
  dim3 gcol(col_tiles(p) * p->B);
  bool done = false;
#define DPC_ZFWD(RB)                                                                                             \
  {                                                                                                              \
    const TapsT<RB> tz = make_taps<RB>(host_kern_z, pz, false);                                                  \
    if (p->D == 32) { DPC_LAUNCH("k_zcol_fwd", (k_zcol_fwd<32, RB>), gcol, dim3(kColThreads), 0, st, *p, rh, Tbuf, s, tz, smoothed, proj, trans, la); done = true; } \
    else if (p->D == 64) { DPC_LAUNCH("k_zcol_fwd", (k_zcol_fwd<64, RB>), gcol, dim3(kColThreads), 0, st, *p, rh, Tbuf, s, tz, smoothed, proj, trans, la); done = true; } \
    else if (p->D == 128) { DPC_LAUNCH("k_zcol_fwd", (k_zcol_fwd<128, RB>), gcol, dim3(kColThreads), 0, st, *p, rh, Tbuf, s, tz, smoothed, proj, trans, la); done = true; } \
  }
  if (pz.bucket >= 0) { DPC_FOR_BUCKET(pz.bucket, DPC_ZFWD) }
#undef DPC_ZFWD
  if (rc != DPC_OK) return rc;
  if (!done) {  // other depths / longer kernels: same arithmetic, column re-read from global
    DPC_LAUNCH("k_zcol_fwd", k_zcol_fwd_dyn, gcol, dim3(kColThreads), 0, st, *p, rh, Tbuf, s,
               make_taps_dyn(host_kern_z, p->taps_z, false), smoothed, proj, trans, la);
  }
  return launch_ok();
}

int launch_zcol_bwd(const DpcParams* p, const float* host_kern_z, const TapPlan& pz, const float* grid_wh, const float* s,
                    const float* dproj, const float* proj, const float* trans, float* dT, float* ds_part, float* dsmall,
                    unsigned int* cg_count, const float* dgrid_extra, const LossArgs& la, hipStream_t st) {
  int rc = DPC_OK;
  const RayHost rh = ray_host(p);
  dim3 gcol(col_tiles(p) * (winners_only(la) ? p->B / la.K : p->B));
  bool done = false;
#define DPC_ZBWD(RB)                                                                                              \
  {                                                                                                               \
    const TapsT<RB> tz = make_taps<RB>(host_kern_z, pz, true), tzf = make_taps<RB>(host_kern_z, pz, false);       \
    if (p->D == 32) { DPC_LAUNCH("k_zcol_bwd", (k_zcol_bwd<32, RB>), gcol, dim3(kColThreads), 0, st, *p, rh, grid_wh, s, dproj, proj, trans, tzf, tz, dT, ds_part, dsmall, cg_count, dgrid_extra, la); done = true; } \
    else if (p->D == 64) { DPC_LAUNCH("k_zcol_bwd", (k_zcol_bwd<64, RB>), gcol, dim3(kColThreads), 0, st, *p, rh, grid_wh, s, dproj, proj, trans, tzf, tz, dT, ds_part, dsmall, cg_count, dgrid_extra, la); done = true; } \
    else if (p->D == 128) { DPC_LAUNCH("k_zcol_bwd", (k_zcol_bwd<128, RB>), gcol, dim3(kColThreads), 0, st, *p, rh, grid_wh, s, dproj, proj, trans, tzf, tz, dT, ds_part, dsmall, cg_count, dgrid_extra, la); done = true; } \
  }
  if (pz.bucket >= 0) { DPC_FOR_BUCKET(pz.bucket, DPC_ZBWD) }
#undef DPC_ZBWD
  if (rc != DPC_OK) return rc;
  if (!done) {
    DPC_LAUNCH("k_zcol_bwd", k_zcol_bwd_dyn, gcol, dim3(kColThreads), 0, st, *p, rh, grid_wh, s, dproj, proj, trans,
               make_taps_dyn(host_kern_z, p->taps_z, false), make_taps_dyn(host_kern_z, p->taps_z, true), dT, ds_part,
               dsmall, cg_count, dgrid_extra, la);
  }
  return launch_ok();
}

int launch_loss_finalize(const float* sse_tiles, int ntile, float* sse, int S, int K, float inv_S, float* loss, int32_t* winner,
                         hipStream_t st) {
  DPC_LAUNCH("k_loss_finalize", k_loss_finalize, dim3(1), dim3(256), 0, st, sse_tiles, ntile, sse, S, K, inv_S, loss, winner);
  return launch_ok();
}

}  // namespace dpck

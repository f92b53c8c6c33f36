// Fused hot path of the differentiable point-cloud projection for MI355X (gfx950).
//
// Replaces pointcloud_project_fast (dpc/util/point_cloud_to.py:191-263 of the reference) and its autograd
// backward with five launches:
//
//   forward   k_locate     one lane per point: the reference's exact transform sequence + fp64 cell location,
//                          then each 256-point block counting-sorts its records by z cell in LDS, so the
//                          slab kernels read only the points that touch their slab.
//             k_splat_hw   one workgroup per (cloud, z-slab): scatter the 8 trilinear corners of the slab's
//                          points into an LDS-resident slab with ds_add_f32, emit the clamp mask
//                          (1 bit/voxel), clamp, run the W- and H-passes of the Gaussian in LDS and store the
//                          slab once, coalesced.
//             k_zcol_fwd   one lane per (cloud, y, x) ray: the whole z column in registers; D-pass of the
//                          Gaussian, occupancy scale + clamp, DRC transmittance product (fp64 register),
//                          silhouette written with the row flip folded into the store.
//   backward  k_zcol_bwd   one lane per ray: DRC backward in closed form, scale/clamp backward (+ ds
//                          partial sums), adjoint D-pass.
//             k_gather_hw  one workgroup per (cloud, z-slab + 1 halo plane): adjoint H-/W-passes in LDS,
//                          clamp mask, trilinear gather of the 8 corners straight from LDS, transform
//                          backward, wave/block reduction of the quaternion/translation/focal gradients.
//
// The slab kernels exist twice: a generic form (any H x W, runtime strides, bounds-checked windows) and a form
// specialised at compile time for H = W = 32/64/128 (rows padded to 16-byte multiples with zero pads instead
// of bounds checks, ds_read_b128 W-pass, packed-FMA H-pass over column pairs).
//
// HBM traffic per cloud: 16 G^3 B forward + 16 G^3 (+1 halo plane per slab) backward + O(N); the raw splat
// grid never leaves LDS (only its 1-bit clamp mask does).
#include <math.h>
#include <string.h>

#include "dpc_common.h"
#include "dpc_profile.h"

#ifdef DPC_ABLATE
__device__ int g_dpc_ablate = 0;  // diagnostic builds only: bit0 skip zero-fill, bit1 skip atomics, bit2 skip W, bit3 skip H
extern "C" int dpc_debug_set_ablate(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_dpc_ablate), &v, sizeof(int)) == hipSuccess ? 0 : -5; }
#define DPC_ABL(bit) (g_dpc_ablate & (1 << (bit)))
// per-phase timestamps (100 MHz s_memrealtime, comparable across CUs), thread 0 of every workgroup; 16 slots per block
__device__ unsigned long long* g_dpc_stamps = nullptr;
extern "C" int dpc_debug_set_stamps(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_dpc_stamps), &p, sizeof(void*)) == hipSuccess ? 0 : -5; }
#define DPC_STAMP(slot)                                                                                    \
  do {                                                                                                     \
    if (g_dpc_stamps != nullptr && threadIdx.x == 0)                                                       \
      g_dpc_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define DPC_ABL(bit) 0
#define DPC_STAMP(slot) do { } while (0)
#endif

namespace {

constexpr int kSlabThreads = 1024;
constexpr int kColThreads = 256;
constexpr int kNumCUs = 256;  // MI355X
// k_zcol_fwdbwd's per-cloud word: [63:51] blocks arrived, [50:0] squared error, 30 fractional bits (a cloud's sum is
// at most H*W <= 2^20)
constexpr int kSseCountShift = 51, kSseFrac = 30;
#ifndef DPC_ZFB_RPL
#define DPC_ZFB_RPL 1  // rays per lane in k_zcol_fwdbwd (1 or 2)
#endif
constexpr int kLocThreads = 256;                     // points per locate block == points per sorted chunk
constexpr int kL = 16;                               // outputs per thread in the generic in-LDS line convolutions
constexpr int kLdsLimit = 160 * 1024;                // bytes of LDS a workgroup may use on gfx950
constexpr int kRedTab = 256;     // float offset of the record table inside the backward kernel's scratch tail
constexpr int kRedMask = 400;    // float offset of the staged clamp-mask words
constexpr int kRedFloats = 1024; // generic kernels: reduction scratch (13 x 16 floats) + record table
constexpr int kLdsBudget = kLdsLimit - 4096;         // generic slab bytes; the rest holds the reduction scratch

__device__ inline int odd_stride(int w) { return w | 1; }  // generic LDS row stride: odd => conflict-free column walks

// ------------------------------------------------------------------------------------------------------
// Binned point storage ("cells"): per cloud, ceil(N/256) chunks; chunk c holds the records of points
// [256c, 256c+256) counting-sorted by bin (bin = z cell iz, or D for out-of-bounds points):
//   [256 x PointRec (16 B)] [256 x {px, py, pz, original index} (16 B)] [(D+2) x uint16 bin start offsets, padded to 16 B]
// offs[k] = first sorted position of bin k; offs[D+1] = number of points in the chunk.
// ------------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t chunk_bytes(int D) {
  return (size_t)kLocThreads * 2 * sizeof(PointRec) + (((size_t)(D + 2) * 2 + 15) / 16) * 16;
}
__host__ __device__ inline int num_chunks(int N) { return (N + kLocThreads - 1) / kLocThreads; }

struct Cells {
  const uint8_t* base;
  size_t chunk;   // bytes per chunk
  int nblk;       // chunks per cloud
  __device__ const uint8_t* at(int b, int blk) const { return base + ((size_t)b * nblk + blk) * chunk; }
  __device__ const PointRec* recs(int b, int blk) const { return reinterpret_cast<const PointRec*>(at(b, blk)); }
  // the point itself and its original index, sorted like the records (the backward reads them sequentially)
  __device__ const int4* aux(int b, int blk) const {
    return reinterpret_cast<const int4*>(at(b, blk) + (size_t)kLocThreads * sizeof(PointRec));
  }
  __device__ const uint16_t* offs(int b, int blk) const {
    return reinterpret_cast<const uint16_t*>(at(b, blk) + (size_t)kLocThreads * 2 * sizeof(PointRec));
  }
};

// Visit every record of cloud b whose bin lies in [bin_lo, bin_hi).  f(rec, aux) with aux -> {px,py,pz,orig}.
//   build_record_table (wave 0, before a barrier the caller already has): lane c reads chunk c's range, an
//   inclusive scan over lanes gives every chunk's first flat index; tab = {prefix[nblk+1], begin[nblk]} in LDS.
//   for_each_record_flat: threads take flat indices tid, tid+nthr, ... and find their chunk by binary search in
//   the table -- balanced over the whole workgroup and only two dependent global reads deep (offsets, record).
// Needs nblk <= 64 (N <= 16384); larger clouds use the wave-per-chunk loop below.
constexpr int kTabInts = 2 * DPC_WAVE + 2;

struct RecordRange {  // wave 0, lane c: sorted range [beg, beg+cnt) of chunk c
  int beg, cnt;
};

// Issue the offset loads early (they are only waited for in finish_record_table, so a whole phase can run under them).
__device__ inline RecordRange load_record_range(const Cells& cells, int b, int bin_lo, int bin_hi) {
  RecordRange r{0, 0};
  const int c = threadIdx.x;
  if (c < DPC_WAVE && c < cells.nblk) {
    const uint16_t* offs = cells.offs(b, c);
    r.beg = offs[bin_lo];
    r.cnt = (int)offs[bin_hi] - r.beg;
  }
  return r;
}

__device__ inline void finish_record_table(const RecordRange& r, int* tab) {
  if (threadIdx.x >= DPC_WAVE) return;
  const int c = threadIdx.x;
  int incl = r.cnt;
#pragma unroll
  for (int off = 1; off < DPC_WAVE; off <<= 1) {
    const int up = __shfl_up(incl, off, DPC_WAVE);
    if (c >= off) incl += up;
  }
  tab[c] = incl - r.cnt;               // exclusive prefix
  tab[DPC_WAVE + 1 + c] = r.beg;
  if (c == DPC_WAVE - 1) tab[DPC_WAVE] = incl;  // total
}

// flat index j -> (chunk, sorted position): largest c with tab[c] <= j (prefix non-decreasing; empty chunks repeat)
__device__ inline void flat_lookup(const int* tab, int j, int& chunk, int& pos) {
  int lo = 0, hi = DPC_WAVE;
#pragma unroll
  for (int step = 0; step < 6; ++step) {
    const int mid = (lo + hi) >> 1;
    if (tab[mid] <= j) lo = mid; else hi = mid;
  }
  chunk = lo;
  pos = tab[DPC_WAVE + 1 + lo] + (j - tab[lo]);
}

template <class F>
__device__ inline void for_each_record_flat(const Cells& cells, int b, const int* tab, F f, int first = 0) {
  const int total = tab[DPC_WAVE];
  for (int j = first + threadIdx.x; j < total; j += blockDim.x) {
    int c, pos;
    flat_lookup(tab, j, c, pos);
    f(load_record(cells.recs(b, c), pos), cells.aux(b, c) + pos);
  }
}

template <class F>
__device__ inline void for_each_record(const Cells& cells, int b, int bin_lo, int bin_hi, F f) {
  const int lane = threadIdx.x & (DPC_WAVE - 1), wave = threadIdx.x / DPC_WAVE, nw = blockDim.x / DPC_WAVE;
  for (int blk = wave; blk < cells.nblk; blk += nw) {
    const uint16_t* offs = cells.offs(b, blk);
    const int beg = __builtin_amdgcn_readfirstlane((int)offs[bin_lo]);
    const int end = __builtin_amdgcn_readfirstlane((int)offs[bin_hi]);
    const PointRec* recs = cells.recs(b, blk);
    const int4* aux = cells.aux(b, blk);
    for (int j = beg + lane; j < end; j += DPC_WAVE) f(load_record(recs, j), aux + j);
  }
}

// ------------------------------------------------------------------------------------------------------
// Forward 0: per-point transform (reference-exact op sequence) + fp64 cell location + per-block z-sort.
//   grid (ceil(N/256), B), 256 threads.
//   SRC = 0: pc/q/t/f -> camera transform (pc_perspective_transform), optional tr_pc output
//   SRC = 1: points are already transformed, fp32 (z,y,x);  SRC = 2: same, fp64
// ------------------------------------------------------------------------------------------------------
template <int SRC>
__global__ __launch_bounds__(kLocThreads) void k_locate(DpcParams P, const void* __restrict__ pts,
                                                        const float* __restrict__ q, const float* __restrict__ t,
                                                        const float* __restrict__ f, float* __restrict__ tr_pc,
                                                        uint8_t* __restrict__ cells_out) {
  __shared__ int hist[1026];  // D + 2 <= 1026 bins (validate() caps D at 1024)
  const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int D = P.D, nbins = D + 1;
  const int i = blk * kLocThreads + tid;
  const bool live = i < P.N;
  for (int k = tid; k < nbins + 1; k += kLocThreads) hist[k] = 0;
  // every thread normalises the quaternion itself (a dozen fp32 ops): cheaper than one thread doing it while 255 wait
  CameraRef cam_s;
  if (SRC == 0) cam_s = load_camera_ref(P, q, t, f, b);

  PointRec rec;
  rec.code = -1; rec.tz = rec.ty = rec.tx = 0.f;
  float src_pt[3] = {0.f, 0.f, 0.f};  // the untransformed point (SRC 0), carried next to its record for the backward
  if (live) {
    const size_t idx = (size_t)b * P.N + i;
    double Z, Y, X;
    if (SRC == 0) {
      const int reps = P.point_replicas > 1 ? P.point_replicas : 1;  // replicas of one point set read the same rows
      const float* p = static_cast<const float*>(pts) + ((size_t)(b / reps) * P.N + i) * 3;
      src_pt[0] = p[0]; src_pt[1] = p[1]; src_pt[2] = p[2];
      project_point_ref(cam_s, p[0], p[1], p[2], Z, Y, X);
      if (tr_pc != nullptr) {
        tr_pc[idx * 3 + 0] = (float)Z; tr_pc[idx * 3 + 1] = (float)Y; tr_pc[idx * 3 + 2] = (float)X;
      }
    } else if (SRC == 1) {
      const float* p = static_cast<const float*>(pts) + idx * 3;
      Z = p[0]; Y = p[1]; X = p[2];
    } else {
      const double* p = static_cast<const double*>(pts) + idx * 3;
      Z = p[0]; Y = p[1]; X = p[2];
    }
    rec = make_record(Z, Y, X, P.D, P.H, P.W);
  }
  const int bin = rec.code < 0 ? D : (rec.code >> 20);
  __syncthreads();  // hist is zeroed (the transform above ran under that latency)
  int rank = 0;
  if (live) rank = atomicAdd(&hist[bin], 1);  // ds_add_rtn_u32: position inside the bin
  __syncthreads();

  // exclusive prefix over the bins by the first wave: lane l owns bins [l*C, (l+1)*C)
  if (tid < DPC_WAVE) {
    const int C = (nbins + DPC_WAVE - 1) / DPC_WAVE;
    int sum = 0;
    for (int k = tid * C; k < min((tid + 1) * C, nbins); ++k) sum += hist[k];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < DPC_WAVE; off <<= 1) {
      const int up = __shfl_up(incl, off, DPC_WAVE);
      if (tid >= off) incl += up;
    }
    int run = incl - sum;
    for (int k = tid * C; k < min((tid + 1) * C, nbins); ++k) {
      const int c = hist[k];
      hist[k] = run;
      run += c;
    }
    if (tid == DPC_WAVE - 1) hist[nbins] = incl;  // total
  }
  __syncthreads();

  // sorted chunk staged in LDS, then copied out with one coalesced 16-byte store per lane and array
  __shared__ int4 stage[2 * kLocThreads];
  if (live) {
    const int pos = hist[bin] + rank;
    int4 v;
    v.x = rec.code; v.y = __float_as_int(rec.tz); v.z = __float_as_int(rec.ty); v.w = __float_as_int(rec.tx);
    stage[pos] = v;
    int4 a;
    a.x = __float_as_int(src_pt[0]); a.y = __float_as_int(src_pt[1]); a.z = __float_as_int(src_pt[2]); a.w = i;
    stage[kLocThreads + pos] = a;
  }
  __syncthreads();
  const size_t chunk = chunk_bytes(D);
  uint8_t* out = cells_out + ((size_t)b * gridDim.x + blk) * chunk;
  const int npts = min(kLocThreads, P.N - blk * kLocThreads);
  if (tid < npts) {
    reinterpret_cast<int4*>(out)[tid] = stage[tid];
    reinterpret_cast<int4*>(out + (size_t)kLocThreads * sizeof(PointRec))[tid] = stage[kLocThreads + tid];
  }
  uint16_t* offs = reinterpret_cast<uint16_t*>(out + (size_t)kLocThreads * 2 * sizeof(PointRec));
  for (int k = tid; k < nbins + 1; k += kLocThreads) offs[k] = (uint16_t)hist[k];
}

// ------------------------------------------------------------------------------------------------------
// Generic in-LDS separable passes over a slab laid out [nz][H][WP] (runtime dims, bounds-checked windows).
// Every thread owns (line, segment-of-kL-outputs); all windows are read, then a barrier, then written back,
// so the pass is in place.  Lanes map to consecutive lines (W-pass: stride WP odd; H-pass: consecutive x),
// which keeps ds_read_b32/ds_write_b32 bank-conflict free.
// ------------------------------------------------------------------------------------------------------
template <int RB, bool CLAMP1, class Post>
__device__ inline void wpass_inplace(float* slab, int nz, int H, int W, int WP, const TapsT<RB>& taps, Post post) {
  const int nseg = (W + kL - 1) / kL;
  const int lines = nz * H;
  const int per_round = blockDim.x / nseg;
  for (int l0 = 0; l0 < lines; l0 += per_round) {
    const int li = threadIdx.x % per_round, seg = threadIdx.x / per_round;
    const int line = l0 + li;
    const bool act = seg < nseg && line < lines;
    float v[kL + 2 * RB];
    if (act) window_load<RB, kL, CLAMP1>(v, slab + line * WP, 1, seg * kL, W);
    __syncthreads();
    if (act) {
#pragma unroll
      for (int j = 0; j < kL; ++j) {
        const int x = seg * kL + j;
        if (x < W) slab[line * WP + x] = post(line, x, window_dot<RB, kL>(v, taps, j));
      }
    }
    __syncthreads();
  }
}

template <int RB, class Store>
__device__ inline void hpass(float* slab, int nz, int H, int W, int WP, const TapsT<RB>& taps, Store store) {
  const int nseg = (H + kL - 1) / kL;
  const int lines = nz * W;
  const int per_round = blockDim.x / nseg;
  for (int l0 = 0; l0 < lines; l0 += per_round) {
    const int li = threadIdx.x % per_round, seg = threadIdx.x / per_round;
    const int line = l0 + li;
    const bool act = seg < nseg && line < lines;
    const int z = line / W, x = line - z * W;
    float v[kL + 2 * RB];
    if (act) window_load<RB, kL, false>(v, slab + z * H * WP + x, WP, seg * kL, H);
    __syncthreads();
    if (act) {
#pragma unroll
      for (int j = 0; j < kL; ++j) {
        const int y = seg * kL + j;
        if (y < H) store(z, y, x, window_dot<RB, kL>(v, taps, j));
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------
// Specialised slab geometry: H = W = GS known at compile time.
//   row layout   [PAD zeros][GS values], PAD = max(4, RB rounded up to 4): rows start 16-byte aligned, the
//                zero pad of row r+1 doubles as the right halo of row r, so W-windows need no bounds checks
//   W-pass       thread = (row, 32-output segment); lanes walk consecutive rows (stride GS+PAD floats keeps
//                ds_read_b128 / ds_write_b128 conflict-free for PAD = 4)
//   H-pass       thread = (plane, column PAIR, 16-output segment); ds_read_b64 of two adjacent columns and
//                packed v_pk_fma_f32 on the pair
// ------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int GS, int RB, int NT_, int LW_, int LH_>
struct SlabGeo {
  static constexpr int PAD = RB <= 4 ? 4 : ((RB + 3) / 4) * 4;
  static constexpr int WP = GS + PAD;
  // row stride of the forward's 64-bit accumulators (u64 units).  GS + PAD is a multiple of 4, i.e. 8 mod 16 dwords:
  // lanes that walk rows with ds_read_b128 then use only every other group of four banks (2-way conflict in every
  // 16-lane group; SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE).  Two spare (always zero) entries per row make
  // the stride 4 or 12 mod 16 dwords: sixteen consecutive rows land on sixteen distinct bank groups.
  static constexpr int WPA = GS + PAD + 2;
  static constexpr int PLANE = GS * WP;
  static constexpr int LW = LW_, NSEGW = GS / LW, LWIN = LW + 2 * PAD;              // W-pass
  static constexpr int LH = LH_, NSEGH = GS / LH, XP = GS / 2, HWIN = LH + 2 * RB;  // H-pass
  static constexpr int NT = NT_;
  __host__ __device__ static constexpr size_t slab_floats(int planes) { return (size_t)planes * PLANE + PAD; }
  __device__ static int at(int z, int y, int x) { return (z * GS + y) * WP + PAD + x; }
};
// forward: ZS planes, 16 voxels per thread, short segments so that every thread owns exactly one W and one H item
template <int GS, int ZS, int RB>
using FwdGeo = SlabGeo<GS, RB, ZS * GS * GS / 16, 16, 8>;
// backward: NPL = ZS+1 planes (halo), long segments; one item per thread up to 1024 threads
constexpr int bwd_threads(int gs, int npl) {
  const int items = npl * gs * (gs / 32);
  return items >= 1024 ? 1024 : (items <= 256 ? 256 : ((items + 63) / 64) * 64);
}
template <int GS, int RB, int NPL>
using BwdGeo = SlabGeo<GS, RB, bwd_threads(GS, NPL), 32, 16>;

constexpr float kFixScale = 17592186044416.0f;          // 2^44: splat weights accumulate as 64-bit fixed point
constexpr float kFixInv = 1.0f / 17592186044416.0f;
constexpr unsigned long long kFixOne = 1ull << 44;

// In-place W-pass over NPL planes.  MASK: 0 none, 1 emit the clamp mask from the raw values (forward),
// 2 multiply the outputs by the stored mask bits (backward).  mask32 points at this slab's first plane.
template <class Geo, int GS, int NPL>
__host__ __device__ constexpr int wpass_items_per_thread() {
  return (NPL * GS * Geo::NSEGW + Geo::NT - 1) / Geo::NT;
}

// Request the clamp-mask words of this thread's W-pass items ahead of time (MASK = 3 below consumes them).
template <class Geo, int GS, int NPL>
__device__ inline void wpass_mask_prefetch(const uint32_t* __restrict__ mask32, int planes_present,
                                           uint32_t (&bits)[wpass_items_per_thread<Geo, GS, NPL>()]) {
  constexpr int ROWS = NPL * GS, ITEMS = ROWS * Geo::NSEGW, IPT = wpass_items_per_thread<Geo, GS, NPL>();
#pragma unroll
  for (int it = 0; it < IPT; ++it) {
    const int item = threadIdx.x + it * Geo::NT;
    bits[it] = 0u;
    if (item < ITEMS) {
      const int row = item % ROWS, seg = item / ROWS;
      if (row / GS < planes_present) bits[it] = mask32[(size_t)row * Geo::NSEGW + seg];
    }
  }
}

template <class Geo, int GS, int RB, int NPL, bool CLAMP1, int MASK>
__device__ inline void wpass_fast(float* slab, const TapsT<RB>& taps, const uint32_t* mask32, int planes_present) {
  constexpr int ROWS = NPL * GS, ITEMS = ROWS * Geo::NSEGW, IPT = (ITEMS + Geo::NT - 1) / Geo::NT;
  float v[IPT][Geo::LWIN];
#pragma unroll
  for (int it = 0; it < IPT; ++it) {
    const int item = threadIdx.x + it * Geo::NT;
    if (item < ITEMS) {
      const int row = item % ROWS, seg = item / ROWS;
      const f32x4* src = reinterpret_cast<const f32x4*>(slab + row * Geo::WP + seg * Geo::LW);
#pragma unroll
      for (int k = 0; k < Geo::LWIN / 4; ++k) {
        const f32x4 q = src[k];
        v[it][4 * k + 0] = q.x; v[it][4 * k + 1] = q.y; v[it][4 * k + 2] = q.z; v[it][4 * k + 3] = q.w;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < IPT; ++it) {
    const int item = threadIdx.x + it * Geo::NT;
    if (item < ITEMS) {
      const int row = item % ROWS, seg = item / ROWS;
      const int z = row / GS;
      uint32_t bits = 0xffffffffu;
      static_assert(MASK != 1, "the forward emits its mask from the fixed-point accumulators");
      static_assert(MASK == 0 || Geo::LW == 32, "mask word addressing assumes 32-output segments");
      if (MASK == 2) bits = z < planes_present ? mask32[(size_t)row * Geo::NSEGW + seg] : 0u;  // bits of the row's mask word
      if (MASK == 3) bits = mask32[it];  // prefetched by wpass_mask_prefetch (registers)
      if (CLAMP1) {
#pragma unroll
        for (int k = 0; k < Geo::LWIN; ++k) v[it][k] = fminf(v[it][k], 1.0f);
      }
      f32x4* dst = reinterpret_cast<f32x4*>(slab + row * Geo::WP + Geo::PAD + seg * Geo::LW);
#pragma unroll
      for (int k = 0; k < Geo::LW / 4; ++k) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = 4 * k + e;
          float acc = 0.f;
#pragma unroll
          for (int tp = 0; tp < 2 * RB + 1; ++tp) acc = fmaf(taps.w[tp], v[it][j + tp + Geo::PAD - RB], acc);
          o[e] = (MASK >= 2 && !((bits >> j) & 1u)) ? 0.f : acc;
        }
        f32x4 q;
        q.x = o[0]; q.y = o[1]; q.z = o[2]; q.w = o[3];
        dst[k] = q;
      }
    }
  }
  __syncthreads();
}

// H-pass over NPL planes, two adjacent columns per thread.  store(z, y, x_even, pair) consumes the results;
// INPLACE inserts the barrier between the window reads and the stores.
template <class Geo, int GS, int RB, int NPL, bool INPLACE, class Store>
__device__ inline void hpass_fast(const float* slab, const TapsT<RB>& taps, Store store) {
  constexpr int ITEMS = NPL * Geo::XP * Geo::NSEGH, IPT = (ITEMS + Geo::NT - 1) / Geo::NT;
  f32x2 v[IPT][Geo::HWIN];
#pragma unroll
  for (int it = 0; it < IPT; ++it) {
    const int item = threadIdx.x + it * Geo::NT;
    if (item < ITEMS) {
      const int xp = item % Geo::XP, rest = item / Geo::XP;
      const int z = rest % NPL, seg = rest / NPL;
      const float* col = slab + Geo::at(z, 0, 2 * xp);
      const int y0 = seg * Geo::LH - RB;
#pragma unroll
      for (int i = 0; i < Geo::HWIN; ++i) {
        // only the first/last RB rows of a window can fall outside the plane (zero padding): clamp the
        // address, then zero the value
        const bool out = (i < RB && y0 + i < 0) || (i >= Geo::LH + RB && y0 + i >= GS);
        const int y = out ? 0 : y0 + i;
        f32x2 q = *reinterpret_cast<const f32x2*>(col + y * Geo::WP);
        if (out) q = f32x2{0.f, 0.f};
        v[it][i] = q;
      }
    }
  }
  if (INPLACE) __syncthreads();
#pragma unroll
  for (int it = 0; it < IPT; ++it) {
    const int item = threadIdx.x + it * Geo::NT;
    if (item < ITEMS) {
      const int xp = item % Geo::XP, rest = item / Geo::XP;
      const int z = rest % NPL, seg = rest / NPL;
#pragma unroll
      for (int j = 0; j < Geo::LH; ++j) {
        f32x2 acc = f32x2{0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < 2 * RB + 1; ++tp)
          acc = __builtin_elementwise_fma(f32x2{taps.w[tp], taps.w[tp]}, v[it][j + tp], acc);
        store(z, seg * Geo::LH + j, 2 * xp, acc);
      }
    }
  }
  if (INPLACE) __syncthreads();
}

// The same H-pass with the input planes in GLOBAL memory ([planes][GS][GS], dense): used by the backward slab
// kernel so that dT is read once, convolved in registers and written to LDS once.  Planes >= planes_present and rows
// outside the plane read as zero.
template <class Geo, int GS, int RB, int NPL, class Store>
__device__ inline void hpass_global(const float* __restrict__ src, int planes_present, const TapsT<RB>& taps, Store store) {
  constexpr int ITEMS = NPL * Geo::XP * Geo::NSEGH, IPT = (ITEMS + Geo::NT - 1) / Geo::NT;
#pragma unroll
  for (int it = 0; it < IPT; ++it) {
    const int item = threadIdx.x + it * Geo::NT;
    if (item < ITEMS) {
      const int xp = item % Geo::XP, rest = item / Geo::XP;
      const int z = rest % NPL, seg = rest / NPL;
      const float* col = src + (size_t)z * GS * GS + 2 * xp;
      const int y0 = seg * Geo::LH - RB;
      const bool have = z < planes_present;
      f32x2 v[Geo::HWIN];
#pragma unroll
      for (int i = 0; i < Geo::HWIN; ++i) {
        const int y = y0 + i;
        const bool in = have && !((i < RB && y < 0) || (i >= Geo::LH + RB && y >= GS));
        v[i] = in ? *reinterpret_cast<const f32x2*>(col + (size_t)y * GS) : f32x2{0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < Geo::LH; ++j) {
        f32x2 acc = f32x2{0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < 2 * RB + 1; ++tp)
          acc = __builtin_elementwise_fma(f32x2{taps.w[tp], taps.w[tp]}, v[j + tp], acc);
        store(z, seg * Geo::LH + j, 2 * xp, acc);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Forward 1: splat + mask + clamp + W/H Gaussian passes.                          grid (nslab, B)
//   GS = 0: generic (runtime dims, Zs = zs_rt);  GS > 0: specialised, ZS planes per slab.
//   Tbuf == nullptr: stage-level pointcloud2voxels3d_fast, only `raw` is written.
// ------------------------------------------------------------------------------------------------------
template <int GS, int ZS, int RB>
__global__ __launch_bounds__(kSlabThreads) void k_splat_hw(DpcParams P, Cells cells, TapsT<RB> taps, int zs_rt,
                                                           float* __restrict__ raw, float* __restrict__ Tbuf,
                                                           uint64_t* __restrict__ mask, float* __restrict__ sse,
                                                           float* __restrict__ loss_zero, int* __restrict__ winner_zero,
                                                           unsigned long long* __restrict__ ticket_zero) {
  extern __shared__ __attribute__((aligned(16))) float slab[];
  if (sse != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {  // k_zcol_fwd accumulates into these
    sse[blockIdx.y] = 0.f;
    if (winner_zero != nullptr) winner_zero[blockIdx.y] = 0;   // K == 1: sample == cloud, candidate 0 wins
    if (ticket_zero != nullptr) ticket_zero[blockIdx.y] = 0ull;  // k_zcol_fwdbwd's per-cloud sum-and-count word
    if (loss_zero != nullptr && blockIdx.y == 0) *loss_zero = 0.f;
  }
  const int D = P.D, H = P.H, W = P.W;
  const int Zs = GS ? ZS : zs_rt;
  const int b = blockIdx.y, z0 = blockIdx.x * Zs;
  const int nz = min(Zs, D - z0);
  const int tid = threadIdx.x, nthr = blockDim.x;
  const size_t HW = (size_t)H * W;

  if constexpr (GS > 0) {
    // Splat accumulation in 64-bit fixed point (2^-44): integer LDS atomics run ~9x faster than ds_add_f32 on
    // gfx950 (measured), the sums are exact to 6e-14 per contribution and independent of arrival order.
    // Accumulator rows carry the same zero pads as the fp32 slab rows ([PAD][GS] u64), so the W-pass can take its
    // windows straight from the accumulators.
    using Geo = FwdGeo<GS, ZS, RB>;
    constexpr int WPA = Geo::WPA;                            // accumulator row stride (u64), see SlabGeo
    constexpr int ACC = ZS * GS * WPA + Geo::PAD;            // u64 words incl. the tail pad
    constexpr int VOX = ZS * GS * GS, VPT = VOX / Geo::NT;
    static_assert(VOX % Geo::NT == 0 && Geo::NT % 64 == 0 && (GS * GS) % 64 == 0 && ACC % 2 == 0, "slab shape");
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(slab);
    f32x4* s4 = reinterpret_cast<f32x4*>(slab);
    int* tab = reinterpret_cast<int*>(acc + ACC);            // record table sits behind the accumulators
    const bool flat = cells.nblk <= DPC_WAVE;
    DPC_STAMP(0);
    // Two dependent global reads feed the scatter (chunk offsets, then records); each hides under one half of the
    // accumulator zero-fill: offsets | zero A | table | barrier | records -> registers | zero B | barrier | atomics.
    constexpr int PRE = 2;                 // records prefetched per thread (covers 2*NT points per slab)
    constexpr int ZH = (ACC / 2) / 2;      // float4 words in the first zero-fill half
    RecordRange rr{0, 0};
    if (flat) rr = load_record_range(cells, b, max(z0 - 1, 0), z0 + nz);
    for (int i = tid; i < ZH; i += Geo::NT) s4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (flat) finish_record_table(rr, tab);
    __syncthreads();
    PointRec pre[PRE];
    int npre = 0;
    if (flat) {
      const int total = tab[DPC_WAVE];
#pragma unroll
      for (int r = 0; r < PRE; ++r) {
        const int j = tid + r * Geo::NT;
        pre[r].code = -1; pre[r].tz = pre[r].ty = pre[r].tx = 0.f;
        if (j < total) {
          int c, pos;
          flat_lookup(tab, j, c, pos);
          pre[r] = load_record(cells.recs(b, c), pos);
        }
      }
      npre = PRE * Geo::NT;
    }
    for (int i = ZH + tid; i < ACC / 2; i += Geo::NT) s4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    DPC_STAMP(1);
    auto scatter = [&](const PointRec& rec, const int4*) {
      const Cell c = cell_from_record(rec);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int zz = c.iz + k - z0;
        if (zz < 0 || zz >= nz) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (c.iy + j >= GS) continue;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (c.ix + e >= GS) continue;
            const float w = c.wz[k] * c.wy[j] * c.wx[e];
            atomicAdd(&acc[(zz * GS + c.iy + j) * WPA + Geo::PAD + c.ix + e], (unsigned long long)(w * kFixScale));  // ds_add_u64
          }
        }
      }
    };
    if (flat) {
#pragma unroll
      for (int r = 0; r < PRE; ++r)
        if (pre[r].code >= 0) scatter(pre[r], nullptr);
      for_each_record_flat(cells, b, tab, scatter, npre);  // slabs holding more than PRE*NT points
    } else {
      for_each_record(cells, b, max(z0 - 1, 0), z0 + nz, scatter);
    }
    __syncthreads();
    DPC_STAMP(2);

    const size_t wpp = (HW + 63) / 64;
    // a < 2^56: hi < 2^24 converts exactly, lo rounds once, the fma rounds once more (<= 1 ulp overall)
    auto to_float = [](unsigned long long a) {
      return fmaf((float)(unsigned)(a >> 32), 0x1p-12f, (float)(unsigned)a * kFixInv);
    };
    if (Tbuf == nullptr || RB == 0) {
      // stage-level splat (raw grid out) or no smoothing: plain conversion, lanes <-> consecutive x;
      // clamp mask straight from the integers (raw <= 1  <=>  acc <= 2^44)
      unsigned long long* mask_out = mask ? reinterpret_cast<unsigned long long*>(mask) + ((size_t)b * D + z0) * wpp : nullptr;
      const float w2 = taps.w[0] * taps.w[0];  // centre tap of a kernel trimmed to radius 0 (1 when there is no kernel)
#pragma unroll
      for (int n = 0; n < VPT; ++n) {
        const int i = tid + n * Geo::NT;
        const unsigned long long a = acc[(i / GS) * WPA + Geo::PAD + (i % GS)];
        const unsigned long long bits = __ballot(a <= kFixOne);
        const bool present = i < nz * GS * GS;
        if (mask_out != nullptr && present && (tid & 63) == 0) mask_out[i >> 6] = bits;
        const float v = to_float(a);
        if (raw != nullptr && present) raw[((size_t)b * D + z0) * HW + i] = v;
        if (Tbuf != nullptr && present) Tbuf[((size_t)b * D + z0) * HW + i] = w2 * fminf(v, 1.0f);
      }
      return;
    } else {
      // W-pass with its windows converted on the fly from the accumulators; every thread owns one (row, segment)
      constexpr int ROWS = ZS * GS;
      static_assert(ROWS * Geo::NSEGW == Geo::NT && Geo::LW == 16, "one W item per thread, 16-bit mask pieces");
      const int row = tid % ROWS, seg = tid / ROWS;
      float v[Geo::LWIN];
      unsigned bits = 0u;
      {
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(acc + row * WPA + seg * Geo::LW);
#pragma unroll
        for (int k = 0; k < Geo::LWIN / 2; ++k) {
          const ulonglong2 q2 = src[k];
          const unsigned long long a2[2] = {q2.x, q2.y};
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int idx = 2 * k + e;
            if (idx >= Geo::PAD && idx < Geo::PAD + Geo::LW) bits |= (a2[e] <= kFixOne ? 1u : 0u) << (idx - Geo::PAD);
            v[idx] = fminf(to_float(a2[e]), 1.0f);
          }
        }
      }
      if (row / GS < nz)  // this thread's 16 voxels of the clamp mask
        reinterpret_cast<unsigned short*>(mask + ((size_t)b * D + z0) * wpp)[row * Geo::NSEGW + seg] = (unsigned short)bits;
      __syncthreads();  // every accumulator has been read: the same LDS now takes the padded fp32 slab
      DPC_STAMP(3);
      f32x4* dst = reinterpret_cast<f32x4*>(slab + row * Geo::WP + Geo::PAD + seg * Geo::LW);
#pragma unroll
      for (int k = 0; k < Geo::LW / 4; ++k) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float accv = 0.f;
#pragma unroll
          for (int tp = 0; tp < 2 * RB + 1; ++tp) accv = fmaf(taps.w[tp], v[4 * k + e + tp + Geo::PAD - RB], accv);
          o[e] = accv;
        }
        dst[k] = f32x4{o[0], o[1], o[2], o[3]};
      }
      __syncthreads();
      DPC_STAMP(4);
      float* Tout = Tbuf + ((size_t)b * D + z0) * HW;
      hpass_fast<Geo, GS, RB, ZS, false>(slab, taps, [&](int z, int y, int x, f32x2 v2) {
        if (z < nz) *reinterpret_cast<f32x2*>(Tout + ((size_t)z * GS + y) * GS + x) = v2;
      });
      DPC_STAMP(5);
    }
  } else {
    const int WP = odd_stride(W);
    for (int i = tid; i < nz * H * WP; i += nthr) slab[i] = 0.f;
    __syncthreads();
    for_each_record(cells, b, max(z0 - 1, 0), z0 + nz, [&](const PointRec& rec, const int4*) {
      const Cell c = cell_from_record(rec);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int zz = c.iz + k - z0;
        if (zz < 0 || zz >= nz) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int yy = c.iy + j;
          if (yy >= H) continue;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int xx = c.ix + e;
            if (xx >= W) continue;
            atomicAdd(&slab[(zz * H + yy) * WP + xx], c.wz[k] * c.wy[j] * c.wx[e]);  // ds_add_f32
          }
        }
      }
    });
    __syncthreads();
    // clamp mask (bit set <=> raw <= 1; raw >= 0 always) and, when asked for, the raw grid itself
    {
      const int iHW = H * W, wpp = (iHW + 63) / 64;
      const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
      for (int item = wave; item < nz * wpp; item += nw) {
        const int z = item / wpp, c = item - z * wpp;
        const int idx = c * 64 + lane;
        const bool in = idx < iHW;
        const int y = idx / W, x = idx - y * W;
        const float vraw = in ? slab[(z * H + y) * WP + x] : 2.f;
        const unsigned long long bits = __ballot(in && vraw <= 1.0f);
        if (mask != nullptr && lane == 0) mask[((size_t)b * D + z0 + z) * wpp + c] = bits;
        if (raw != nullptr && in) raw[((size_t)b * D + z0 + z) * iHW + idx] = vraw;
      }
    }
    if (Tbuf == nullptr) return;
    float* Tout = Tbuf + ((size_t)b * D + z0) * HW;
    if (RB == 0) {
      const float w2 = taps.w[0] * taps.w[0];
      for (int i = tid; i < nz * H * W; i += nthr) {
        const int x = i % W, zy = i / W;
        Tout[i] = w2 * fminf(slab[zy * WP + x], 1.0f);
      }
      return;
    }
    wpass_inplace<RB, true>(slab, nz, H, W, WP, taps, [](int, int, float val) { return val; });
    hpass<RB>(slab, nz, H, W, WP, taps, [&](int z, int y, int x, float val) { Tout[(z * H + y) * W + x] = val; });
  }
}

// ------------------------------------------------------------------------------------------------------
// Shared per-voxel DRC pieces (dpc/util/drc.py:48-129)
// ------------------------------------------------------------------------------------------------------
struct RayConst {
  float eps, hi;  // clamp bounds eps, 1-eps
  float em1;      // e^eps - 1: the reference's "log-unity" rows are eps, not 0
  float s;        // occupancy scale of this cloud
  bool has_s;
};

// eps-derived constants are computed on the host in fp64 and travel as kernel arguments
struct RayHost {
  float eps, hi, em1;
};

__device__ inline RayConst ray_const(const RayHost& h, const float* s, int b) {
  RayConst r;
  r.eps = h.eps;
  r.hi = h.hi;
  r.em1 = h.em1;
  r.has_s = s != nullptr;
  r.s = r.has_s ? s[b] : 1.0f;
  return r;
}

__device__ inline float occupancy(const RayConst& r, float v2) {  // scale + clamp (point_cloud_to.py:218-222)
  return r.has_s ? fminf(fmaxf(r.s * v2, 0.f), 1.f) : v2;
}

__device__ inline float drc_clamp(const RayConst& r, float v3) { return fminf(fmaxf(v3, r.eps), r.hi); }

// d proj / d v2 for one voxel given the ray's total transmittance; also returns v2 * dL/dv3 * mask for ds.
// Branch-free and short (this loop is not hidden behind memory): 1/(1-y) via v_rcp_f32 (1 ulp; 1-y >= eps);
//   inside = (eps <= v3 <= 1-eps)  <=>  the DRC clamp left v3 unchanged  <=>  y == v3;
//   the scale clamp's pass-through set (0 <= s v2 <= 1) is implied by `inside` (v3 in [eps,1-eps] means the first
//   clamp did not act either), so one select serves both masks.
__device__ inline float drc_voxel_bwd(const RayConst& r, float v2, float g, float Tf, bool first, float& ds_term) {
  const float v3 = occupancy(r, v2);
  const float y = drc_clamp(r, v3);
  float dv3 = g * fmaf(Tf, __builtin_amdgcn_rcpf(1.0f - y), first ? r.em1 : 0.f);
  dv3 = (y == v3) ? dv3 : 0.f;
  if (!r.has_s) {
    ds_term = 0.f;
    return dv3;
  }
  ds_term = v2 * dv3;
  return r.s * dv3;
}

// Fused silhouette loss (dpc/models/model_pc_to.py:339-385, 410-440): cloud b is candidate b % K of sample b / K.
//   forward : sse[b] += sum_pixels (gt - proj)^2            (k_zcol_fwd epilogue; zeroed by k_splat_hw)
//   finalize: winner[s] = argmin_k sse[s*K+k], loss = sum_s min_k sse / S      (k_loss_finalize)
//   backward: dproj = winner ? 2 (proj - gt) / S * dloss : 0, formed on the fly; losing candidates do nothing
struct LossArgs {
  const float* gt;      // [S, H*W] in image orientation (rows already flipped like proj); nullptr = no fused loss
  float* sse;           // [B]
  const int* winner;    // [S] (backward)
  const float* dloss;   // device scalar, gradient arriving at the loss (backward); nullptr = 1
  int K;
  float inv_S;
  float* loss_direct;   // forward, K == 1 only: the scalar loss, accumulated by the ray-march blocks (no finalize launch)
  int* winner_out;      // forward, K == 1 only: zero-filled by k_splat_hw
  int scale_in_gather;  // backward: dT was produced by the forward for dloss = 1; k_gather_hw multiplies by *dloss
};

__global__ __launch_bounds__(256) void k_loss_finalize(const float* __restrict__ sse, int S, int K, float inv_S,
                                                       float* __restrict__ loss, int* __restrict__ winner) {
  __shared__ float red[256 / DPC_WAVE];
  float acc = 0.f;
  for (int smp = threadIdx.x; smp < S; smp += blockDim.x) {
    float best = sse[(size_t)smp * K];
    int bk = 0;
    for (int k = 1; k < K; ++k) {
      const float v = sse[(size_t)smp * K + k];
      if (v < best) { best = v; bk = k; }  // first minimum wins, like torch.argmin
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

// ------------------------------------------------------------------------------------------------------
// Forward 2: D-pass + scale/clamp + DRC silhouette, whole z column in registers.   grid (ceil(HW/256), B)
// ------------------------------------------------------------------------------------------------------
// Epilogue shared by the forward column kernels: silhouette (row flip folded into the index), saved ray
// transmittance, fused loss partial.
__device__ inline void zcol_fwd_epilogue(const DpcParams& P, const RayConst& rc, int b, int ray, bool live, double trans,
                                         float y0, float* __restrict__ proj, float* __restrict__ trans_out,
                                         const LossArgs& la) {
  const int HW = P.H * P.W;
  float sq = 0.f;
  if (live) {
    const int yrow = ray / P.W, x = ray - yrow * P.W;
    const int pix = (P.H - 1 - yrow) * P.W + x;
    const float pr = (float)(1.0 - trans + (double)rc.em1 * (double)y0);
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
      atomicAdd(la.sse + b, tot);
      if (la.loss_direct != nullptr) atomicAdd(la.loss_direct, tot * la.inv_S);  // K == 1: every cloud wins
    }
  }
}

template <int DD, int RB>
__global__ __launch_bounds__(kColThreads, (DD <= 64 ? 4 : 2)) void k_zcol_fwd(DpcParams P, RayHost rh, const float* __restrict__ Tbuf,
                                                          const float* __restrict__ s, TapsT<RB> taps,
                                                          float* __restrict__ smoothed, float* __restrict__ proj,
                                                          float* __restrict__ trans_out, LossArgs la) {
  const int HW = P.H * P.W;
  const int b = blockIdx.y, ray = blockIdx.x * kColThreads + threadIdx.x;
  const bool live = ray < HW;
  const RayConst rc = ray_const(rh, s, b);
  double trans = 1.0;
  float y0 = 0.f;
  if (live) {
    const float* col = Tbuf + (size_t)b * DD * HW + ray;
    float* out = smoothed + (size_t)b * DD * HW + ray;
    float c[DD];
#pragma unroll
    for (int z = 0; z < DD; ++z) c[z] = col[(size_t)z * HW];
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
  zcol_fwd_epilogue(P, rc, b, ray, live, trans, y0, proj, trans_out, la);
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
template <int DD, int RB, int RPL>
__global__ __launch_bounds__(kColThreads, (RPL * DD <= 128 ? 2 : 1))
void k_zcol_fwdbwd(DpcParams P, RayHost rh, const float* __restrict__ Tbuf, const float* __restrict__ s, TapsT<RB> taps,
                   TapsT<RB> taps_adj, float* __restrict__ proj, float* __restrict__ dT, float* __restrict__ ds_part,
                   int n_ds_part, unsigned long long* __restrict__ tickets, float* __restrict__ dsmall, LossArgs la) {
  typedef float vec __attribute__((ext_vector_type(RPL)));
  const int HW = P.H * P.W;
  const int b = blockIdx.y, ray = RPL * (blockIdx.x * kColThreads + threadIdx.x);
  const bool live = ray < HW;
  const RayConst rc = ray_const(rh, s, b);
  float sq = 0.f, ds_acc = 0.f;
  float y[DD][RPL], g[RPL], gT[RPL];
  if (live) {
    const float* col = Tbuf + (size_t)b * DD * HW + ray;
    float c[DD][RPL];  // T column, then y (encoded), then dL/dv3: each value dies as the next is born
#pragma unroll
    for (int z = 0; z < DD; ++z) {
      const vec v = *reinterpret_cast<const vec*>(col + (size_t)z * HW);
#pragma unroll
      for (int r = 0; r < RPL; ++r) c[z][r] = v[r];
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
  // the squared error in fixed point (kSseFrac fractional bits) plus a block count in the top bits.  The returned value
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
    mine = (1ull << kSseCountShift) | (unsigned long long)((double)tot * (double)(1ull << kSseFrac) + 0.5);
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
          for (int k = 0; k < 2 * RB + 1; ++k) {
            const int zz = zo + k - RB;
            if (zz >= 0 && zz < DD) a = fmaf(wadj[k], y[zz][r], a);
          }
          acc[r] = a;
        }
        if (!DPC_ABL(18) || acc[0] == 123.456f) *reinterpret_cast<vec*>(out + (size_t)zo * HW) = acc;
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
    for (int i = blockIdx.x; i < n_ds_part; i += gridDim.x) ds_part[(size_t)b * n_ds_part + i] = i == (int)blockIdx.x ? dst : 0.f;
    if ((before >> kSseCountShift) == gridDim.x - 1) {  // every other block of this cloud has added its share
      const unsigned long long sum = (before + mine) & ((1ull << kSseCountShift) - 1);
      const float tot = (float)((double)sum * (1.0 / (double)(1ull << kSseFrac)));
      la.sse[b] = tot;
      atomicAdd(la.loss_direct, tot * la.inv_S);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < DPC_SMALL_COLS) dsmall[(size_t)threadIdx.x * gridDim.y + b] = 0.f;  // [col][B]
}

// Generic depth / tap count: same arithmetic, column re-read from global (L1/L2 serve the re-reads).
__global__ __launch_bounds__(kColThreads) void k_zcol_fwd_dyn(DpcParams P, RayHost rh, const float* __restrict__ Tbuf,
                                                              const float* __restrict__ s, TapsDyn taps,
                                                              float* __restrict__ smoothed, float* __restrict__ proj,
                                                              float* __restrict__ trans_out, LossArgs la) {
  const int HW = P.H * P.W, D = P.D;
  const int b = blockIdx.y, ray = blockIdx.x * kColThreads + threadIdx.x;
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
  zcol_fwd_epilogue(P, rc, b, ray, live, trans, y0, proj, trans_out, la);
}

// ------------------------------------------------------------------------------------------------------
// Backward 1: DRC backward + scale/clamp backward + adjoint D-pass.                grid (ceil(HW/256), B)
//   Also zeroes the dq/dt/df accumulators that k_gather_hw adds into, and writes this tile's ds partial.
// ------------------------------------------------------------------------------------------------------
// Gradient arriving at this ray's silhouette pixel: either read from dproj, or formed from the fused loss.
__device__ inline bool cloud_loses(const LossArgs& la, int b) {
  return la.gt != nullptr && la.winner[b / la.K] != b % la.K;
}
__device__ inline float ray_grad(const DpcParams& P, const LossArgs& la, const float* __restrict__ dproj,
                                 const float* __restrict__ proj, int b, int ray) {
  const int HW = P.H * P.W;
  const int yrow = ray / P.W, x = ray - yrow * P.W;
  const int pix = (P.H - 1 - yrow) * P.W + x;
  if (la.gt == nullptr) return dproj[(size_t)b * HW + pix];
  const float up = la.dloss ? *la.dloss : 1.0f;
  return 2.0f * la.inv_S * up * (proj[(size_t)b * HW + pix] - la.gt[(size_t)(b / la.K) * HW + pix]);
}

__device__ inline void zcol_bwd_epilogue(float ds_acc, float* ds_part, float* dsmall, int b) {
  __shared__ float red[kColThreads / DPC_WAVE];
  const float w = wave_sum(ds_acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < kColThreads / DPC_WAVE; ++i) tot += red[i];
    ds_part[(size_t)b * gridDim.x + blockIdx.x] = tot;
  }
  if (blockIdx.x == 0 && threadIdx.x < DPC_SMALL_COLS) dsmall[(size_t)threadIdx.x * gridDim.y + b] = 0.f;  // [col][B]
}

// Reads the grid saved by the forward slab kernel (after clamp + W/H passes), recomputes the forward D-pass in
// registers (cheaper than having the forward write, and this kernel read, a second full grid), then DRC backward,
// scale/clamp backward and the adjoint D-pass.
template <int DD, int RB>
__global__ __launch_bounds__(kColThreads, 2) void k_zcol_bwd(DpcParams P, RayHost rh, const float* __restrict__ Tin,
                                                          const float* __restrict__ s,
                                                          const float* __restrict__ dproj, const float* __restrict__ proj,
                                                          const float* __restrict__ trans_in, TapsT<RB> taps,
                                                          TapsT<RB> taps_adj,
                                                          float* __restrict__ dT, float* __restrict__ ds_part,
                                                          float* __restrict__ dsmall, LossArgs la) {
  const int HW = P.H * P.W;
  const int b = blockIdx.y, ray = blockIdx.x * kColThreads + threadIdx.x;
  float ds_acc = 0.f;
  if (ray < HW && !cloud_loses(la, b)) {
    const RayConst rc = ray_const(rh, s, b);
    const float* col = Tin + (size_t)b * DD * HW + ray;
    float c[DD], d[DD];
#pragma unroll
    for (int z = 0; z < DD; ++z) c[z] = col[(size_t)z * HW];
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
    float* out = dT + (size_t)b * DD * HW + ray;
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
        for (int k = 0; k < 2 * RB + 1; ++k) {
          const int zz = zo + k - RB;
          if (zz >= 0 && zz < DD) acc = fmaf(taps_adj.w[k], d[zz], acc);
        }
        out[(size_t)zo * HW] = acc;
      }
      // keep the unrolled per-voxel chains from being interleaved across voxels (it would spill the columns)
      if ((z & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  zcol_bwd_epilogue(ds_acc, ds_part, dsmall, b);
}

__global__ __launch_bounds__(kColThreads) void k_zcol_bwd_dyn(DpcParams P, RayHost rh, const float* __restrict__ Tin,
                                                              const float* __restrict__ s,
                                                              const float* __restrict__ dproj, const float* __restrict__ proj,
                                                              const float* __restrict__ trans_in, TapsDyn taps,
                                                              TapsDyn taps_adj,
                                                              float* __restrict__ dT, float* __restrict__ ds_part,
                                                              float* __restrict__ dsmall, LossArgs la) {
  const int HW = P.H * P.W, D = P.D;
  const int b = blockIdx.y, ray = blockIdx.x * kColThreads + threadIdx.x;
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
      out[(size_t)z * HW] = acc;
    }
  }
  zcol_bwd_epilogue(ds_acc, ds_part, dsmall, b);
}

// ------------------------------------------------------------------------------------------------------
// Backward 2: adjoint H/W passes + clamp mask + trilinear gather + transform backward.   grid (nslab, B)
//   The slab holds cell layers [z0, z0+Zs) plus one halo plane so each point's 8 corners are local.
// ------------------------------------------------------------------------------------------------------
template <int GS, int ZS, int RB>
__global__ __launch_bounds__(kSlabThreads) void k_gather_hw(DpcParams P, Cells cells, const float* __restrict__ pc,
                                                            const float* __restrict__ q, const float* __restrict__ t,
                                                            const float* __restrict__ f, TapsT<RB> taps_adj, int zs_rt,
                                                            const float* __restrict__ dT,
                                                            const uint64_t* __restrict__ mask,
                                                            const float* __restrict__ ds_part, int n_ds_part,
                                                            float* __restrict__ dpc, float* __restrict__ dsmall,
                                                            LossArgs la) {
  extern __shared__ __attribute__((aligned(16))) float slab[];
  const int D = P.D, H = P.H, W = P.W, N = P.N, HW = H * W;
  const int Zs = GS ? ZS : zs_rt;
  // One-layer slabs (planes too big for more: 128^2) ROLL: the workgroup walks `roll` consecutive layers, keeps the plane
  // two layers share in LDS (ping-pong of the two plane buffers) and pays start-up, reduction and atomics once.
  constexpr bool kRolls = GS > 0 && ZS == 1 && RB > 0;
  const int roll = kRolls ? zs_rt : 1;
  const int b = blockIdx.y, z0 = blockIdx.x * Zs * roll;
  const int reps = P.point_replicas > 1 ? P.point_replicas : 1;
  const bool shared_points = reps > 1;  // dpc is [B/reps,N,3], zeroed by the caller; replicas add into it
  if (cloud_loses(la, b)) {  // a losing pose candidate: zero gradient, no work (block-uniform)
    if (shared_points) {
      if (blockIdx.x == 0 && threadIdx.x == 0) dsmall[(size_t)DPC_COL_DS * P.B + b] = 0.f;
      return;
    }
    float* dz = dpc + (size_t)b * N * 3;
    auto zero3 = [&](const PointRec&, const int4* aux) {
      const int i = aux->w;
      dz[3 * i + 0] = 0.f; dz[3 * i + 1] = 0.f; dz[3 * i + 2] = 0.f;
    };
    for_each_record(cells, b, z0, min(z0 + Zs * roll, D), zero3);
    if (blockIdx.x == 0) {
      for_each_record(cells, b, D, D + 1, zero3);
      if (threadIdx.x == 0) dsmall[(size_t)DPC_COL_DS * P.B + b] = 0.f;
    }
    return;
  }
  const int nzp = min(Zs + 1, D - z0);  // planes present (cell layers + halo)
  const int tid = threadIdx.x, nthr = blockDim.x;
  // camera inputs and the upstream scalar: requested now, first used after the slab is in LDS
  const CameraRaw cam_raw = load_camera_raw(P, q, t, f, b);
  const float upstream = (la.scale_in_gather && la.dloss != nullptr) ? *la.dloss : 1.0f;
  const int wpp = (HW + 63) / 64;
  const float* src = dT + ((size_t)b * D + z0) * HW;
  const uint64_t* mrow = mask + ((size_t)b * D + z0) * wpp;
  float* red;

  if constexpr (GS > 0) {
    constexpr int NPL = ZS + 1;
    using Geo = BwdGeo<GS, RB, NPL>;
    red = slab + ((Geo::slab_floats(NPL) + 3) / 4) * 4;
    RecordRange rr{0, 0};
    if (cells.nblk <= DPC_WAVE) rr = load_record_range(cells, b, z0, min(z0 + Zs, D));  // in flight under the H-pass
    const uint32_t* mask32 = reinterpret_cast<const uint32_t*>(mrow);
    DPC_STAMP(8);
    for (int i = tid; i < (NPL * GS + 1) * (Geo::PAD / 4); i += Geo::NT) {  // zero the row pads (W-pass halo)
      const int p4 = i % (Geo::PAD / 4), row = i / (Geo::PAD / 4);
      *reinterpret_cast<f32x4*>(slab + row * Geo::WP + 4 * p4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (RB == 0) {
      // planes -> LDS (16-byte global loads, 16-byte LDS stores); absent planes are zeroed
      for (int i = tid; i < NPL * GS * (GS / 4); i += Geo::NT) {
        const int x4 = i % (GS / 4), zy = i / (GS / 4);
        f32x4 val = f32x4{0.f, 0.f, 0.f, 0.f};
        if (zy < nzp * GS) val = *reinterpret_cast<const f32x4*>(src + (size_t)zy * GS + 4 * x4);
        *reinterpret_cast<f32x4*>(slab + zy * Geo::WP + Geo::PAD + 4 * x4) = val;
      }
      if (cells.nblk <= DPC_WAVE) finish_record_table(rr, reinterpret_cast<int*>(red + kRedTab));
      __syncthreads();
      const float w2 = taps_adj.w[0] * taps_adj.w[0];
      for (int i = tid; i < NPL * GS * GS; i += Geo::NT) {
        const int x = i % GS, zy = i / GS;
        const bool pass = zy < nzp * GS && ((mask32[i >> 5] >> (i & 31)) & 1u);
        float* cell = slab + zy * Geo::WP + Geo::PAD + x;
        *cell = pass ? w2 * *cell : 0.f;
      }
      __syncthreads();
    } else {
      // adjoint H-pass with its windows read straight from global dT (lanes walk x: coalesced; the halo rows
      // shared by neighbouring segments come from L1/L2), results stored to LDS once
      // The adjoint W-pass is NOT run over the slab: only ~8 voxels per point are ever gathered (64k per cloud vs
      // 262k voxels), so it is evaluated at the gathered corners below.  The clamp-mask words of the slab's planes
      // are staged in LDS (requested now, stored after the H-pass so their latency hides under it).
      constexpr int MW = NPL * GS * (GS / 32), MPT = (MW + Geo::NT - 1) / Geo::NT;
      uint32_t mreg[MPT];
#pragma unroll
      for (int it = 0; it < MPT; ++it) {
        const int w = tid + it * Geo::NT;
        mreg[it] = (w < MW && w / (GS * (GS / 32)) < nzp) ? mask32[w] : 0u;
      }
      hpass_global<Geo, GS, RB, NPL>(src, nzp, taps_adj, [&](int z, int y, int x, f32x2 val) {
        *reinterpret_cast<f32x2*>(slab + Geo::at(z, y, x)) = val;
      });
      uint32_t* mlds = reinterpret_cast<uint32_t*>(red + kRedMask);
#pragma unroll
      for (int it = 0; it < MPT; ++it) {
        const int w = tid + it * Geo::NT;
        if (w < MW) mlds[w] = mreg[it];
      }
      if (cells.nblk <= DPC_WAVE) finish_record_table(rr, reinterpret_cast<int*>(red + kRedTab));
      __syncthreads();
      DPC_STAMP(9);
      DPC_STAMP(10);
    }
  } else {
    const int WP = odd_stride(W);
    red = slab + (size_t)(Zs + 1) * H * WP;
    for (int i = tid; i < nzp * HW; i += nthr) {
      const int x = i % W, zy = i / W;
      slab[zy * WP + x] = src[i];
    }
    __syncthreads();
    if (RB == 0) {
      const float w2 = taps_adj.w[0] * taps_adj.w[0];
      for (int i = tid; i < nzp * HW; i += nthr) {
        const int z = i / HW, r = i - z * HW;
        const int x = r % W, y = r / W;
        const bool pass = (mrow[(size_t)z * wpp + (r >> 6)] >> (r & 63)) & 1ull;
        slab[(z * H + y) * WP + x] = pass ? w2 * slab[(z * H + y) * WP + x] : 0.f;
      }
      __syncthreads();
    } else {
      hpass<RB>(slab, nzp, H, W, WP, taps_adj, [&](int z, int y, int x, float val) { slab[(z * H + y) * WP + x] = val; });
      wpass_inplace<RB, false>(slab, nzp, H, W, WP, taps_adj, [&](int line, int x, float val) {
        const int z = line / H, y = line - z * H;
        const int bit = y * W + x;
        return ((mrow[(size_t)z * wpp + (bit >> 6)] >> (bit & 63)) & 1ull) ? val : 0.f;
      });
    }
  }

  // gather: every in-bounds point belongs to the slab of its cell layer iz; slab 0 also zero-fills the
  // gradient of the out-of-bounds points (bin D)
  if (DPC_ABL(12)) return;
  const Camera cam = make_camera(P, cam_raw);
  CamGrad g;
  camgrad_zero(g);
  float* dcloud = dpc + (size_t)(b / reps) * N * 3;
  auto corner = [&](int zz, int yy, int xx) -> float {
    if constexpr (GS > 0) return slab[BwdGeo<GS, RB, ZS + 1>::at(zz, yy, xx)];
    else return slab[(zz * H + yy) * odd_stride(W) + xx];
  };
  auto gather = [&](const PointRec& rec, const int4* aux) {
    const int4 pt = *aux;  // {px, py, pz, original index}: one 16-byte load, issued next to the record's
    const int i = pt.w;
    const Cell c = cell_from_record(rec);
    float cv[2][2][2];
    if constexpr (GS > 0 && RB > 0) {
      // adjoint W-pass evaluated right here, at the two x corners of each of the four (z,y) rows, then masked
      using Geo = BwdGeo<GS, RB, ZS + 1>;
      const uint32_t* mlds = reinterpret_cast<const uint32_t*>(red + kRedMask);
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          cv[k][j][0] = cv[k][j][1] = 0.f;
          if ((c.iz + k < D) && (c.iy + j < GS)) {
            const int plane = kRolls ? ((c.iz - z0 + k) & 1) : (c.iz - z0 + k);  // rolling: plane z lives in buffer (z - z0) & 1
            const int row = plane * GS + c.iy + j;
            const float* rp = slab + row * Geo::WP + Geo::PAD + c.ix - RB;  // x = ix-RB .. ix+1+RB, pads are zero
            float v[2 * RB + 2];
#pragma unroll
            for (int i = 0; i < 2 * RB + 2; ++i) v[i] = rp[i];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              float acc = 0.f;
#pragma unroll
              for (int tp = 0; tp < 2 * RB + 1; ++tp) acc = fmaf(taps_adj.w[tp], v[e + tp], acc);
              const int x = c.ix + e;
              const bool pass = x < GS && ((mlds[row * (GS / 32) + (x >> 5)] >> (x & 31)) & 1u);
              cv[k][j][e] = pass ? acc : 0.f;
            }
          }
        }
    } else {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const bool ok = (c.iz + k < D) && (c.iy + j < H) && (c.ix + e < W);
            cv[k][j][e] = ok ? corner(c.iz - z0 + k, c.iy + j, c.ix + e) : 0.f;
          }
    }
    float dgz = 0.f, dgy = 0.f, dgx = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        dgz += (cv[1][a][e] - cv[0][a][e]) * c.wy[a] * c.wx[e];
        dgy += (cv[a][1][e] - cv[a][0][e]) * c.wz[a] * c.wx[e];
        dgx += (cv[a][e][1] - cv[a][e][0]) * c.wz[a] * c.wy[e];
      }
    dgz *= upstream; dgy *= upstream; dgx *= upstream;  // 1 unless dT was produced by the forward for dloss = 1
    const float px = __int_as_float(pt.x), py = __int_as_float(pt.y), pz = __int_as_float(pt.z);
    const Projected o = project_point(cam, px, py, pz);
    float dpx, dpy, dpz;
    project_point_bwd(cam, o, px, py, pz, dgz * (float)(D - 1), dgy * (float)(H - 1), dgx * (float)(W - 1), dpx, dpy, dpz, g);
    if (shared_points) {
      atomicAdd(dcloud + 3 * i + 0, dpx); atomicAdd(dcloud + 3 * i + 1, dpy); atomicAdd(dcloud + 3 * i + 2, dpz);
    } else {
      dcloud[3 * i + 0] = dpx; dcloud[3 * i + 1] = dpy; dcloud[3 * i + 2] = dpz;
    }
  };
  if (!DPC_ABL(10)) {
    if (GS > 0 && cells.nblk <= DPC_WAVE) for_each_record_flat(cells, b, reinterpret_cast<const int*>(red + kRedTab), gather);
    else for_each_record(cells, b, z0, min(z0 + Zs, D), gather);
  }
  if constexpr (kRolls) {
    using Geo = BwdGeo<GS, RB, 2>;
    constexpr int MW1 = GS * (GS / 32), MPT1 = (MW1 + Geo::NT - 1) / Geo::NT;  // clamp-mask words of one plane
    const uint32_t* mask32 = reinterpret_cast<const uint32_t*>(mrow);
    uint32_t* mlds = reinterpret_cast<uint32_t*>(red + kRedMask);
    int* tab = reinterpret_cast<int*>(red + kRedTab);
    const bool flat = cells.nblk <= DPC_WAVE;
    for (int l = 1; l < roll && z0 + l < D; ++l) {
      __syncthreads();  // layer l-1 is gathered: plane z0+l-1 (buffer (l-1)&1 == (l+1)&1) and the record table are free
      const int buf = (l + 1) & 1;
      const bool present = z0 + l + 1 < D;
      RecordRange rr{0, 0};
      if (flat) rr = load_record_range(cells, b, z0 + l, z0 + l + 1);
      uint32_t mreg[MPT1];
#pragma unroll
      for (int it = 0; it < MPT1; ++it) {
        const int w = tid + it * Geo::NT;
        mreg[it] = (w < MW1 && present) ? mask32[(size_t)(l + 1) * MW1 + w] : 0u;
      }
      hpass_global<Geo, GS, RB, 1>(src + (size_t)(l + 1) * HW, present ? 1 : 0, taps_adj, [&](int, int y, int x, f32x2 val) {
        *reinterpret_cast<f32x2*>(slab + Geo::at(buf, y, x)) = val;
      });
#pragma unroll
      for (int it = 0; it < MPT1; ++it) {
        const int w = tid + it * Geo::NT;
        if (w < MW1) mlds[buf * MW1 + w] = mreg[it];
      }
      if (flat) finish_record_table(rr, tab);
      __syncthreads();
      if (flat) for_each_record_flat(cells, b, tab, gather);
      else for_each_record(cells, b, z0 + l, z0 + l + 1, gather);
    }
  }
  DPC_STAMP(11);
  if (blockIdx.x == 0 && !shared_points)
    for_each_record(cells, b, D, D + 1, [&](const PointRec&, const int4* aux) {
      const int i = aux->w;
      dcloud[3 * i + 0] = 0.f; dcloud[3 * i + 1] = 0.f; dcloud[3 * i + 2] = 0.f;
    });

  float vals[13];
#pragma unroll
  for (int i = 0; i < 9; ++i) vals[i] = g.m[i];
  vals[9] = g.dt[0]; vals[10] = g.dt[1]; vals[11] = g.dt[2]; vals[12] = g.df;
  if (DPC_ABL(13)) { if (vals[0] == 123.f) dsmall[0] = vals[1]; return; }
  block_sum<13>(vals, red);
  DPC_STAMP(12);
  if (DPC_ABL(14)) { if (vals[0] == 123.f) dsmall[0] = vals[1]; return; }
  if (tid == 0) {
    float dq[4];
    quaternion_grad(cam, vals, dq);
    // dsmall is [DPC_SMALL_COLS][B] with dq stored as a [B,4] block, dt as a [B,3] block (see dpc_render.h)
    float* dqb = dsmall + (size_t)DPC_COL_DQ * P.B + (size_t)b * 4;
    float* dtb = dsmall + (size_t)DPC_COL_DT * P.B + (size_t)b * 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) atomicAdd(dqb + i, dq[i]);
    if (t != nullptr)
      for (int i = 0; i < 3; ++i) atomicAdd(dtb + i, vals[9 + i]);
    if (f != nullptr) atomicAdd(dsmall + (size_t)DPC_COL_DF * P.B + b, vals[12]);
    if (blockIdx.x == 0) {
      float ds = 0.f;
      for (int i = 0; i < n_ds_part; ++i) ds += ds_part[(size_t)b * n_ds_part + i];
      dsmall[(size_t)DPC_COL_DS * P.B + b] = ds * upstream;
    }
  }
  DPC_STAMP(13);
}

// ------------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------------
struct TapPlan {
  int taps;    // original length
  int radius;  // effective radius after dropping negligible outer taps
  int bucket;  // compile-time radius bucket, -1 = needs the generic column kernel / staged path
};

// Outer taps whose total |weight| is below 1e-8 of the kernel's mass change no fp32 result at the 1e-5
// parity tolerance (inputs are clamped to [0,1], so each pass errs by < 1e-8); for sigma_rel = 0.64 this
// keeps 7 of 21 taps (the dropped +-4 taps weigh 2e-9 each).
TapPlan plan_taps(const float* k, int taps) {
  TapPlan p{taps, 0, 0};
  if (taps <= 0) return p;
  const int c = (taps - 1) / 2;
  double total = 0.0;
  for (int i = 0; i < taps; ++i) total += fabs((double)k[i]);
  int r = c;
  double dropped = 0.0;
  while (r > 0) {
    const double d = fabs((double)k[c - r]) + fabs((double)k[c + r]);
    if (dropped + d > 1e-8 * total) break;
    dropped += d;
    --r;
  }
  p.radius = r;
  static const int buckets[] = {0, 1, 2, 3, 4, 6, 10, 15};
  p.bucket = -1;
  for (int bk : buckets)
    if (r <= bk) {
      p.bucket = bk;
      break;
    }
  return p;
}

template <int RB>
TapsT<RB> make_taps(const float* k, const TapPlan& p, bool flip) {
  TapsT<RB> t;
  for (int i = 0; i < 2 * RB + 1; ++i) t.w[i] = 0.f;
  if (p.taps > 0) {
    const int c = (p.taps - 1) / 2;
    for (int o = -p.radius; o <= p.radius; ++o) t.w[RB + o] = k[c + (flip ? -o : o)];
  } else {
    t.w[RB] = 1.f;
  }
  return t;
}

TapsDyn make_taps_dyn(const float* k, int taps, bool flip) {
  TapsDyn t;
  t.n = taps;
  for (int i = 0; i < DPC_MAX_TAPS; ++i) t.w[i] = 0.f;
  for (int i = 0; i < taps; ++i) t.w[i] = k[flip ? taps - 1 - i : i];
  return t;
}

int validate(const DpcParams* p) {
  if (p == nullptr) return DPC_ERR_NULL;
  if (p->B < 0 || p->N < 0 || p->D < 1 || p->H < 1 || p->W < 1) return DPC_ERR_SHAPE;
  if (p->D > 1024 || p->H > 1024 || p->W > 1024 || p->B > 65535) return DPC_ERR_SHAPE;  // 10-bit cell indices
  if (p->point_replicas < 0 || (p->point_replicas > 1 && p->B % p->point_replicas != 0)) return DPC_ERR_SHAPE;
  for (int taps : {p->taps_xy, p->taps_z})
    if (taps < 0 || taps > DPC_MAX_TAPS || (taps > 0 && taps % 2 == 0)) return DPC_ERR_TAPS;
  return DPC_OK;
}

// planes of an H x W slab that fit the LDS tile of the generic kernels
int planes_fit(const DpcParams* p) { return kLdsBudget / ((p->H * (p->W | 1)) * (int)sizeof(float)); }
int slab_threads(const DpcParams* p) { return (long long)p->H * p->W >= 2048 ? kSlabThreads : 256; }
int col_tiles(const DpcParams* p) { return (p->H * p->W + kColThreads - 1) / kColThreads; }

template <class K>
int set_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes) == hipSuccess
             ? DPC_OK
             : DPC_ERR_LAUNCH;
}

RayHost ray_host(const DpcParams* p) {
  const double eps = (double)p->clip_val;
  return RayHost{p->clip_val, (float)(1.0 - eps), (float)expm1(eps)};
}

int launch_ok() { return hipGetLastError() == hipSuccess ? DPC_OK : DPC_ERR_LAUNCH; }

Cells cells_view(const DpcParams* p, const void* cells) {
  return Cells{static_cast<const uint8_t*>(cells), chunk_bytes(p->D), num_chunks(p->N)};
}

#define DPC_FOR_BUCKET(bucket, MACRO) \
  switch (bucket) {                   \
    case 0: MACRO(0); break;          \
    case 1: MACRO(1); break;          \
    case 2: MACRO(2); break;          \
    case 3: MACRO(3); break;          \
    case 4: MACRO(4); break;          \
    case 6: MACRO(6); break;          \
    case 10: MACRO(10); break;        \
    case 15: MACRO(15); break;        \
    default: rc = DPC_ERR_TAPS;       \
  }

#ifndef DPC_FWD_ZS64
#define DPC_FWD_ZS64 4
#endif
constexpr int kFwdZs64 = DPC_FWD_ZS64;
#ifndef DPC_BWD_ZS64
#define DPC_BWD_ZS64 8
#endif
constexpr int kBwdZs64 = DPC_BWD_ZS64;  // cell layers per backward slab at G = 64 (8 -> 9 planes, 1 workgroup/CU; 3 -> 4 planes, 2/CU)  // planes per forward slab at G = 64 (4 -> 1 workgroup/CU, 2 -> 2 workgroups/CU)

// ---- slab kernel dispatch: specialised when H = W in {32, 64, 128} and the padded slab fits, else generic
template <int GS, int ZS, int RB>
int launch_splat_fast(const DpcParams* p, Cells cells, const float* kxy, const TapPlan& pxy, float* raw, float* Tbuf,
                      uint64_t* mask, float* sse, float* loss_zero, int* winner_zero, unsigned long long* ticket_zero, hipStream_t st) {
  using Geo = FwdGeo<GS, ZS, RB>;
  constexpr size_t lds = ((size_t)ZS * GS * Geo::WPA + Geo::PAD) * sizeof(unsigned long long) + kTabInts * sizeof(int);
  static_assert(lds >= Geo::slab_floats(ZS) * sizeof(float), "the fp32 slab reuses the accumulator memory");
  static_assert(lds <= kLdsLimit, "forward slab does not fit LDS");
  auto kern = k_splat_hw<GS, ZS, RB>;
  int rc = set_lds(kern, lds);
  if (rc != DPC_OK) return rc;
  DPC_LAUNCH("k_splat_hw", kern, dim3((p->D + ZS - 1) / ZS, p->B), dim3(Geo::NT), lds, st, *p, cells,
             make_taps<RB>(kxy, pxy, false), ZS, raw, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero);
  return launch_ok();
}

template <int RB>
int launch_splat(const DpcParams* p, Cells cells, const float* kxy, const TapPlan& pxy, float* raw, float* Tbuf,
                 uint64_t* mask, float* sse, float* loss_zero, int* winner_zero, unsigned long long* ticket_zero, hipStream_t st) {
  // The specialised kernel's W-pass branch does not produce the optional unclamped grid (`raw`, asked for by direct
  // users of dpc_project_fwd only): that request takes the generic kernel, which keeps the hot loop free of it.
  const bool raw_with_passes = raw != nullptr && Tbuf != nullptr && RB > 0;
  if (p->H == p->W && !raw_with_passes) {
    if constexpr (RB <= 4) {
      if (p->H == 32) return launch_splat_fast<32, 4, RB>(p, cells, kxy, pxy, raw, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, st);
      if (p->H == 128) return launch_splat_fast<128, 1, RB>(p, cells, kxy, pxy, raw, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, st);
    }
    if constexpr (RB <= 10) {
      if (p->H == 64) return launch_splat_fast<64, kFwdZs64, RB>(p, cells, kxy, pxy, raw, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, st);
      if constexpr (RB > 4) {
        if (p->H == 128) return launch_splat_fast<128, 1, RB>(p, cells, kxy, pxy, raw, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, st);
        if (p->H == 32) return launch_splat_fast<32, 4, RB>(p, cells, kxy, pxy, raw, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, st);
      }
    }
  }
  const int fit = planes_fit(p);
  if (fit < 1) return DPC_ERR_LDS;
  const int Zs = std::min(fit, std::max(1, (p->D + 7) / 8));
  const size_t lds = (size_t)Zs * p->H * (p->W | 1) * sizeof(float);
  auto kern = k_splat_hw<0, 0, RB>;
  int rc = set_lds(kern, lds);
  if (rc != DPC_OK) return rc;
  DPC_LAUNCH("k_splat_hw", kern, dim3((p->D + Zs - 1) / Zs, p->B), dim3(slab_threads(p)), lds, st, *p, cells,
             make_taps<RB>(kxy, pxy, false), Zs, raw, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero);
  return launch_ok();
}

template <int GS, int ZS, int RB>
int launch_gather_fast(const DpcParams* p, Cells cells, const float* pc, const float* q, const float* t, const float* f,
                       const float* kxy, const TapPlan& pxy, const float* dT, const uint64_t* mask,
                       const float* ds_part, int ntile, float* dpc, float* dsmall, const LossArgs& la, hipStream_t st) {
  using Geo = BwdGeo<GS, RB, ZS + 1>;
  // slab + scratch tail: reduction floats, record table, staged mask words
  constexpr size_t lds = (((Geo::slab_floats(ZS + 1) + 3) / 4) * 4 + kRedMask + (ZS + 1) * GS * (GS / 32)) * sizeof(float);
  static_assert(lds <= kLdsLimit, "backward slab does not fit LDS");
  static_assert(kRedTab >= 13 * (Geo::NT / DPC_WAVE) && kRedMask >= kRedTab + kTabInts, "scratch tail layout");
  auto kern = k_gather_hw<GS, ZS, RB>;
  int rc = set_lds(kern, lds);
  if (rc != DPC_OK) return rc;
  // one-layer slabs roll over several layers per workgroup (see the kernel): as many as still leave a workgroup per CU
  int roll = 1;
  if (ZS == 1 && RB > 0)
    for (int c = 2; c <= 16; c *= 2)
      if (p->D % c == 0 && (size_t)(p->D / c) * p->B >= (size_t)kNumCUs) roll = c;
  const int nslab = (p->D + ZS - 1) / ZS;
  DPC_LAUNCH("k_gather_hw", kern, dim3((nslab + roll - 1) / roll, p->B), dim3(Geo::NT), lds, st, *p, cells, pc, q, t, f,
             make_taps<RB>(kxy, pxy, true), ZS == 1 ? roll : ZS, dT, mask, ds_part, ntile, dpc, dsmall, la);
  return launch_ok();
}

template <int RB>
int launch_gather(const DpcParams* p, Cells cells, const float* pc, const float* q, const float* t, const float* f,
                  const float* kxy, const TapPlan& pxy, const float* dT, const uint64_t* mask, const float* ds_part,
                  int ntile, float* dpc, float* dsmall, const LossArgs& la, hipStream_t st) {
  if (p->H == p->W) {
    if constexpr (RB <= 4) {
      if (p->H == 32) return launch_gather_fast<32, 4, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, la, st);
      if (p->H == 128) return launch_gather_fast<128, 1, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, la, st);
      if (p->H == 64) return launch_gather_fast<64, kBwdZs64, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, la, st);
    } else if constexpr (RB <= 10) {
      if (p->H == 64) return launch_gather_fast<64, 7, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, la, st);
      if (p->H == 128) return launch_gather_fast<128, 1, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, la, st);  // c4: sigma_rel 1.28 -> radius 8
      if (p->H == 32) return launch_gather_fast<32, 4, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, la, st);
    }
  }
  const int fit = planes_fit(p);
  if (fit < 2) return DPC_ERR_LDS;
  const int Zs = std::min(fit - 1, std::max(1, (p->D + 7) / 8));
  const size_t lds = ((size_t)(Zs + 1) * p->H * (p->W | 1) + kRedFloats) * sizeof(float);
  auto kern = k_gather_hw<0, 0, RB>;
  int rc = set_lds(kern, lds);
  if (rc != DPC_OK) return rc;
  DPC_LAUNCH("k_gather_hw", kern, dim3((p->D + Zs - 1) / Zs, p->B), dim3(slab_threads(p)), lds, st, *p, cells, pc, q, t, f,
             make_taps<RB>(kxy, pxy, true), Zs, dT, mask, ds_part, ntile, dpc, dsmall, la);
  return launch_ok();
}

int launch_locate(const DpcParams* p, int src, const void* pts, const float* q, const float* t, const float* f,
                  float* tr_pc, void* cells, hipStream_t st) {
  if (p->N == 0 || p->B == 0) return DPC_OK;
  dim3 g(num_chunks(p->N), p->B), blk(kLocThreads);
  uint8_t* out = static_cast<uint8_t*>(cells);
  if (src == 0) DPC_LAUNCH("k_locate", k_locate<0>, g, blk, 0, st, *p, pts, q, t, f, tr_pc, out);
  else if (src == 1) DPC_LAUNCH("k_locate", k_locate<1>, g, blk, 0, st, *p, pts, q, t, f, tr_pc, out);
  else DPC_LAUNCH("k_locate", k_locate<2>, g, blk, 0, st, *p, pts, q, t, f, tr_pc, out);
  return launch_ok();
}

}  // namespace

extern "C" {

size_t dpc_mask_words_per_plane(const DpcParams* p) { return p ? ((size_t)p->H * p->W + 63) / 64 : 0; }

size_t dpc_cells_bytes(const DpcParams* p) {
  if (validate(p) != DPC_OK) return 0;
  return (size_t)p->B * num_chunks(p->N) * chunk_bytes(p->D);
}

size_t dpc_workspace_bytes(const DpcParams* p) {
  if (validate(p) != DPC_OK) return 0;
  const size_t grid = (size_t)p->B * p->D * p->H * p->W * sizeof(float);
  const size_t parts = (size_t)p->B * col_tiles(p) * sizeof(float) + (size_t)p->B * 8 + 8;  // ds partials, sum-and-count words
  return ((grid + 255) / 256) * 256 + ((parts + 255) / 256) * 256;
}

int dpc_locate(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f, float* tr_pc,
               void* cells, void* stream) {
  int rc = validate(p);
  if (rc != DPC_OK) return rc;
  if (p->B == 0 || p->N == 0) return DPC_OK;
  if (!pc || !q || !cells) return DPC_ERR_NULL;
  return launch_locate(p, 0, pc, q, t, f, tr_pc, cells, (hipStream_t)stream);
}

// Stage-level splat (pointcloud2voxels3d_fast): locate + the same slab kernel, raw grid only.
int dpc_splat_fwd(const DpcParams* p, const void* tr, int tr_is_f64, void* cells, float* vox, void* stream) {
  int rc = validate(p);
  if (rc != DPC_OK) return rc;
  if (!vox || (p->N > 0 && p->B > 0 && (!tr || !cells))) return DPC_ERR_NULL;
  if (p->B == 0) return DPC_OK;
  hipStream_t st = (hipStream_t)stream;
  if ((rc = launch_locate(p, tr_is_f64 ? 2 : 1, tr, nullptr, nullptr, nullptr, nullptr, cells, st)) != DPC_OK) return rc;
  const TapPlan none{0, 0, 0};
  return launch_splat<0>(p, cells_view(p, cells), nullptr, none, vox, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st);
}

}  // extern "C"

namespace {

// k_zcol_fwdbwd handles one pose candidate per sample, 32-, 64- or 128-deep columns with a bucketed z kernel; with two rays
// per lane its accesses are float2: an even image width and 8-byte aligned grids/images.
bool can_fuse_column_backward(const DpcParams* p, const TapPlan& pz, int K, const void* grid_wh, const void* proj,
                              const void* gt, const void* workspace) {
  const auto aligned8 = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 7u) == 0; };
  return K == 1 && gt != nullptr && p->B > 0 && pz.bucket >= 0 && (p->D == 32 || p->D == 64 || p->D == 128) && p->W % DPC_ZFB_RPL == 0 &&
         aligned8(grid_wh) && aligned8(proj) && aligned8(gt) && aligned8(workspace);
}

int project_fwd_impl(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f, const float* s,
                     const float* host_kern_xy, const float* host_kern_z, float* tr_pc, void* cells, float* raw,
                     float* grid_wh, float* smoothed, uint64_t* mask, float* proj, float* trans, const LossArgs& la,
                     void* bwd_workspace, float* bwd_dsmall, hipStream_t st) {
  int rc = validate(p);
  if (rc != DPC_OK) return rc;
  if (!q || !grid_wh || !mask || !proj) return DPC_ERR_NULL;
  if (p->N > 0 && p->B > 0 && (!pc || !cells)) return DPC_ERR_NULL;
  if ((p->taps_xy > 0 && !host_kern_xy) || (p->taps_z > 0 && !host_kern_z)) return DPC_ERR_NULL;
  if (p->B == 0) return DPC_OK;
  const TapPlan pxy = plan_taps(host_kern_xy, p->taps_xy), pz = plan_taps(host_kern_z, p->taps_z);
  if (pxy.bucket < 0) return DPC_ERR_TAPS;  // in-LDS passes need a radius bucket; caller composes the stage ops
  float* Tbuf = grid_wh;

  if ((rc = launch_locate(p, 0, pc, q, t, f, tr_pc, cells, st)) != DPC_OK) return rc;
  const Cells cv = cells_view(p, cells);
  // fused loss with one candidate per sample and a backward workspace: the ray-march kernel also runs the column
  // backward; its per-cloud sum-and-count words live behind the ds partials and are zeroed by the slab kernel
  const bool fuse_bwd = bwd_workspace != nullptr && bwd_dsmall != nullptr && la.loss_direct != nullptr &&
                        can_fuse_column_backward(p, pz, la.K, Tbuf, proj, la.gt, bwd_workspace);
  const size_t grid_bytes = (((size_t)p->B * p->D * p->H * p->W * sizeof(float) + 255) / 256) * 256;
  float* ds_part = fuse_bwd ? reinterpret_cast<float*>(static_cast<char*>(bwd_workspace) + grid_bytes) : nullptr;
  unsigned long long* tickets = fuse_bwd ? reinterpret_cast<unsigned long long*>(ds_part + (size_t)p->B * col_tiles(p)) : nullptr;
#define LAUNCH_SPLAT(RB) rc = launch_splat<RB>(p, cv, host_kern_xy, pxy, raw, Tbuf, mask, la.sse, la.loss_direct, la.winner_out, tickets, st)
  DPC_FOR_BUCKET(pxy.bucket, LAUNCH_SPLAT)
#undef LAUNCH_SPLAT
  if (rc != DPC_OK) return rc;

  dim3 gcol(col_tiles(p), p->B);
  const RayHost rh = ray_host(p);
  bool done = false;
  if (fuse_bwd) {
    float* dT = static_cast<float*>(bwd_workspace);
    const int ntile = col_tiles(p);
    constexpr int kRpl = DPC_ZFB_RPL;
    dim3 gpair((p->H * p->W / kRpl + kColThreads - 1) / kColThreads, p->B);
#define LAUNCH_ZFB(RB)                                                                                             \
  {                                                                                                                \
    const TapsT<RB> tzf = make_taps<RB>(host_kern_z, pz, false), tza = make_taps<RB>(host_kern_z, pz, true);       \
    if (p->D == 32) DPC_LAUNCH("k_zcol_fwdbwd", (k_zcol_fwdbwd<32, RB, kRpl>), gpair, dim3(kColThreads), 0, st, *p, rh, Tbuf, s, tzf, tza, proj, dT, ds_part, ntile, tickets, bwd_dsmall, la); \
    else if (p->D == 64) DPC_LAUNCH("k_zcol_fwdbwd", (k_zcol_fwdbwd<64, RB, kRpl>), gpair, dim3(kColThreads), 0, st, *p, rh, Tbuf, s, tzf, tza, proj, dT, ds_part, ntile, tickets, bwd_dsmall, la); \
    else DPC_LAUNCH("k_zcol_fwdbwd", (k_zcol_fwdbwd<128, RB, kRpl>), gpair, dim3(kColThreads), 0, st, *p, rh, Tbuf, s, tzf, tza, proj, dT, ds_part, ntile, tickets, bwd_dsmall, la); \
  }
    DPC_FOR_BUCKET(pz.bucket, LAUNCH_ZFB)
#undef LAUNCH_ZFB
    if (rc != DPC_OK) return rc;
    return launch_ok();
  }
#define LAUNCH_ZFWD(RB)                                                                                          \
  {                                                                                                              \
    const TapsT<RB> tz = make_taps<RB>(host_kern_z, pz, false);                                                  \
    if (p->D == 32) { DPC_LAUNCH("k_zcol_fwd", (k_zcol_fwd<32, RB>), gcol, dim3(kColThreads), 0, st, *p, rh, Tbuf, s, tz, smoothed, proj, trans, la); done = true; } \
    else if (p->D == 64) { DPC_LAUNCH("k_zcol_fwd", (k_zcol_fwd<64, RB>), gcol, dim3(kColThreads), 0, st, *p, rh, Tbuf, s, tz, smoothed, proj, trans, la); done = true; } \
    else if (p->D == 128) { DPC_LAUNCH("k_zcol_fwd", (k_zcol_fwd<128, RB>), gcol, dim3(kColThreads), 0, st, *p, rh, Tbuf, s, tz, smoothed, proj, trans, la); done = true; } \
  }
  if (pz.bucket >= 0) { DPC_FOR_BUCKET(pz.bucket, LAUNCH_ZFWD) }
#undef LAUNCH_ZFWD
  if (!done) {
    DPC_LAUNCH("k_zcol_fwd", k_zcol_fwd_dyn, gcol, dim3(kColThreads), 0, st, *p, rh, Tbuf, s,
               make_taps_dyn(host_kern_z, p->taps_z, false), smoothed, proj, trans, la);
  }
  return launch_ok();
}

int project_bwd_impl(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f, const float* s,
                     const float* host_kern_xy, const float* host_kern_z, const void* cells, const float* grid_wh,
                     const uint64_t* mask, const float* dproj, const float* proj, const float* trans, const LossArgs& la,
                     float* dpc, float* dsmall, void* workspace, hipStream_t st) {
  const bool column_done = la.scale_in_gather != 0;  // dT, ds partials and zeroed dsmall come from the forward
  int rc = validate(p);
  if (rc != DPC_OK) return rc;
  if (!q || !grid_wh || !mask || !dsmall || !workspace) return DPC_ERR_NULL;
  if (!column_done && (la.gt == nullptr ? !dproj : (!proj || !la.winner))) return DPC_ERR_NULL;
  if (p->N > 0 && p->B > 0 && (!pc || !cells || !dpc)) return DPC_ERR_NULL;
  if ((p->taps_xy > 0 && !host_kern_xy) || (p->taps_z > 0 && !host_kern_z)) return DPC_ERR_NULL;
  if (p->B == 0) return DPC_OK;
  const TapPlan pxy = plan_taps(host_kern_xy, p->taps_xy), pz = plan_taps(host_kern_z, p->taps_z);
  if (pxy.bucket < 0) return DPC_ERR_TAPS;
  float* dT = static_cast<float*>(workspace);
  const size_t grid_bytes = (((size_t)p->B * p->D * p->H * p->W * sizeof(float) + 255) / 256) * 256;
  float* ds_part = reinterpret_cast<float*>(static_cast<char*>(workspace) + grid_bytes);
  const int ntile = col_tiles(p);
  dim3 gcol(ntile, p->B);
  const RayHost rh = ray_host(p);

  bool done = column_done;
#define LAUNCH_ZBWD(RB)                                                                                           \
  {                                                                                                               \
    const TapsT<RB> tz = make_taps<RB>(host_kern_z, pz, true), tzf = make_taps<RB>(host_kern_z, pz, false);       \
    if (p->D == 32) { DPC_LAUNCH("k_zcol_bwd", (k_zcol_bwd<32, RB>), gcol, dim3(kColThreads), 0, st, *p, rh, grid_wh, s, dproj, proj, trans, tzf, tz, dT, ds_part, dsmall, la); done = true; } \
    else if (p->D == 64) { DPC_LAUNCH("k_zcol_bwd", (k_zcol_bwd<64, RB>), gcol, dim3(kColThreads), 0, st, *p, rh, grid_wh, s, dproj, proj, trans, tzf, tz, dT, ds_part, dsmall, la); done = true; } \
    else if (p->D == 128) { DPC_LAUNCH("k_zcol_bwd", (k_zcol_bwd<128, RB>), gcol, dim3(kColThreads), 0, st, *p, rh, grid_wh, s, dproj, proj, trans, tzf, tz, dT, ds_part, dsmall, la); done = true; } \
  }
  if (!done && pz.bucket >= 0) { DPC_FOR_BUCKET(pz.bucket, LAUNCH_ZBWD) }
#undef LAUNCH_ZBWD
  if (!done) {
    DPC_LAUNCH("k_zcol_bwd", k_zcol_bwd_dyn, gcol, dim3(kColThreads), 0, st, *p, rh, grid_wh, s, dproj, proj, trans,
               make_taps_dyn(host_kern_z, p->taps_z, false), make_taps_dyn(host_kern_z, p->taps_z, true), dT, ds_part,
               dsmall, la);
  }
  if ((rc = launch_ok()) != DPC_OK) return rc;

  const Cells cv = cells_view(p, cells);
#define LAUNCH_GATHER(RB) \
  rc = launch_gather<RB>(p, cv, pc, q, t, f, host_kern_xy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, la, st)
  DPC_FOR_BUCKET(pxy.bucket, LAUNCH_GATHER)
#undef LAUNCH_GATHER
  return rc;
}

const LossArgs kNoLoss{nullptr, nullptr, nullptr, nullptr, 1, 1.0f, nullptr, nullptr, 0};

}  // namespace

extern "C" {

int dpc_project_fwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                    const float* s, const float* host_kern_xy, const float* host_kern_z, float* tr_pc, void* cells,
                    float* raw, float* grid_wh, float* smoothed, uint64_t* mask, float* proj, float* trans, void* stream) {
  return project_fwd_impl(p, pc, q, t, f, s, host_kern_xy, host_kern_z, tr_pc, cells, raw, grid_wh, smoothed, mask, proj,
                          trans, kNoLoss, nullptr, nullptr, (hipStream_t)stream);
}

int dpc_project_bwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                    const float* s, const float* host_kern_xy, const float* host_kern_z, const void* cells,
                    const float* grid_wh, const uint64_t* mask, const float* trans, const float* dproj, float* dpc,
                    float* dsmall, void* workspace, void* stream) {
  if (!dproj) return DPC_ERR_NULL;
  return project_bwd_impl(p, pc, q, t, f, s, host_kern_xy, host_kern_z, cells, grid_wh, mask, dproj, nullptr, trans,
                          kNoLoss, dpc, dsmall, workspace, (hipStream_t)stream);
}

int dpc_project_loss_fwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                         const float* s, const float* host_kern_xy, const float* host_kern_z, const float* gt,
                         int num_candidates, float* tr_pc, void* cells, float* grid_wh, uint64_t* mask, float* proj,
                         float* trans, float* sse, float* loss, int32_t* winner, void* bwd_workspace, float* bwd_dsmall,
                         int* column_backward_done, void* stream) {
  if (!p || !gt || !sse || !loss || !winner) return DPC_ERR_NULL;
  if (num_candidates < 1 || p->B % num_candidates != 0) return DPC_ERR_SHAPE;
  const int S = p->B / num_candidates;
  const bool direct = num_candidates == 1;  // every cloud is its sample's winner: blocks add straight into the loss
  const LossArgs la{gt, sse, nullptr, nullptr, num_candidates, S > 0 ? 1.0f / (float)S : 0.f,
                    direct ? loss : nullptr, direct ? winner : nullptr, 0};
  // the forward can also run the column half of the backward when there is one candidate per sample (see k_zcol_fwdbwd)
  const TapPlan pz = plan_taps(host_kern_z, p->taps_z);
  const bool fuse = direct && bwd_workspace && bwd_dsmall &&
                    can_fuse_column_backward(p, pz, num_candidates, grid_wh, proj, gt, bwd_workspace);
  if (column_backward_done) *column_backward_done = fuse ? 1 : 0;
  int rc = project_fwd_impl(p, pc, q, t, f, s, host_kern_xy, host_kern_z, tr_pc, cells, nullptr, grid_wh, nullptr, mask,
                            proj, trans, la, fuse ? bwd_workspace : nullptr, fuse ? bwd_dsmall : nullptr,
                            (hipStream_t)stream);
  if (rc != DPC_OK || p->B == 0 || direct) return rc;
  DPC_LAUNCH("k_loss_finalize", k_loss_finalize, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)sse, S,
             num_candidates, la.inv_S, loss, winner);
  return launch_ok();
}

int dpc_project_loss_bwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                         const float* s, const float* host_kern_xy, const float* host_kern_z, const void* cells,
                         const float* grid_wh, const uint64_t* mask, const float* proj, const float* trans,
                         const float* gt, int num_candidates, const int32_t* winner, const float* dloss,
                         int column_backward_done, float* dpc, float* dsmall, void* workspace, void* stream) {
  if (!p || !gt || !winner || !proj) return DPC_ERR_NULL;
  if (num_candidates < 1 || p->B % num_candidates != 0) return DPC_ERR_SHAPE;
  const int S = p->B / num_candidates;
  const LossArgs la{gt, nullptr, winner, dloss, num_candidates, S > 0 ? 1.0f / (float)S : 0.f, nullptr, nullptr,
                    column_backward_done ? 1 : 0};
  return project_bwd_impl(p, pc, q, t, f, s, host_kern_xy, host_kern_z, cells, grid_wh, mask, nullptr, proj, trans, la,
                          dpc, dsmall, workspace, (hipStream_t)stream);
}

}  // extern "C"

// Shared by the kernel files of libdpc_render.so: layouts, LDS pass helpers, host-side launch plumbing.
//   dpc_slab_fwd.hip   k_locate, k_splat_hw            (transform + cell location, splat + W/H passes)
//   dpc_slab_xl.hip    k_splat_xl                      (the same slab work with x in the lanes, 64-wide grids)
//   dpc_column.hip     k_zcol_fwd / _bwd / _fwdbwd     (D pass, DRC ray march and its backward, loss finalize)
//   dpc_slab_bwd.hip   k_gather_hw                     (adjoint H/W passes, 8-corner gather, transform backward)
//   dpc_entry.hip      C ABI of the fused path
#pragma once
#include <math.h>
#include <string.h>

#include <atomic>

#include "dpc_common.h"
#include "dpc_profile.h"

#ifdef DPC_ABLATE
// Diagnostic builds only (-DDPC_ABLATE): per-translation-unit switches that cut work out of the kernels (timing
// experiments; results are then wrong) and per-phase timestamps.  Every kernel file instantiates its own setters with
// DPC_DEBUG_SETTERS(tag); dpc_entry.hip's dpc_debug_set_ablate / dpc_debug_set_stamps forward to all of them.
static __device__ int g_dpc_ablate = 0;
#define DPC_ABL(bit) (g_dpc_ablate & (1 << (bit)))
// 100 MHz s_memrealtime (comparable across CUs), thread 0 of every workgroup; 16 slots per block
static __device__ unsigned long long* g_dpc_stamps = nullptr;
#define DPC_STAMP(slot)                                                                                    \
  do {                                                                                                     \
    if (g_dpc_stamps != nullptr && threadIdx.x == 0)                                                       \
      g_dpc_stamps[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define DPC_DEBUG_SETTERS(tag)                                                                                          \
  extern "C" int dpc_debug_set_ablate_##tag(int v) {                                                                    \
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dpc_ablate), &v, sizeof(int)) == hipSuccess ? 0 : -5;                         \
  }                                                                                                                     \
  extern "C" int dpc_debug_set_stamps_##tag(void* p) {                                                                  \
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dpc_stamps), &p, sizeof(void*)) == hipSuccess ? 0 : -5;                       \
  }
#else
#define DPC_ABL(bit) 0
#define DPC_STAMP(slot) do { } while (0)
#define DPC_DEBUG_SETTERS(tag)
#endif

namespace dpck {

constexpr int kSlabThreads = 1024;
constexpr int kColThreads = 256;
constexpr int kNumCUs = 256;  // MI355X
// k_zcol_fwdbwd's sum-and-count words (one per cloud, one for the batch), laid out for the launch at hand:
//   [ poison | arrivals | squared error in fixed point ]
// arrivals: ray tiles of a cloud / clouds of the batch, `bits` wide; poison: the same width above it -- a contributor whose
// share is not a finite number inside the representable range adds 1 THERE instead of a sum (so a NaN scale, or a `gt`
// that is no mask, surfaces as a NaN loss like in the reference, instead of silently corrupting the arrival count);
// the sum field takes the rest.  A tile's share is at most 256 rays x 16 (|gt - proj| <= 4).
struct SseFormat {
  int frac;    // fractional bits of the sum field
  int cshift;  // first bit of the arrival field (= width of the sum field)
  int bits;    // width of the arrival field and of the poison field above it
};
constexpr float kTileSseCap = 4096.0f;
inline int bit_width(unsigned long long v) {
  int n = 0;
  while (v) { ++n; v >>= 1; }
  return n;
}
inline SseFormat sse_format(unsigned long long contributors, double max_share) {
  SseFormat f;
  f.bits = bit_width(contributors);                     // arrivals 0 .. contributors
  f.cshift = 64 - 2 * f.bits;
  const int int_bits = bit_width((unsigned long long)(contributors * max_share) + 1);
  f.frac = f.cshift - int_bits;
  if (f.frac > 30) f.frac = 30;
  if (f.frac < 0) f.frac = 0;
  return f;
}
__device__ inline unsigned long long sse_share(const SseFormat& f, double value, double cap) {
  const bool bad = !(value >= 0.0 && value <= cap);     // NaN fails both comparisons
  return (1ull << f.cshift) | (bad ? (1ull << (f.cshift + f.bits)) : (unsigned long long)(value * (double)(1ull << f.frac) + 0.5));
}
__device__ inline bool sse_complete(const SseFormat& f, unsigned long long before, int contributors) {
  return (int)((before >> f.cshift) & ((1ull << f.bits) - 1ull)) == contributors - 1;
}
__device__ inline double sse_total(const SseFormat& f, unsigned long long word) {  // NaN when any contributor poisoned it
  if ((word >> (f.cshift + f.bits)) != 0ull) return __longlong_as_double(0x7ff8000000000000ll);
  return (double)(word & ((1ull << f.cshift) - 1ull)) * (1.0 / (double)(1ull << f.frac));
}
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

// ------------------------------------------------------------------------------------------------------
// Workgroup -> (part, cloud).  Every grid of the path is (parts per cloud) x (clouds), launched 1-D.  The eight XCDs of
// the chip have private L2s and workgroups are dealt to them round-robin by linear id (MI355X_MICROARCH.md, "Workgroup
// dispatch"), so the map sends ALL workgroups of cloud c, in every kernel, to the XCD group c % 8: what one kernel of
// the chain writes for a cloud (point records, the W/H grid, dT: plain stores keep their lines in the writer's L2) is
// read by the next kernel from the same L2 instead of from HBM.  Placement is a speed matter only; nothing depends on it.
// ------------------------------------------------------------------------------------------------------
struct Blk {
  int x, y;    // part index (chunk / slab / ray tile), cloud
  int nx, ny;  // parts per cloud, clouds
};
// `group` > 1 (the backward with K pose candidates per sample, K = group): the candidates of sample s go to the XCD group
// s % 8 instead, so that the ONE winning cloud per sample that has work to do lands on every XCD equally often (winners
// share their candidate index more often than not, and cloud % 8 would pile them up on a few XCDs).
__device__ inline Blk block_coords(int clouds, int group = 1) {
  Blk k;
  k.ny = clouds;
  k.nx = (int)gridDim.x / clouds;
  const int L = blockIdx.x;
#ifndef DPC_NO_XCD_MAP
  if (group > 1 && clouds % (8 * group) == 0) {
    const int q = L >> 3, j = q / k.nx;  // j: running index of this XCD group's clouds
    k.y = ((L & 7) + 8 * (j / group)) * group + j % group;
    k.x = q % k.nx;
    return k;
  }
  if ((clouds & 7) == 0) {
    const int q = L >> 3;
    k.y = (L & 7) + 8 * (q / k.nx);
    k.x = q % k.nx;
    return k;
  }
#endif
  k.y = L / k.nx;
  k.x = L - k.y * k.nx;
  return k;
}

__host__ __device__ inline int points_per_set(const DpcParams& P) { return P.point_index != nullptr ? P.N_src : P.N; }

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
  const int incl = wave_inclusive_scan(r.cnt);   // wave 0, all 64 lanes (lanes without a chunk hold 0)
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

// Write-through stores (the sc1 bit) for the big streams a kernel writes and does not read again: T, dT.
// An ordinary store is acknowledged by the L2 and its line stays dirty there; the whole written volume -- a c2 grid is
// 33.5 MB, the eight L2s hold 32 MB -- is then flushed when the kernel ends, after the last wave: 3-7 us in which the CUs
// do nothing.  Written through, the stream goes to memory WHILE the kernel still computes.  Measured on the c2 step
// (same box, alternating builds, profiles/r03_ab_runs.txt): T -1.6..-2.2 us, dT -1.6..-2.2 us, together 55.9 -> 51.7 us.
// The non-temporal bit instead of sc1 gains the same in the writing kernel and loses it again in the reading one (the
// lines do not stay in the memory-side cache).  Relaxed agent-scope atomic stores are how the compiler is asked for
// "global_store ... sc1"; they are plain stores to memory otherwise (no ordering is implied or needed: the consumer is
// the next kernel).
__device__ inline void store_through(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void store_through(f32x2* p, f32x2 v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int kAuxThrough = 16;   // the same bit for the raw-buffer store builtins (cache policy operand: sc1)

#ifndef DPC_BWD_NARROW_PAD
#define DPC_BWD_NARROW_PAD 1
#endif
constexpr bool kBwdNarrowPad = DPC_BWD_NARROW_PAD != 0;   // 64-wide thick backward slabs at radius > 4: see BwdGeo
#ifndef DPC_BWD_ZS64
#define DPC_BWD_ZS64 8
#endif
constexpr int kBwdZs64 = DPC_BWD_ZS64;   // cell layers of the thick backward slab at G = 64 (the other thicknesses: further down)

// PADO >= 0 fixes the zero pad between rows (the backward's narrow layout, see BwdGeo); -1: as wide as the tap radius
template <int GS, int RB, int NT_, int LW_, int LH_, int PADO = -1>
struct SlabGeo {
  static constexpr int PAD = PADO >= 0 ? PADO : (RB <= 4 ? 4 : ((RB + 3) / 4) * 4);
  // NARROW: the pad is shorter than the radius, so a point's W window can reach into the neighbouring rows (or LEAD floats in
  // front of the first row / behind the last): the reader masks those taps itself and the LEAD floats are kept zero
  static constexpr bool NARROW = PAD < RB;
  static constexpr int LEAD = NARROW ? ((RB + 1 - PAD + 3) / 4) * 4 : 0;   // a window reaches RB in front of a row, RB + 1 behind it
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
  __host__ __device__ static constexpr size_t slab_floats(int planes) { return (size_t)planes * PLANE + PAD + 2 * LEAD; }
  __device__ static int at(int z, int y, int x) { return LEAD + (z * GS + y) * WP + PAD + x; }
  __device__ static int row_start(int row) { return LEAD + row * WP; }   // the row's pad; its first voxel PAD floats on
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
// 128-wide planes: 8-row H-pass segments -- a one-plane step of the rolling backward then has an item for every one of
// its 1024 threads (16-row segments left half of them idle under a latency-bound load: k_gather_hw<128,1,8> 41.9 -> 38.7 us)
// 64-wide planes, radius > 4: pads of 4 whatever the radius -- with pads as wide as the radius (8, 12 floats per row) a
// workgroup's LDS held 3 + 1 planes at best (8 + 1 need 166 / 175 KB), 22 slabs per cloud instead of 8; the gather masks the
// taps that would reach over a row's end instead (k_gather_hw, corner_row)
using BwdGeo = SlabGeo<GS, RB, bwd_threads(GS, NPL), 32, (GS == 128 ? 8 : 16), (GS == 64 && RB > 4 && NPL == kBwdZs64 + 1 && kBwdNarrowPad ? 4 : -1)>;

constexpr float kFixScale = 17592186044416.0f;          // 2^44: splat weights accumulate as 64-bit fixed point
constexpr float kFixInv = 1.0f / 17592186044416.0f;
constexpr unsigned long long kFixOne = 1ull << 44;
static_assert((unsigned long long)DPC_MAX_POINTS <= ~0ull / kFixOne, "DPC_MAX_POINTS unit weights must fit the 64-bit accumulator");

// Conversions between fp32 and the 64-bit fixed point, written on the two 32-bit halves.  Left as `(unsigned long long)(w *
// 2^44)` and `(float)(a >> 32) ...` the compiler expands generic 64-bit <-> float conversions (it widens the halves back to
// i64 first): a dozen instructions each, in kernels that are bound by instruction issue.
//   weight -> fixed point: trunc(w 2^44) = trunc(w 2^12) 2^32 + trunc((w 2^12 - trunc(w 2^12)) 2^32); every step exact
__device__ inline unsigned long long to_fixed(float w) {
  const float scaled = w * 4096.0f;                         // w 2^12, exact
  const unsigned int hi = (unsigned int)scaled;             // v_cvt_u32_f32 truncates
  const float rem = scaled - (float)hi;                     // exact: both are multiples of ulp(scaled), 0 <= rem < 1
  const unsigned int lo = (unsigned int)(rem * 4294967296.0f);
  return ((unsigned long long)hi << 32) | lo;
}
//   fixed point -> float: a < 2^56: hi < 2^24 converts exactly, lo rounds once, the fma rounds once more (<= 1 ulp overall)
__device__ inline float from_fixed(unsigned long long a) {
  unsigned int hi = (unsigned int)(a >> 32), lo = (unsigned int)a;
  asm("" : "+v"(hi), "+v"(lo));                             // opaque: keep the halves 32-bit (two v_cvt_f32_u32)
  return fmaf((float)hi, 0x1p-12f, (float)lo * kFixInv);
}

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
__device__ inline void hpass_fast(const float* slab, const TapsT<RB>& taps, Store store, int tid) {   // tid: tid (a loop around the call may pass it opaque)
  constexpr int ITEMS = NPL * Geo::XP * Geo::NSEGH, IPT = (ITEMS + Geo::NT - 1) / Geo::NT;
  f32x2 v[IPT][Geo::HWIN];
#pragma unroll
  for (int it = 0; it < IPT; ++it) {
    const int item = tid + it * Geo::NT;
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
    const int item = tid + it * Geo::NT;
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
  // (64^2: 8 + 1 planes are 1152 items for 1024 threads, i.e. a second item for the first 128 threads.  Giving every thread a
  // two-row sliver of the ninth plane instead, its loads in flight together with the main item's, changed nothing: the pass
  // is bound by the bandwidth of the dT read, 6.2-6.5 us in every slab -- profiles/r03_overlap_experiments.txt item 6.)
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
        for (int i = 0; i < 2 * RB + 1; ++i) {
          const int tp = tap_edge_first<RB>(i);   // an adjoint pass: edges first, centre last (dpc_common.h)
          acc = __builtin_elementwise_fma(f32x2{taps.w[tp], taps.w[tp]}, v[j + tp], acc);
        }
        store(z, seg * Geo::LH + j, 2 * xp, acc);
      }
    }
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
  float* sse_tiles;     // forward, unfused ray march: [B, tiles] per-tile squared errors, summed in tile order by k_loss_finalize
  // backward of the one-call step with K > 1 (dpc_project_loss_step): the min-of-K selection is made HERE instead of by a
  // launch of its own -- every workgroup of k_zcol_bwd picks its sample's winner from sse_tiles, workgroup 0 also writes
  // sse, winner and the loss with k_loss_finalize's own code (same bits)
  int* winner_write;    // [S] | nullptr
  float* loss_write;    // [1]
  int ntile;            // ray tiles per cloud in sse_tiles
};

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
inline TapPlan plan_taps(const float* k, int taps) {
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
  static const int buckets[] = {0, 1, 2, 3, 4, 6, 8, 10, 15};   // 8: sigma = 0.01 world units on a 128^3 grid (sigma_rel 1.28)
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

inline TapsDyn make_taps_dyn(const float* k, int taps, bool flip) {
  TapsDyn t;
  t.n = taps;
  for (int i = 0; i < DPC_MAX_TAPS; ++i) t.w[i] = 0.f;
  for (int i = 0; i < taps; ++i) t.w[i] = k[flip ? taps - 1 - i : i];
  return t;
}

inline int validate(const DpcParams* p) {
  if (p == nullptr) return DPC_ERR_NULL;
  if (p->B < 0 || p->N < 0 || p->D < 1 || p->H < 1 || p->W < 1) return DPC_ERR_SHAPE;
  if (p->D > 1024 || p->H > 1024 || p->W > 1024 || p->B > 65535) return DPC_ERR_SHAPE;  // 10-bit cell indices
  if (p->N > DPC_MAX_POINTS) return DPC_ERR_SHAPE;   // one voxel's fixed-point sum (kFixOne = 2^44 per unit weight) must not wrap
  if (p->point_replicas < 0 || (p->point_replicas > 1 && p->B % p->point_replicas != 0)) return DPC_ERR_SHAPE;
  if (p->point_index != nullptr && p->N_src < 1) return DPC_ERR_SHAPE;
  if ((p->dev_taps_xy != nullptr && p->taps_xy < 1) || (p->dev_taps_z != nullptr && p->taps_z < 1)) return DPC_ERR_TAPS;
  for (int taps : {p->taps_xy, p->taps_z})
    if (taps < 0 || taps > DPC_MAX_TAPS || (taps > 0 && taps % 2 == 0)) return DPC_ERR_TAPS;
  return DPC_OK;
}

// planes of an H x W slab that fit the LDS tile of the generic kernels
inline int planes_fit(const DpcParams* p) { return kLdsBudget / ((p->H * (p->W | 1)) * (int)sizeof(float)); }
inline int slab_threads(const DpcParams* p) { return (long long)p->H * p->W >= 2048 ? kSlabThreads : 256; }
inline int col_tiles(const DpcParams* p) { return (p->H * p->W + kColThreads - 1) / kColThreads; }

// Raise a kernel's dynamic-LDS limit.  hipFuncSetAttribute is a driver call, so it is made once per kernel, device and
// size instead of on every launch: every launcher keeps, per device, the largest size it has set (relaxed atomics: two
// threads racing on the first launch both make the call, which is idempotent).  `slot` is a static of the launcher's
// template instantiation = one per kernel.
struct LdsLimit {
  std::atomic<size_t> set[16];
};
template <class K>
int set_lds(K kernel, size_t bytes, LdsLimit& slot) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return DPC_ERR_LAUNCH;
  const bool tracked = dev >= 0 && dev < 16;
  if (tracked && slot.set[dev].load(std::memory_order_relaxed) >= bytes) return DPC_OK;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return DPC_ERR_LAUNCH;
  if (tracked) slot.set[dev].store(bytes, std::memory_order_relaxed);
  return DPC_OK;
}

inline RayHost ray_host(const DpcParams* p) {
  const double eps = (double)p->clip_val;
  return RayHost{p->clip_val, (float)(1.0 - eps), (float)expm1(eps)};
}

inline int launch_ok() { return hipGetLastError() == hipSuccess ? DPC_OK : DPC_ERR_LAUNCH; }

inline Cells cells_view(const DpcParams* p, const void* cells) {
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
    case 8: MACRO(8); break;          \
    case 10: MACRO(10); break;        \
    case 15: MACRO(15); break;        \
    default: rc = DPC_ERR_TAPS;       \
  }

#ifndef DPC_FWD_ZS64
#define DPC_FWD_ZS64 4
#endif
constexpr int kFwdZs64 = DPC_FWD_ZS64;
// Cell layers per backward slab at G = 64.  THICK: 8 + 1 planes fill the LDS, one 1024-thread workgroup per CU, 8 slabs per cloud:
// the best shape whenever (clouds with backward work) x 8 fills the chip -- 32 clouds are exactly one round of 256 workgroups
// (8: 17.0-17.7, 4: 22.8-23.1, 3: 21.2-21.3, 2: 25.8-26.4 us at c2, radius 3).  At radius 5..10 the thick slab fits because
// its rows carry pads of 4, not of the radius (BwdGeo's narrow layout: k_gather_hw masks the taps that reach over a row's end);
// with pads as wide as the radius 3 + 1 planes was the best that fit (22 slabs per cloud, two workgroups per CU):
// sigma_rel 0.9 / 1.2 / 3.0: 23.0 / 25.9 / 27.1 -> 19.2 / 21.7 / 24.1 us (profiles/r04_ab/narrow_pad_*.jsonl).
// THIN: when the thick slabs would leave CUs empty (c5: 16 winning clouds of 128 = 128 workgroups) thinner ones win:
// radius 3, 16 clouds: 4 layers 13.6 us against 8: 15.5 (3: 14.7, 2: 15.4); radius 10, 16 clouds: 3 layers with wide pads 20.0
// against 8 narrow 23.8.  launch_gather_rb picks per call.
#ifndef DPC_BWD_ZS64_THIN
#define DPC_BWD_ZS64_THIN 4
#endif
#ifndef DPC_BWD_ZS64_WIDE
#define DPC_BWD_ZS64_WIDE 3
#endif
// kBwdZs64 (thick slab, every radius; narrow pads beyond radius 4): defined next to BwdGeo
constexpr int kBwdZs64Thin = DPC_BWD_ZS64_THIN;  // few working clouds, radius <= 4
constexpr int kBwdZs64Wide = DPC_BWD_ZS64_WIDE;  // few working clouds, radius 5..10 (pads as wide as the radius)

// losing pose candidates of the fused min-of-K loss do no backward work
__device__ inline bool cloud_loses(const LossArgs& la, int b) {
  return la.gt != nullptr && la.winner[b / la.K] != b % la.K;
}

// K > 1 pose candidates per sample: only the winning cloud of a sample has any backward work.  The backward grids are
// then (parts per cloud) x (SAMPLES) and a workgroup looks its cloud up -- launching a (1024-thread, 157 KB LDS)
// workgroup per losing cloud just to let it return kept the winners' workgroups waiting for a CU (c5: 26 -> 15 us).
__host__ __device__ inline bool winners_only(const LossArgs& la) {
  return la.gt != nullptr && (la.winner != nullptr || la.winner_write != nullptr) && la.K > 1;
}

// ------------------------------------------------------------------------------------------------------
// Backward workspace (dpc_workspace_bytes): [dT grid][ds partials B x ntile][sum-and-count words B + 1]
//                                           [camera-gradient partials B x D x 16 doubles][arrival counters B]
//                                           [64-bit fixed-point d(point set) sums, shared point sets only]
// ------------------------------------------------------------------------------------------------------
struct Workspace {
  float* dT;
  float* ds_part;
  unsigned long long* tickets;   // k_zcol_fwdbwd: per-cloud squared-error words + the batch word
  double* cg_part;               // k_gather_hw: per-slab sums of the 13 camera-gradient accumulators
  unsigned int* cg_count;        // k_gather_hw: slabs of a cloud that have published (zeroed by the column kernels)
  unsigned long long* dpc_fixed; // [B/R, N_set, 3]: where several clouds add into one point set's gradient they add 64-bit
                                 // fixed point (2^-40): integer adds commute, so the sum is the same bits whatever order the
                                 // clouds arrive in (float atomics made the last bits vary from run to run); nullptr otherwise
};

// d(point) contributions of clouds that share a point set, as 64-bit fixed point: |sum| < 2^23, resolution 2^-40 (9e-13)
constexpr double kGradFixScale = 1099511627776.0;          // 2^40
constexpr double kGradFixInv = 1.0 / 1099511627776.0;
__device__ inline unsigned long long grad_to_fixed(float v) { return (unsigned long long)(long long)__double2ll_rn((double)v * kGradFixScale); }
// A contribution that is not a finite number below this bound (a diverged network: NaN scale, Inf gradient) has no integer to
// stand for it -- the conversion would turn it into a healthy-looking value.  It is not added; the point set's POISON word
// (one per set, behind the fixed-point sums) is set instead and k_fixed_to_dpc writes NaN for that set's gradient, so the
// divergence reaches the decoder as it does through the reference's float sums (where NaN + x = NaN).
constexpr float kGradFixMax = 1048576.0f;                   // 2^20 per contribution
__device__ inline bool grad_fits_fixed(float x, float y, float z) {
  return fabsf(x) < kGradFixMax && fabsf(y) < kGradFixMax && fabsf(z) < kGradFixMax;   // NaN fails every comparison
}
inline size_t ws_round(size_t n) { return (n + 255) / 256 * 256; }
inline size_t ws_grid_bytes(const DpcParams* p) { return ws_round((size_t)p->B * p->D * p->H * p->W * sizeof(float)); }
inline size_t ws_parts_bytes(const DpcParams* p) {
  return ws_round((size_t)p->B * col_tiles(p) * sizeof(float) + ((size_t)p->B + 1) * 8 + 8);
}
inline size_t ws_camgrad_bytes(const DpcParams* p) { return ws_round((size_t)p->B * p->D * 16 * sizeof(double)); }
inline bool shares_points(const DpcParams* p) { return p->point_replicas > 1 || p->point_index != nullptr; }
inline size_t ws_dpcfix_bytes(const DpcParams* p) {   // the sums, then one poison word per point set
  const int reps = p->point_replicas > 1 ? p->point_replicas : 1;
  const size_t sets = (size_t)(p->B / reps);
  return shares_points(p) ? ws_round(sets * points_per_set(*p) * 3 * sizeof(unsigned long long) + sets * sizeof(unsigned int)) : 0;
}
inline size_t ws_total_bytes(const DpcParams* p) {
  return ws_grid_bytes(p) + ws_parts_bytes(p) + ws_camgrad_bytes(p) + ws_round((size_t)p->B * sizeof(unsigned int)) + ws_dpcfix_bytes(p);
}
inline Workspace workspace_view(const DpcParams* p, void* ws) {
  Workspace w;
  char* base = static_cast<char*>(ws);
  w.dT = reinterpret_cast<float*>(base);
  w.ds_part = reinterpret_cast<float*>(base + ws_grid_bytes(p));
  w.tickets = reinterpret_cast<unsigned long long*>(
      (reinterpret_cast<uintptr_t>(w.ds_part + (size_t)p->B * col_tiles(p)) + 7u) & ~(uintptr_t)7u);
  w.cg_part = reinterpret_cast<double*>(base + ws_grid_bytes(p) + ws_parts_bytes(p));
  w.cg_count = reinterpret_cast<unsigned int*>(base + ws_grid_bytes(p) + ws_parts_bytes(p) + ws_camgrad_bytes(p));
  w.dpc_fixed = shares_points(p) ? reinterpret_cast<unsigned long long*>(base + ws_grid_bytes(p) + ws_parts_bytes(p) +
                                                                          ws_camgrad_bytes(p) + ws_round((size_t)p->B * sizeof(unsigned int)))
                                 : nullptr;
  return w;
}

// ------------------------------------------------------------------------------------------------------
// Camera gradient of a cloud (dq, dt, df): 13 sums over the cloud's points, taken in a FIXED order so that the result
// is the same bits on every run (float atomics between the slab workgroups made dq/dt/df vary in the last places), in fp64
// from the wave totals on.
//   block_sum13_fixed  per wave a DPP butterfly (fp32, a fixed tree), then the wave totals in wave order in fp64 (dpc_common.h)
//   camgrad_publish    the slab workgroups of a cloud hand their sums over INSIDE the launch: write-through (sc1) stores
//                      of the 13 doubles, the storing wave's vmcnt(0), one agent-scope ticket add; the workgroup that
//                      drew the last ticket reads all slabs' sums with sc1 loads in slab order, turns the moment matrix
//                      into d(q) in fp64 and writes dq/dt/df.  This is the hand-off form MI355X_MICROARCH.md lists as
//                      measured valid on gfx950 ("Valid forms", first table row: one lane's agent-scope add as the signal,
//                      the last adder told by the returned value, all handed-off bytes stored and loaded sc1); no L2
//                      write-back, no acquire.  The counter is left at zero again for the next backward.
// ------------------------------------------------------------------------------------------------------
// tot: this slab's sum of accumulator `tid` (threads 0..12, from block_sum13_fixed).  Called by the whole first wave.
__device__ inline void camgrad_publish(double tot, int tid, const CameraRaw& raw, int B, int b, int slab, int nslab,
                                       double* __restrict__ cg_part, unsigned int* __restrict__ cg_count,
                                       float* __restrict__ dsmall, bool has_t, bool has_f) {
  if (tid >= DPC_WAVE) return;
  unsigned long long* mine = reinterpret_cast<unsigned long long*>(cg_part + ((size_t)b * nslab + slab) * 16);
  if (tid < 13) __hip_atomic_store(mine + tid, (unsigned long long)__double_as_longlong(tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left this wave before the ticket is drawn
  unsigned int ticket = 0;
  if (tid == 0) ticket = __hip_atomic_fetch_add(cg_count + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
  if (ticket != (unsigned int)(nslab - 1)) return;   // somebody else arrives later and does the sum
  double sum = 0.0;
  if (tid < 13) {
    const unsigned long long* all = reinterpret_cast<const unsigned long long*>(cg_part + (size_t)b * nslab * 16);
    for (int k0 = 0; k0 < nslab; k0 += 8) {   // slab order: the same sum on every run; eight loads in flight at a time
      unsigned long long raw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        raw[u] = __hip_atomic_load(all + (size_t)min(k0 + u, nslab - 1) * 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u < nslab) sum += __longlong_as_double((long long)raw[u]);
    }
  }
  double m[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) m[i] = __shfl(sum, i, DPC_WAVE);
  if (tid == 0) {
    double dq[4];
    quaternion_grad_f64(raw.q, m, dq);
    float* dqb = dsmall + (size_t)DPC_COL_DQ * B + (size_t)b * 4;   // dq as a [B,4] block, dt [B,3], df [B,1] (dpc_render.h)
#pragma unroll
    for (int i = 0; i < 4; ++i) dqb[i] = (float)dq[i];
    if (has_t)
      for (int i = 0; i < 3; ++i) dsmall[(size_t)DPC_COL_DT * B + (size_t)b * 3 + i] = (float)m[9 + i];
    if (has_f) dsmall[(size_t)DPC_COL_DF * B + b] = (float)m[12];
    __hip_atomic_store(cg_count + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next backward
  }
}

// ---- launchers defined next to their kernels (bucket = compile-time tap radius bucket chosen by plan_taps) ----
int launch_locate(const DpcParams* p, int src, const void* pts, const float* q, const float* t, const float* f, float* tr_pc,
                  void* cells, hipStream_t st);
int launch_splat(int bucket, const DpcParams* p, Cells cells, const float* kxy, const TapPlan& pxy, float* raw, float* Tbuf,
                 uint64_t* mask, float* sse, float* loss_zero, int* winner_zero, unsigned long long* ticket_zero, hipStream_t st);
int launch_gather(int bucket, const DpcParams* p, Cells cells, const float* pc, const float* q, const float* t, const float* f,
                  const float* kxy, const TapPlan& pxy, const float* dT, const uint64_t* mask, const float* ds_part, int ntile,
                  float* dpc, float* dsmall, double* cg_part, unsigned int* cg_count, const LossArgs& la, hipStream_t st,
                  unsigned long long* dpc_fixed = nullptr);
int launch_zcol_fwd(const DpcParams* p, const float* host_kern_z, const TapPlan& pz, const float* Tbuf, const float* s,
                    float* smoothed, float* proj, float* trans, const LossArgs& la, hipStream_t st);
int launch_zcol_fwdbwd(const DpcParams* p, const float* host_kern_z, const TapPlan& pz, const float* Tbuf, const float* s,
                       float* proj, float* dT, float* ds_part, int ntile, unsigned long long* tickets, float* bwd_dsmall,
                       unsigned int* cg_count, const LossArgs& la, hipStream_t st);
int launch_zcol_bwd(const DpcParams* p, const float* host_kern_z, const TapPlan& pz, const float* grid_wh, const float* s,
                    const float* dproj, const float* proj, const float* trans, float* dT, float* ds_part, float* dsmall,
                    unsigned int* cg_count, const float* dgrid_extra, const LossArgs& la, hipStream_t st);
int launch_loss_finalize(const float* sse_tiles, int ntile, float* sse, int S, int K, float inv_S, float* loss, int32_t* winner,
                         hipStream_t st);
// dpc_slab_xl.hip: the x-in-lanes slab kernels (64 x 64 planes, radius bucket 1..6); DPC_NO_XL builds keep the older kernels
bool xl_applies(const DpcParams* p, int bucket);
int launch_splat_xl(int bucket, const DpcParams* p, Cells cells, const float* kxy, const TapPlan& pxy, float* Tbuf, uint64_t* mask,
                    float* sse, float* loss_zero, int* winner_zero, unsigned long long* ticket_zero, hipStream_t st);

}  // namespace dpck

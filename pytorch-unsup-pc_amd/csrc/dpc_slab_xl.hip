// Forward slab kernel for 64-wide grids with x in the lanes (k_splat_xl).  Design notes: DESIGN.md section 4.
//
// A grid row of 64 voxels is exactly one wavefront, so a thread that owns a run of y at ONE x gets
//   * the H pass in registers (a window along y, read with lanes on consecutive x: coalesced from global, conflict-free
//     from LDS),
//   * the W pass across the lanes with whole-wave DPP shifts (wave_shr:1 / wave_shl:1, zero fill = the zero padding of
//     the correlation at both ends of the row) -- no LDS round trip and no barrier between the two passes,
//   * the clamp mask for free: the 64 mask bits of a grid row ARE a 64-bit lane mask (forward: the SGPR pair a v_cmp
//     writes; backward: a scalar load used as the select mask of a v_cndmask).
// The forward needs no fp32 copy of the slab at all (accumulators -> registers -> global).  (A backward on the same plan --
// d(raw) fully filtered in LDS, 8 reads per gathered point -- measured slower than k_gather_hw in round 2: its dT load phase
// is 4-byte loads at one x per lane against 8-byte column pairs; it was removed in round 3, git history has it.)
#include <type_traits>

#include "dpc_kernels.h"

DPC_DEBUG_SETTERS(xl)

namespace dpck {
namespace {

constexpr int kXG = 64;     // H = W = 64: one wave per grid row
#ifndef DPC_XL_FWD_SEG
#define DPC_XL_FWD_SEG 16
#endif
constexpr int kXSeg = DPC_XL_FWD_SEG;   // y outputs per thread (16: ZS x 256 threads per workgroup; 8: ZS x 512)
constexpr int kXWavesPerPlane = kXG / kXSeg;

template <int CTRL>
__device__ inline float dpp_zero_fill(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ inline float from_lane_below(float v) { return dpp_zero_fill<0x138>(v); }  // wave_shr:1, lane x reads lane x-1
__device__ inline float from_lane_above(float v) { return dpp_zero_fill<0x130>(v); }  // wave_shl:1, lane x reads lane x+1

// out(x) = sum_t w[t] in(x + t - RB) over the 64 lanes of the wave, zero outside the row
template <int RB>
__device__ inline float wpass_lanes(float v, const TapsT<RB>& taps) {
  float acc = taps.w[RB] * v;
  float lo = v, hi = v;
#pragma unroll
  for (int k = 1; k <= RB; ++k) {
    lo = from_lane_below(lo);  // in(x - k)
    hi = from_lane_above(hi);  // in(x + k)
    acc = fmaf(taps.w[RB - k], lo, acc);
    acc = fmaf(taps.w[RB + k], hi, acc);
  }
  return acc;
}

// The same for a SYMMETRIC kernel (every Gaussian is, bit for bit: exp(-x^2 / 2 sigma^2) of +-x), as a Horner scheme over the
// lanes:  out = w_0 v + shr(P_1 + shr(P_2 + ... shr(P_RB))) + shl(P_1 + shl(P_2 + ... shl(P_RB))),  P_k = w_k v.
// The products are shared by the two sides and every shifted value is used exactly once, so the shift rides on the add as its
// DPP operand: 3 RB + 1 instructions per output instead of 4 RB + 1 (RB = 3: 10 against 13).
template <int RB>
__device__ inline float wpass_lanes_sym(float v, const TapsT<RB>& taps) {
#pragma clang fp contract(off)   // the products stay products (shared by both sides); the adds take the shifts as DPP operands
  float p[RB + 1];
#pragma unroll
  for (int k = 1; k <= RB; ++k) p[k] = taps.w[RB + k] * v;
  float l = p[RB], r = p[RB];
#pragma unroll
  for (int k = RB - 1; k >= 1; --k) {
    l = p[k] + from_lane_below(l);
    r = p[k] + from_lane_above(r);
  }
  float acc = taps.w[RB] * v;
  acc += from_lane_below(l);
  acc += from_lane_above(r);
  return acc;
}

// ------------------------------------------------------------------------------------------------------
// Forward: splat into 64-bit fixed-point accumulators (as k_splat_hw), then per thread: a y window of accumulators at
// one x -> clamp mask words by ballot -> H pass in registers -> W pass across lanes -> T.     grid (D/ZS) x B, ZS*256 threads
//   accumulator planes carry RB zero rows above and below (the zero padding of the H pass)
// ------------------------------------------------------------------------------------------------------
template <int ZS, int RB>
__global__ __launch_bounds__(ZS * kXWavesPerPlane * 64) void k_splat_xl(DpcParams P, Cells cells, TapsT<RB> taps_arg, int nround,
                                                       float* __restrict__ Tbuf,
                                                       uint64_t* __restrict__ mask, float* __restrict__ sse,
                                                       float* __restrict__ loss_zero, int* __restrict__ winner_zero,
                                                       unsigned long long* __restrict__ ticket_zero, int symmetric) {
  extern __shared__ __attribute__((aligned(16))) float slab[];
  constexpr int NT = ZS * kXWavesPerPlane * 64, PR = kXG + 2 * RB, ACC = ZS * PR * kXG, WIN = kXSeg + 2 * RB;
  static_assert(ACC % 4 == 0, "zero fill in 16-byte words, two halves");
  const TapsT<RB> taps = resolve_taps<RB>(taps_arg, P.dev_taps_xy, P.taps_xy, false);
  const Blk bk = block_coords(P.B);
  if (sse != nullptr && bk.x == 0 && threadIdx.x == 0) {  // the ray-march kernel accumulates into these
    sse[bk.y] = 0.f;
    if (winner_zero != nullptr) winner_zero[bk.y] = 0;
    if (ticket_zero != nullptr) ticket_zero[bk.y] = 0ull;
    if (ticket_zero != nullptr && bk.y == 0) ticket_zero[bk.ny] = 0ull;
    if (loss_zero != nullptr && bk.y == 0) *loss_zero = 0.f;
  }
  const int D = P.D;
  const int b = bk.y;
  const int tid = threadIdx.x;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(slab);
  f32x4* s4 = reinterpret_cast<f32x4*>(slab);
  int* tab = reinterpret_cast<int*>(acc + ACC);
  const bool flat = cells.nblk <= DPC_WAVE;
  // wave = (plane zz, rows y0 .. y0+15), lane = x
  const int lane = tid & (DPC_WAVE - 1);
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int zz = w % ZS, y0 = (w / ZS) * kXSeg;
  // The workgroup STAYS for `nround` slabs of its cloud (slabs bk.x, bk.x + nx, ...: the launcher picks nround so that the
  // grid is still one workgroup per CU).  A workgroup that ends is only replaced when its stores have drained -- with every
  // CU storing its planes at the same moment that is 3 us of an empty CU per round (stamps, profiles/r03_overlap_experiments
  // item 6); a workgroup that goes on zero-fills and scatters the next slab under the stores of this one.
  constexpr int PRE = 2, ZH = (ACC / 2) / 2;
  const int slab0 = bk.x, slab_step = bk.nx;
  const int nslab = (D + ZS - 1) / ZS;
  // offsets | zero A | table | barrier | records -> registers | zero B | barrier | atomics  (see k_splat_hw)
  // (Holding a slab's stores back in registers until the next slab's record loads are out -- a wave's vector memory
  // operations complete in order, so those loads are waited for together with the stores in front of them -- measured
  // slower: 21.4 against 19.3 us; the write stream is the scarce resource and wants to start as early as it can.)
  DPC_STAMP(0);
  PointRec pre[PRE];
  int npre = 0;
  auto open_slab = [&](int sl) {   // record offsets on their way, first half of the accumulators zeroed
    const int z0 = sl * ZS;
    RecordRange rr{0, 0};
    if (flat) rr = load_record_range(cells, b, max(z0 - 1, 0), z0 + min(ZS, D - z0));
#ifdef DPC_ABLATE
    if (!DPC_ABL(3))
#endif
    for (int i = tid; i < ZH; i += NT) s4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    return rr;
  };
  auto request_records = [&](const RecordRange& rr) {   // table, then every thread's first records into registers
    if (flat) finish_record_table(rr, tab);
    __syncthreads();
    npre = 0;
    if (flat) {
      const int total = tab[DPC_WAVE];
#pragma unroll
      for (int r = 0; r < PRE; ++r) {
        const int j = tid + r * NT;
        pre[r].code = -1; pre[r].tz = pre[r].ty = pre[r].tx = 0.f;
        if (j < total) {
          int c, pos;
          flat_lookup(tab, j, c, pos);
          pre[r] = load_record(cells.recs(b, c), pos);
        }
      }
      npre = PRE * NT;
    }
  };
  auto zero_rest = [&]() {
#ifdef DPC_ABLATE
    if (!DPC_ABL(3))
#endif
    for (int i = ZH + tid; i < ACC / 2; i += NT) s4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
  };
  if (slab0 < nslab) {
    const RecordRange rr = open_slab(slab0);
    request_records(rr);
    zero_rest();
  }
  DPC_STAMP(1);
  for (int round = 0; round < nround; ++round) {
    const int sl = slab0 + round * slab_step;
    if (sl >= nslab) break;   // block-uniform
    const int z0 = sl * ZS;
    const int nz = min(ZS, D - z0);
    auto scatter = [&](const PointRec& rec, const int4*) {
      const Cell c = cell_from_record(rec);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int pz = c.iz + k - z0;
        if (pz < 0 || pz >= nz) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (c.iy + j >= kXG) continue;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (c.ix + e >= kXG) continue;
            const float wt = c.wz[k] * c.wy[j] * c.wx[e];
#ifdef DPC_ABLATE
            if (!DPC_ABL(4) || wt == 123.456f)
#endif
            atomicAdd(&acc[(pz * PR + RB + c.iy + j) * kXG + c.ix + e], to_fixed(wt));  // ds_add_u64
          }
        }
      }
    };
    if (flat) {
#pragma unroll
      for (int r = 0; r < PRE; ++r)
        if (pre[r].code >= 0) scatter(pre[r], nullptr);
      for_each_record_flat(cells, b, tab, scatter, npre);
    } else {
      for_each_record(cells, b, max(z0 - 1, 0), z0 + nz, scatter);
    }
    __syncthreads();
    if (round == 0) DPC_STAMP(2);

    const bool active = zz < nz;   // wave-uniform: a last slab of fewer planes leaves waves without one
    const unsigned long long* col = acc + ((size_t)zz * PR + y0) * kXG + lane;  // window row i = grid row y0 + i - RB
    unsigned long long a[WIN];
#pragma unroll
    for (int i = 0; i < WIN; ++i) a[i] = col[i * kXG];
    // once every thread holds its window the accumulators are free again: open the next slab under the passes below
    const int sl_next = sl + slab_step;
    const bool more = round + 1 < nround && sl_next < nslab;
    RecordRange rr_next{0, 0};
    if (more) {
      __syncthreads();
      rr_next = open_slab(sl_next);
    }
    if (active) {
      // clamp mask: bit x of word y <=> raw <= 1 (raw >= 0 always); 64 lanes = the 64 bits of the row's word.  Lane j keeps
      // row j's word: the 16 words of this wave leave in one 128-byte store.
      unsigned long long* mask_out = reinterpret_cast<unsigned long long*>(mask) + ((size_t)b * D + z0 + zz) * kXG + y0;
      unsigned long long mword = 0ull;
#pragma unroll
      for (int j = 0; j < kXSeg; ++j) {
        const unsigned long long bits = __ballot(a[RB + j] <= kFixOne);
        mword = lane == j ? bits : mword;
      }
      if (lane < kXSeg) mask_out[lane] = mword;
      float v[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i) v[i] = fminf(from_fixed(a[i]), 1.0f);
      if (round == 0) DPC_STAMP(3);
      // four rows at a time: H pass in registers, W pass across the lanes, four row stores
      float* Tout = Tbuf + (((size_t)b * D + z0 + zz) * kXG + y0) * kXG + lane;
      // `symmetric` (block-uniform; set by the launcher for a kernel that equals its mirror image bit for bit and whose values
      // travel with the launch): the W pass takes the Horner form of wpass_lanes_sym, 10 instead of 13 instructions per output
      // at radius 3 (k_splat_xl 15.0 -> 14.75 us at c2, 52.1 -> 49.3 us for the 128 clouds of c5)
      auto rows = [&](auto sym_tag) {
        constexpr bool kSym = decltype(sym_tag)::value;
#pragma unroll
        for (int j = 0; j < kXSeg; j += 4) {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float h = 0.f;
#ifdef DPC_ABLATE
            if (DPC_ABL(6)) { o[e] = v[j + e + RB]; continue; }
#endif
#pragma unroll
            for (int tp = 0; tp < 2 * RB + 1; ++tp) h = fmaf(taps.w[tp], v[j + e + tp], h);
            if constexpr (kSym) o[e] = wpass_lanes_sym<RB>(h, taps);
            else o[e] = wpass_lanes<RB>(h, taps);
          }
          // one grid row per store instruction (lane = x: 256 contiguous bytes), written through (dpc_kernels.h).
          // (Round 3's other shape -- a 4 x 4 quad transpose over DPP quad permutes and ONE 16-byte buffer_store ... sc1 per lane
          // -- gave wrong values in lanes 13 + 16 k at tap radius 1.  Round 4 isolated that shape
          // (tools/microbench/sc1_b128_store_probe.hip): value-identical to the ordinary store on the hardware at radius 1 and 3,
          // and the old kernel rebuilt with it keeps every gfx9 hazard distance (VALU -> DPP two wait states, > 64-bit store
          // data -> VALU write one).  Not a property of the instruction sequence; the failing build is not in the history.  The
          // row stores are as fast and need no transpose -- profiles/LAB_NOTES.md A.6.)
#ifdef DPC_ABLATE
          if (!DPC_ABL(5) || o[0] == 123.456f)
#endif
          {
#pragma unroll
            for (int e = 0; e < 4; ++e) store_through(Tout + (size_t)(j + e) * kXG, o[e]);
          }
        }
      };
      if (symmetric) rows(std::true_type{});
      else rows(std::false_type{});
    }
    if (!more) break;
    request_records(rr_next);
    zero_rest();
  }
  DPC_STAMP(5);
}

template <int ZS, int RB>
int launch_splat_xl_zr(const DpcParams* p, Cells cells, const float* kxy, const TapPlan& pxy, float* Tbuf, uint64_t* mask,
                       float* sse, float* loss_zero, int* winner_zero, unsigned long long* ticket_zero, hipStream_t st) {
  constexpr size_t lds = (size_t)ZS * (kXG + 2 * RB) * kXG * sizeof(unsigned long long) + kTabInts * sizeof(int);
  static_assert(lds <= kLdsLimit, "forward slab does not fit LDS");
  auto kern = k_splat_xl<ZS, RB>;
  static LdsLimit limit;
  int rc = set_lds(kern, lds, limit);
  if (rc != DPC_OK) return rc;
  // slabs per workgroup (see the kernel): as many as still leave one workgroup per CU
  const int nslab = (p->D + ZS - 1) / ZS;
  int nround = 1;
#ifndef DPC_XL_ONE_SLAB
#ifndef DPC_XL_WGS_PER_CU
#define DPC_XL_WGS_PER_CU 1
#endif
  for (int c = 2; c <= 16; c *= 2)
    if (nslab % c == 0 && (size_t)(nslab / c) * p->B >= (size_t)kNumCUs * DPC_XL_WGS_PER_CU) nround = c;
#endif
  // the Horner W pass wants w[RB - k] == w[RB + k] bit for bit (every Gaussian: exp(-x^2 / 2 sigma^2) of +-x) and the values
  // in the launch itself (tap values read from device memory under a DeviceSchedule are not inspected here)
  const TapsT<RB> tw = make_taps<RB>(kxy, pxy, false);
  int symmetric = p->dev_taps_xy == nullptr;
  for (int k = 1; k <= RB; ++k) symmetric = symmetric && memcmp(&tw.w[RB - k], &tw.w[RB + k], sizeof(float)) == 0;
  DPC_LAUNCH("k_splat_xl", kern, dim3((nslab / nround) * p->B), dim3(ZS * kXWavesPerPlane * 64), lds, st, *p, cells,
             tw, nround, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, symmetric);
  return launch_ok();
}

}  // namespace

// 64 x 64 planes, a Gaussian of effective radius 1..6 (sigma_rel up to ~1.4 at 21 taps): the x-in-lanes kernels
bool xl_applies(const DpcParams* p, int bucket) {
  return p->H == kXG && p->W == kXG && (bucket == 1 || bucket == 2 || bucket == 3 || bucket == 4 || bucket == 6);
}

#ifndef DPC_XL_FWD_ZS
#define DPC_XL_FWD_ZS 4   // planes per forward slab: 4 -> one 1024-thread workgroup per CU, 2 -> two 512-thread workgroups per CU
#endif
int launch_splat_xl(int bucket, const DpcParams* p, Cells cells, const float* kxy, const TapPlan& pxy, float* Tbuf, uint64_t* mask,
                    float* sse, float* loss_zero, int* winner_zero, unsigned long long* ticket_zero, hipStream_t st) {
  constexpr int ZS = DPC_XL_FWD_ZS;
  switch (bucket) {
    case 1: return launch_splat_xl_zr<ZS, 1>(p, cells, kxy, pxy, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, st);
    case 2: return launch_splat_xl_zr<ZS, 2>(p, cells, kxy, pxy, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, st);
    case 3: return launch_splat_xl_zr<ZS, 3>(p, cells, kxy, pxy, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, st);
    case 4: return launch_splat_xl_zr<ZS, 4>(p, cells, kxy, pxy, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, st);
    case 6: return launch_splat_xl_zr<ZS, 6>(p, cells, kxy, pxy, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, st);
  }
  return DPC_ERR_TAPS;
}

}  // namespace dpck

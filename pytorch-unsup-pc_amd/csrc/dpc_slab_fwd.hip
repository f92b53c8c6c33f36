// Forward slab kernels: per-point transform + cell location + z-sort (k_locate), splat + clamp mask + W/H Gaussian
// passes in LDS (k_splat_hw).  Design notes: DESIGN.md section 4.
#include "dpc_kernels.h"

DPC_DEBUG_SETTERS(fwd)

namespace dpck {
namespace {

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
  // Counting sort by bin that is STABLE without any ordered atomic (ranks handed out by an atomic counter are arrival order,
  // which changes from run to run -- and the order of the records inside a bin decides which thread of the backward adds which
  // point into its partial sums).  [r4] By ballots instead of 256-bit membership masks in LDS: a wave finds, for every lane,
  // the lanes that share its bin (one ballot per bit of the bin number), the rank inside the wave is the popcount below the
  // lane (v_mbcnt), the first lane of every group leaves the group's size in cnt[wave][bin], and the first wave turns the
  // 4 x (D + 1) counts into start positions -- no LDS atomics, a quarter of the LDS, the same order as before bit for bit.
  extern __shared__ __attribute__((aligned(16))) int cnt[];   // [waves][D + 1] group sizes, then start positions; sized by the launch
  __shared__ int total_pts;
  const Blk bk = block_coords(P.B);
  const int b = bk.y, blk = bk.x, tid = threadIdx.x;
  const int D = P.D, nbins = D + 1;
  const int i = blk * kLocThreads + tid;
  // N is the capacity of the call; a device-side count of live points (a scheduled keep-count under graph replay) may cut
  // it short: the points beyond it become out-of-bounds records (bin D) that no later kernel looks at
  const int n_live = P.n_live != nullptr ? min(P.N, *P.n_live) : P.N;
  const bool present = i < P.N, live = i < n_live;
  constexpr int NW = kLocThreads / DPC_WAVE;
  for (int k = tid; k < nbins * NW; k += kLocThreads) cnt[k] = 0;
  PointRec rec;
  rec.code = -1; rec.tz = rec.ty = rec.tx = 0.f;
  float src_pt[3] = {0.f, 0.f, 0.f};  // the untransformed point (SRC 0), carried next to its record for the backward
  int src_i = i;                      // ... and its index inside the stored point set (where its gradient goes)
  const size_t idx = (size_t)b * P.N + (live ? i : 0);
  bool in_set = true;
  if (SRC == 0 && live) {
    // the point is requested FIRST: its load (behind the index load, when there is one) is the longest latency of the
    // kernel, and the camera below -- scalar loads of q, t, f, an fp32 normalisation -- fits under it
    const int reps = P.point_replicas > 1 ? P.point_replicas : 1;  // replicas of one point set read the same rows
    if (P.point_index != nullptr) {   // this cloud's own subset of the stored set
      src_i = P.point_index[idx];
      // an index outside the stored set never becomes an address (the reference's fancy indexing raises IndexError,
      // point_cloud_to.py:266-295): the point is dropped -- an out-of-bounds record at source index 0, which neither the
      // forward nor the backward touches -- and the caller's status word says so
      in_set = (unsigned)src_i < (unsigned)P.N_src;
      if (!in_set) {
        src_i = 0;
        if (P.status != nullptr) atomicOr(P.status, (int)DPC_STATUS_BAD_INDEX);
      }
    }
    const float* p = static_cast<const float*>(pts) + ((size_t)(b / reps) * points_per_set(P) + src_i) * 3;
    src_pt[0] = p[0]; src_pt[1] = p[1]; src_pt[2] = p[2];
  }
  // every thread normalises the quaternion itself (a dozen fp32 ops): cheaper than one thread doing it while 255 wait
  CameraRef cam_s;
  if (SRC == 0) cam_s = load_camera_ref(P, q, t, f, b);

  if (live) {
    double Z, Y, X;
    if (SRC == 0) {
      project_point_ref(cam_s, src_pt[0], src_pt[1], src_pt[2], Z, Y, X);
      if (!in_set) Z = Y = X = 2.0;   // outside [-1/2, 1/2]^3: make_record marks it out of bounds
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
  if (present && !live && P.point_index != nullptr) src_i = 0;   // a skipped point: an out-of-bounds record at a valid source index
  const int bin = rec.code < 0 ? D : (rec.code >> 20);
  // the lanes of this wave that hold a point of the same bin: one ballot per bit of the bin number
  const int wave = tid / DPC_WAVE;
  unsigned long long same = __ballot(present);
  for (int bit = 0; (nbins - 1) >> bit; ++bit) {   // block-uniform trip count: the bits of the largest bin number
    const bool mine = (bin >> bit) & 1;
    const unsigned long long has = __ballot(mine);
    same &= mine ? has : ~has;
  }
  const int below = __builtin_amdgcn_mbcnt_hi((unsigned int)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)same, 0u));
  __syncthreads();  // the counts are zeroed (the transform above ran under that latency)
  if (present && below == 0) cnt[wave * nbins + bin] = __popcll(same);   // the group's first lane
  __syncthreads();

  // start positions by the first wave: lane l owns bins [l*C, (l+1)*C); cnt[w][k] becomes the first sorted position of the
  // points of bin k that sit in wave w (waves in order inside a bin: stable)
  if (tid < DPC_WAVE) {
    const int C = (nbins + DPC_WAVE - 1) / DPC_WAVE;
    auto population = [&](int k) {
      int n = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) n += cnt[w * nbins + k];
      return n;
    };
    int sum = 0;
    for (int k = tid * C; k < min((tid + 1) * C, nbins); ++k) sum += population(k);
    const int incl = wave_inclusive_scan(sum);
    int run = incl - sum;
    for (int k = tid * C; k < min((tid + 1) * C, nbins); ++k) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const int c = cnt[w * nbins + k];
        cnt[w * nbins + k] = run;
        run += c;
      }
    }
    if (tid == DPC_WAVE - 1) total_pts = incl;
  }
  __syncthreads();

  // sorted chunk staged in LDS, then copied out with one coalesced 16-byte store per lane and array
  __shared__ int4 stage[2 * kLocThreads];
  if (present) {
    const int pos = cnt[wave * nbins + bin] + below;   // bin start of this wave's group + rank inside the group
    int4 v;
    v.x = rec.code; v.y = __float_as_int(rec.tz); v.z = __float_as_int(rec.ty); v.w = __float_as_int(rec.tx);
    stage[pos] = v;
    int4 a;
    a.x = __float_as_int(src_pt[0]); a.y = __float_as_int(src_pt[1]); a.z = __float_as_int(src_pt[2]); a.w = src_i;
    stage[kLocThreads + pos] = a;
  }
  __syncthreads();
  const size_t chunk = chunk_bytes(D);
  uint8_t* out = cells_out + ((size_t)b * bk.nx + blk) * chunk;
  const int npts = min(kLocThreads, P.N - blk * kLocThreads);
  if (tid < npts) {
    // (ordinary stores: written through, these 32-byte-per-thread record stores cost k_locate 0.4 us at c2 and 4.7 us at c5;
    // with the non-temporal bit nothing changes)
    reinterpret_cast<int4*>(out)[tid] = stage[tid];
    reinterpret_cast<int4*>(out + (size_t)kLocThreads * sizeof(PointRec))[tid] = stage[kLocThreads + tid];
  }
  uint16_t* offs = reinterpret_cast<uint16_t*>(out + (size_t)kLocThreads * 2 * sizeof(PointRec));
  for (int k = tid; k < nbins + 1; k += kLocThreads) offs[k] = (uint16_t)(k < nbins ? cnt[k] : total_pts);   // cnt[0][k]: the bin's start
}

// ------------------------------------------------------------------------------------------------------
// Forward 1: splat + mask + clamp + W/H Gaussian passes.                          grid (nslab, B)
//   GS = 0: generic (runtime dims, Zs = zs_rt);  GS > 0: specialised, ZS planes per slab.
//   Tbuf == nullptr: stage-level pointcloud2voxels3d_fast, only `raw` is written.
// ------------------------------------------------------------------------------------------------------
template <int GS, int ZS, int RB>
__global__ __launch_bounds__(kSlabThreads) void k_splat_hw(DpcParams P, Cells cells, TapsT<RB> taps_arg, int zs_rt,
                                                           float* __restrict__ raw, float* __restrict__ Tbuf,
                                                           uint64_t* __restrict__ mask, float* __restrict__ sse,
                                                           float* __restrict__ loss_zero, int* __restrict__ winner_zero,
                                                           unsigned long long* __restrict__ ticket_zero) {
  extern __shared__ __attribute__((aligned(16))) float slab[];
  const TapsT<RB> taps = resolve_taps<RB>(taps_arg, P.dev_taps_xy, P.taps_xy, false);
  const Blk bk = block_coords(P.B);
  if (sse != nullptr && bk.x == 0 && threadIdx.x == 0) {  // k_zcol_fwd accumulates into these
    sse[bk.y] = 0.f;
    if (winner_zero != nullptr) winner_zero[bk.y] = 0;   // K == 1: sample == cloud, candidate 0 wins
    if (ticket_zero != nullptr) ticket_zero[bk.y] = 0ull;  // k_zcol_fwdbwd's per-cloud sum-and-count word
    if (ticket_zero != nullptr && bk.y == 0) ticket_zero[bk.ny] = 0ull;  // ... and the batch's, behind them
    if (loss_zero != nullptr && bk.y == 0) *loss_zero = 0.f;
  }
  const int D = P.D, H = P.H, W = P.W;
  // generic kernel: |zs_rt| planes per slab; zs_rt < 0 = float accumulators (planes so wide that only ONE 4-byte plane fits)
  const int Zs = GS ? ZS : (zs_rt < 0 ? -zs_rt : zs_rt);
  const int b = bk.y, z0 = bk.x * Zs;
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
    // The workgroup stays for `nround` slabs of its cloud (bk.x, bk.x + nx, ...; the launcher keeps one workgroup per CU):
    // a workgroup that ends is only replaced once its stores have drained, one that goes on prepares the next slab under
    // them (k_splat_xl has the measurement).  The next slab's record offsets are requested BEFORE this slab's stores: a
    // wave's vector memory operations complete in order.
    const int nround = zs_rt, nslab = (D + ZS - 1) / ZS, slab0 = bk.x, slab_step = bk.nx;
    const TapsT<RB>& taps_outer = taps;
    RecordRange rr{0, 0};
    if (flat) rr = load_record_range(cells, b, max(slab0 * ZS - 1, 0), slab0 * ZS + min(ZS, D - slab0 * ZS));
   for (int round = 0; round < nround; ++round) {
    const int sl = slab0 + round * slab_step;
    if (sl >= nslab) break;                // block-uniform
    const int z0 = sl * ZS, nz = min(ZS, D - z0);
    // the thread id is made opaque every round: otherwise everything a round derives from it (rows, segments, addresses of
    // five phases) is hoisted in front of the loop and lives through all of it (85 -> 123 registers, spills at radius 10)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    int b = bk.y;                      // likewise the cloud: every global address of a round hangs on it
    asm volatile("" : "+s"(b));
    TapsT<RB> taps = taps_outer;       // ... and the taps: their {w, w} pairs for the packed FMAs were built once in front
#pragma unroll                         // of the loop and parked in VGPR lanes, two v_readlane per use
    for (int i = 0; i < 2 * RB + 1; ++i) asm volatile("" : "+s"(taps.w[i]));
    const int sl_next = sl + slab_step;
    const bool more = round + 1 < nround && sl_next < nslab;
    if (round > 0) __syncthreads();        // the last slab's fp32 rows have been read
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
            atomicAdd(&acc[(zz * GS + c.iy + j) * WPA + Geo::PAD + c.ix + e], to_fixed(w));  // ds_add_u64
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
    auto to_float = [](unsigned long long a) { return from_fixed(a); };
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
        if (Tbuf != nullptr && present) store_through(Tbuf + ((size_t)b * D + z0) * HW + i, w2 * fminf(v, 1.0f));
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
      if (more && flat) {
        const int zn = sl_next * ZS;
        rr = load_record_range(cells, b, max(zn - 1, 0), zn + min(ZS, D - zn));
      }
      float* Tout = Tbuf + ((size_t)b * D + z0) * HW;
      hpass_fast<Geo, GS, RB, ZS, false>(slab, taps, [&](int z, int y, int x, f32x2 v2) {
        if (z < nz) store_through(reinterpret_cast<f32x2*>(Tout + ((size_t)z * GS + y) * GS + x), v2);   // dpc_kernels.h
      }, tid);
      DPC_STAMP(5);
    }
    if (!more) break;
   }
  } else {
    const int WP = odd_stride(W);
    const int nvox = nz * H * WP;
    // Accumulators: 64-bit fixed point like the specialised kernels (exact, so the sums -- and everything downstream -- are the
    // same bits whatever order the points arrive in), 8 bytes per voxel, turned into the fp32 slab in place afterwards.  Only
    // planes so wide that a single 8-byte plane does not fit (zs_rt < 0: beyond ~141 x 141, forward-only territory) fall back
    // to float LDS atomics, whose arrival order shows in the last bits.
    const bool acc64 = zs_rt > 0;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(slab);
    if (acc64) for (int i = tid; i < nvox; i += nthr) acc[i] = 0ull;
    else for (int i = tid; i < nvox; i += nthr) slab[i] = 0.f;
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
            const float wt = c.wz[k] * c.wy[j] * c.wx[e];
            if (acc64) atomicAdd(&acc[(zz * H + yy) * WP + xx], to_fixed(wt));   // ds_add_u64
            else atomicAdd(&slab[(zz * H + yy) * WP + xx], wt);                 // ds_add_f32
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
        float vraw = 2.f;
        bool pass = false;
        if (in) {
          if (acc64) {
            const unsigned long long a = acc[(z * H + y) * WP + x];
            vraw = from_fixed(a);
            pass = a <= kFixOne;           // decided on the exact integer, like the specialised kernels
          } else {
            vraw = slab[(z * H + y) * WP + x];
            pass = vraw <= 1.0f;
          }
        }
        const unsigned long long bits = __ballot(pass);
        if (mask != nullptr && lane == 0) mask[((size_t)b * D + z0 + z) * wpp + c] = bits;
        if (raw != nullptr && in) raw[((size_t)b * D + z0 + z) * iHW + idx] = vraw;
      }
    }
    if (acc64 && Tbuf != nullptr) {
      // accumulators -> fp32 slab, in place: float i lands on the bytes of accumulator i / 2, so the slab is converted in
      // chunks -- a chunk's accumulators are all read before any of its floats is written, and the floats of chunk c only
      // reach into accumulators of chunks <= c
      constexpr int CH = 8;
      __syncthreads();   // the mask pass is done with the accumulators
      for (int base = 0; base < nvox; base += CH * nthr) {
        unsigned long long r[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const int i = base + k * nthr + tid;
          r[k] = i < nvox ? acc[i] : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const int i = base + k * nthr + tid;
          if (i < nvox) slab[i] = from_fixed(r[k]);
        }
      }
      __syncthreads();
    }
    if (Tbuf == nullptr) return;
    float* Tout = Tbuf + ((size_t)b * D + z0) * HW;
    if (RB == 0) {
      const float w2 = taps.w[0] * taps.w[0];
      for (int i = tid; i < nz * H * W; i += nthr) {
        const int x = i % W, zy = i / W;
        store_through(Tout + i, w2 * fminf(slab[zy * WP + x], 1.0f));
      }
      return;
    }
    wpass_inplace<RB, true>(slab, nz, H, W, WP, taps, [](int, int, float val) { return val; });
    hpass<RB>(slab, nz, H, W, WP, taps, [&](int z, int y, int x, float val) { store_through(Tout + (z * H + y) * W + x, val); });
  }
}

template <int GS, int ZS, int RB>
int launch_splat_fast(const DpcParams* p, Cells cells, const float* kxy, const TapPlan& pxy, float* raw, float* Tbuf,
                      uint64_t* mask, float* sse, float* loss_zero, int* winner_zero, unsigned long long* ticket_zero, hipStream_t st) {
  using Geo = FwdGeo<GS, ZS, RB>;
  constexpr size_t lds = ((size_t)ZS * GS * Geo::WPA + Geo::PAD) * sizeof(unsigned long long) + kTabInts * sizeof(int);
  static_assert(lds >= Geo::slab_floats(ZS) * sizeof(float), "the fp32 slab reuses the accumulator memory");
  static_assert(lds <= kLdsLimit, "forward slab does not fit LDS");
  auto kern = k_splat_hw<GS, ZS, RB>;
  static LdsLimit limit;
  int rc = set_lds(kern, lds, limit);
  if (rc != DPC_OK) return rc;
  // slabs per workgroup (see the kernel): as many as still leave one workgroup per CU; the branches without passes end the
  // kernel after their one slab
  const int nslab = (p->D + ZS - 1) / ZS;
  int nround = 1;
#ifndef DPC_HW_ONE_SLAB
  if (Tbuf != nullptr && RB > 0)
    for (int c = 2; c <= 16; c *= 2)
      if (nslab % c == 0 && (size_t)(nslab / c) * p->B >= (size_t)kNumCUs) nround = c;
#endif
  DPC_LAUNCH("k_splat_hw", kern, dim3((nslab / nround) * p->B), dim3(Geo::NT), lds, st, *p, cells,
             make_taps<RB>(kxy, pxy, false), nround, raw, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero);
  return launch_ok();
}

template <int RB>
int launch_splat_rb(const DpcParams* p, Cells cells, const float* kxy, const TapPlan& pxy, float* raw, float* Tbuf,
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
  // 64-bit fixed-point accumulators (8 bytes per voxel) wherever a plane of them fits; the kernel says which by the sign of
  // its slab-thickness argument
  const bool acc64 = fit >= 2;
  const int Zs = std::min(acc64 ? fit / 2 : fit, std::max(1, (p->D + 7) / 8));
  const size_t lds = (size_t)Zs * p->H * (p->W | 1) * (acc64 ? sizeof(unsigned long long) : sizeof(float));
  auto kern = k_splat_hw<0, 0, RB>;
  static LdsLimit limit;
  int rc = set_lds(kern, lds, limit);
  if (rc != DPC_OK) return rc;
  DPC_LAUNCH("k_splat_hw", kern, dim3(((p->D + Zs - 1) / Zs) * p->B), dim3(slab_threads(p)), lds, st, *p, cells,
             make_taps<RB>(kxy, pxy, false), acc64 ? Zs : -Zs, raw, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero);
  return launch_ok();
}

}  // namespace

int launch_splat(int bucket, const DpcParams* p, Cells cells, const float* kxy, const TapPlan& pxy, float* raw, float* Tbuf,
                 uint64_t* mask, float* sse, float* loss_zero, int* winner_zero, unsigned long long* ticket_zero, hipStream_t st) {
#ifndef DPC_NO_XL
  if (xl_applies(p, bucket) && raw == nullptr && Tbuf != nullptr)
    return launch_splat_xl(bucket, p, cells, kxy, pxy, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, st);
#endif
  int rc = DPC_OK;
#define DPC_SPLAT(RB) rc = launch_splat_rb<RB>(p, cells, kxy, pxy, raw, Tbuf, mask, sse, loss_zero, winner_zero, ticket_zero, st)
  DPC_FOR_BUCKET(bucket, DPC_SPLAT)
#undef DPC_SPLAT
  return rc;
}

int launch_locate(const DpcParams* p, int src, const void* pts, const float* q, const float* t, const float* f,
                  float* tr_pc, void* cells, hipStream_t st) {
  if (p->N == 0 || p->B == 0) return DPC_OK;
  dim3 g(num_chunks(p->N) * p->B), blk(kLocThreads);
  uint8_t* out = static_cast<uint8_t*>(cells);
  const size_t lds = (size_t)(p->D + 1) * (kLocThreads / DPC_WAVE) * sizeof(int);  // group sizes / start positions per wave and bin
  if (src == 0) DPC_LAUNCH("k_locate", k_locate<0>, g, blk, lds, st, *p, pts, q, t, f, tr_pc, out);
  else if (src == 1) DPC_LAUNCH("k_locate", k_locate<1>, g, blk, lds, st, *p, pts, q, t, f, tr_pc, out);
  else DPC_LAUNCH("k_locate", k_locate<2>, g, blk, lds, st, *p, pts, q, t, f, tr_pc, out);
  return launch_ok();
}

}  // namespace dpck

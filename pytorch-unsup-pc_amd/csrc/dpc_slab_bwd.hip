// Backward slab kernel: adjoint H pass from global dT, adjoint W pass at the gathered corners, 8-corner gather,
// transform backward and the per-cloud reductions (k_gather_hw).  Design notes: DESIGN.md section 4.
#include "dpc_kernels.h"

DPC_DEBUG_SETTERS(bwd)

namespace dpck {
namespace {

// ------------------------------------------------------------------------------------------------------
// Backward 2: adjoint H/W passes + clamp mask + trilinear gather + transform backward.   grid (nslab, B)
//   The slab holds cell layers [z0, z0+Zs) plus one halo plane so each point's 8 corners are local.
// ------------------------------------------------------------------------------------------------------
template <int GS, int ZS, int RB>
__global__ __launch_bounds__(kSlabThreads) void k_gather_hw(DpcParams P, Cells cells, const float* __restrict__ pc,
                                                            const float* __restrict__ q, const float* __restrict__ t,
                                                            const float* __restrict__ f, TapsT<RB> taps_arg, int zs_rt,
                                                            const float* __restrict__ dT,
                                                            const uint64_t* __restrict__ mask,
                                                            const float* __restrict__ ds_part, int n_ds_part,
                                                            float* __restrict__ dpc, float* __restrict__ dsmall,
                                                            double* __restrict__ cg_part,
                                                            unsigned int* __restrict__ cg_count, LossArgs la,
                                                            unsigned long long* __restrict__ dpc_fixed) {
  extern __shared__ __attribute__((aligned(16))) float slab[];
  const TapsT<RB> taps_adj = resolve_taps<RB>(taps_arg, P.dev_taps_xy, P.taps_xy, true);
  const int D = P.D, H = P.H, W = P.W, HW = H * W;
  const int Zs = GS ? ZS : zs_rt;
  // One-layer slabs (planes too big for more: 128^2) ROLL: the workgroup walks `roll` consecutive layers, keeps the plane
  // two layers share in LDS (ping-pong of the two plane buffers) and pays start-up, reduction and atomics once.
  constexpr bool kRolls = GS > 0 && ZS == 1 && RB > 0;
  const int roll = kRolls ? zs_rt : 1;
  const int reps = P.point_replicas > 1 ? P.point_replicas : 1;
  // replicas of a point set, or clouds that picked their points out of a stored set: dpc is [B/reps,Nset,3], zeroed by the
  // caller, and every cloud ADDS its gradients into it
  const bool shared_points = reps > 1 || P.point_index != nullptr;
  // K candidates per sample and nothing to write for the losers (shared gradient buffer, small gradients zeroed by
  // k_zcol_bwd): the grid runs over SAMPLES and the workgroup takes the winning cloud (see winners_only())
  const bool wo = winners_only(la) && shared_points;
  // ... and when a point set belongs to the K candidates of ONE sample, its single winner is the only cloud that ever
  // writes into the set's gradient: plain stores then (float atomics to 24 000 scattered addresses tripled the gather
  // phase of the c5 winners: 11.4 -> 3.9 us)
  // (not with a point_index: its rows may repeat an index, and two contributions to one point must be summed)
  const bool single_writer = wo && reps == la.K && P.point_index == nullptr;
  const Blk bk = block_coords(wo ? P.B / la.K : P.B, la.winner != nullptr ? la.K : 1);
  const int b = wo ? bk.y * la.K + la.winner[bk.y] : bk.y, z0 = bk.x * Zs * roll;
  const int Nset = points_per_set(P);
  if (cloud_loses(la, b)) {  // a losing pose candidate: zero gradient, no work (block-uniform)
    if (shared_points) {
      if (bk.x == 0 && threadIdx.x == 0) dsmall[(size_t)DPC_COL_DS * P.B + b] = 0.f;
      return;
    }
    float* dz = dpc + (size_t)b * Nset * 3;
    auto zero3 = [&](const PointRec&, const int4* aux) {
      const int i = aux->w;
      dz[3 * i + 0] = 0.f; dz[3 * i + 1] = 0.f; dz[3 * i + 2] = 0.f;
    };
    for_each_record(cells, b, z0, min(z0 + Zs * roll, D), zero3);
    if (bk.x == 0) {
      for_each_record(cells, b, D, D + 1, zero3);
      if (threadIdx.x == 0) dsmall[(size_t)DPC_COL_DS * P.B + b] = 0.f;
    }
    return;
  }
  // Rolling workgroups walk their layers in ALTERNATING directions: slab x walks z0 -> z0+roll-1, its neighbour x+1 walks
  // z0'+roll-1 -> z0'.  The plane two neighbours share (z0' = z0+roll: the last halo of one, the first plane of the other) is
  // then read by both at the same moment -- the end of their walks, or the start -- so one of them finds it in the XCD's L2
  // (all slabs of a cloud run on one XCD) instead of each fetching it from memory 30 us apart: the rolling backward read
  // 1.42 x its algorithmic bytes from HBM in round 3 (profiles/r03_rocprof_summary_c4.json).
  const bool down = kRolls && (bk.x & 1);
  const int zfirst = down ? min(z0 + roll, D) - 1 : z0;   // the first layer this workgroup gathers
  const int nzp = min(Zs + 1, D - zfirst);  // planes present (cell layers + halo)
  const int tid = threadIdx.x, nthr = blockDim.x;
  // camera inputs and the upstream scalar: requested now, first used after the slab is in LDS
  const CameraRaw cam_raw = load_camera_raw(P, q, t, f, b);
  const float upstream = (la.scale_in_gather && la.dloss != nullptr) ? *la.dloss : 1.0f;
  const int wpp = (HW + 63) / 64;
  const float* src = dT + ((size_t)b * D + zfirst) * HW;
  const uint64_t* mrow = mask + ((size_t)b * D + zfirst) * wpp;
  float* red;

  if constexpr (GS > 0) {
    constexpr int NPL = ZS + 1;
    using Geo = BwdGeo<GS, RB, NPL>;
    red = slab + ((Geo::slab_floats(NPL) + 3) / 4) * 4;
    RecordRange rr{0, 0};
    if (cells.nblk <= DPC_WAVE) rr = load_record_range(cells, b, zfirst, min(zfirst + Zs, D));  // in flight under the H-pass
    const uint32_t* mask32 = reinterpret_cast<const uint32_t*>(mrow);
    DPC_STAMP(8);
    for (int i = tid; i < (NPL * GS + 1) * (Geo::PAD / 4); i += Geo::NT) {  // zero the row pads (W-pass halo)
      const int p4 = i % (Geo::PAD / 4), row = i / (Geo::PAD / 4);
      *reinterpret_cast<f32x4*>(slab + Geo::row_start(row) + 4 * p4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (Geo::NARROW)   // ... and the floats a window can reach in front of the first row and behind the last pad
      if (tid < 2 * Geo::LEAD) slab[tid < Geo::LEAD ? tid : Geo::row_start(NPL * GS) + Geo::PAD + tid - Geo::LEAD] = 0.f;
    if constexpr (RB == 0) {
      // planes -> LDS (16-byte global loads, 16-byte LDS stores); absent planes are zeroed
      for (int i = tid; i < NPL * GS * (GS / 4); i += Geo::NT) {
        const int x4 = i % (GS / 4), zy = i / (GS / 4);
        f32x4 val = f32x4{0.f, 0.f, 0.f, 0.f};
        if (zy < nzp * GS) val = *reinterpret_cast<const f32x4*>(src + (size_t)zy * GS + 4 * x4);
        *reinterpret_cast<f32x4*>(slab + Geo::row_start(zy) + Geo::PAD + 4 * x4) = val;
      }
      if (cells.nblk <= DPC_WAVE) finish_record_table(rr, reinterpret_cast<int*>(red + kRedTab));
      __syncthreads();
      const float w2 = taps_adj.w[0] * taps_adj.w[0];
      for (int i = tid; i < NPL * GS * GS; i += Geo::NT) {
        const int x = i % GS, zy = i / GS;
        const bool pass = zy < nzp * GS && ((mask32[i >> 5] >> (i & 31)) & 1u);
        float* cell = slab + Geo::row_start(zy) + Geo::PAD + x;
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
  float* dcloud = dpc + (size_t)(b / reps) * Nset * 3;
  auto corner = [&](int zz, int yy, int xx) -> float {
    if constexpr (GS > 0) return slab[BwdGeo<GS, RB, ZS + 1>::at(zz, yy, xx)];
    else return slab[(zz * H + yy) * odd_stride(W) + xx];
  };
  // (Requesting a thread's NEXT record and point before it works on the current one -- so that the second pass over a slab
  // with more records than threads starts with its loads answered -- measured 0.3-0.5 us slower: the pass is bound by its
  // arithmetic, 2.1-2.7 us per pass of four waves per SIMD, not by the two dependent loads in front of it.)
  // One (z, y) row of a point's cell: the adjoint W pass at its two x corners, masked (GS > 0, RB > 0 only).
  // Narrow layout (pads shorter than the radius): the taps of a point's window that fall outside its row AND beyond the row's
  // zero pad -- there lies the neighbouring row, not zeros -- get weight zero, once per point for its four rows.  Only window
  // positions within RB - PAD of the window's ends can do that (`reaches_over`), and only for points near the row's ends; the
  // other taps keep their scalar weights.  wm[e][tp] is tap tp as seen from the point's x corner e (window position e + tp).
  // (0 x finite = 0: exact.  Where d T is not finite the step has diverged anyway.)
  auto reaches_over = [](int pos) constexpr {
    constexpr int P4 = BwdGeo<(GS > 0 ? GS : 64), RB, ZS + 1>::PAD;
    return pos < RB - P4 || pos >= RB + P4 + 1;
  };
  auto masked_taps = [&](const Cell& c, float (&wm)[2][2 * RB + 1]) {
    constexpr int P4 = BwdGeo<(GS > 0 ? GS : 64), RB, ZS + 1>::PAD;
#pragma unroll
    for (int pos = 0; pos < 2 * RB + 2; ++pos) {
      if (!reaches_over(pos)) continue;
      // x = ix - RB + pos: in front of the row for the low positions (x >= 0 wanted), behind it for the high ones (x < GS)
      const bool inside = pos < RB - P4 ? c.ix >= RB - pos : c.ix <= GS - 1 + RB - pos;
      if (pos <= 2 * RB) wm[0][pos] = inside ? taps_adj.w[pos] : 0.f;
      if (pos >= 1) wm[1][pos - 1] = inside ? taps_adj.w[pos - 1] : 0.f;
    }
  };
  auto corner_row = [&](const Cell& c, int k, int j, float& o0, float& o1, const float (&wm)[2][2 * RB + 1]) {
    if constexpr (GS > 0 && RB > 0) {
      using Geo = BwdGeo<GS, RB, ZS + 1>;
      const uint32_t* mlds = reinterpret_cast<const uint32_t*>(red + kRedMask);
      o0 = o1 = 0.f;
      if ((c.iz + k < D) && (c.iy + j < GS)) {
        const int plane = kRolls ? ((c.iz - zfirst + k) & 1) : (c.iz - z0 + k);  // rolling: plane z lives in buffer (z - zfirst) & 1
        const int row = plane * GS + c.iy + j;
        const float* rp = slab + Geo::row_start(row) + Geo::PAD + c.ix - RB;  // x = ix-RB .. ix+1+RB; pads are zero
        float v[2 * RB + 2];
#pragma unroll
        for (int i = 0; i < 2 * RB + 2; ++i) v[i] = rp[i];
        float o[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float acc = 0.f;
#pragma unroll
          for (int i = 0; i < 2 * RB + 1; ++i) {
            const int tp = tap_edge_first<RB>(i);   // edges first, centre last (dpc_common.h)
            if (Geo::NARROW && reaches_over(e + tp)) acc = fmaf(wm[e][tp], v[e + tp], acc);
            else acc = fmaf(taps_adj.w[tp], v[e + tp], acc);
          }
          const int x = c.ix + e;
          const bool pass = x < GS && ((mlds[row * (GS / 32) + (x >> 5)] >> (x & 31)) & 1u);
          o[e] = pass ? acc : 0.f;
        }
        o0 = o[0]; o1 = o[1];
      }
    }
  };
  // The rest of a point's backward, from its grid-coordinate gradient on: transform backward, camera sums, d(point).
  auto finish_point = [&](const int4 pt, float dgz, float dgy, float dgx) {
    const int i = pt.w;
    dgz *= upstream; dgy *= upstream; dgx *= upstream;  // 1 unless dT was produced by the forward for dloss = 1
    const float px = __int_as_float(pt.x), py = __int_as_float(pt.y), pz = __int_as_float(pt.z);
    const Projected o = project_point(cam, px, py, pz);
    float dpx, dpy, dpz;
    project_point_bwd(cam, o, px, py, pz, dgz * (float)(D - 1), dgy * (float)(H - 1), dgx * (float)(W - 1), dpx, dpy, dpz, g);
    if (shared_points && !single_writer) {
      // several clouds add into this point set's gradient: 64-bit fixed-point adds (exact, so the sum does not depend on
      // who arrives first); k_fixed_to_dpc turns the sums into floats behind this launch
      if (grad_fits_fixed(dpx, dpy, dpz)) {
        unsigned long long* acc = dpc_fixed + ((size_t)(b / reps) * Nset + i) * 3;
        atomicAdd(acc + 0, grad_to_fixed(dpx)); atomicAdd(acc + 1, grad_to_fixed(dpy)); atomicAdd(acc + 2, grad_to_fixed(dpz));
      } else {   // NaN / Inf / out of range: poison the set (dpc_kernels.h), k_fixed_to_dpc writes NaN
        atomicOr(reinterpret_cast<unsigned int*>(dpc_fixed + (size_t)(P.B / reps) * Nset * 3) + b / reps, 1u);
      }
    } else {
      // (ordinary stores: these 12-byte scattered writes cost the kernel 4 us when written through)
      dcloud[3 * i + 0] = dpx; dcloud[3 * i + 1] = dpy; dcloud[3 * i + 2] = dpz;
    }
  };
  auto gather = [&](const PointRec& rec, const int4* aux) {
    const int4 pt = *aux;  // {px, py, pz, original index}: one 16-byte load, issued next to the record's
    const Cell c = cell_from_record(rec);
    float cv[2][2][2];
    if constexpr (GS > 0 && RB > 0) {
      // adjoint W-pass evaluated right here, at the two x corners of each of the four (z,y) rows, then masked
      float wm[2][2 * RB + 1];
      if constexpr (BwdGeo<GS, RB, ZS + 1>::NARROW) masked_taps(c, wm);
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) corner_row(c, k, j, cv[k][j][0], cv[k][j][1], wm);
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
    finish_point(pt, dgz, dgy, dgx);
  };
  // (Rolling steps hold ~N/D records for 1024 threads -- c4: 125.  Four threads per record, one per (z, y) row, their partial
  // sums added over DPP quad permutes, measured SLOWER: k_gather_hw<128,1,8> 37.9 -> 40.1 us.  The step is a chain of phases
  // -- plane load, H pass, barrier, gather, barrier -- of which the gather's arithmetic is the smallest part;
  // profiles/r04_patches/, profiles/LAB_NOTES.md.)
  if (!DPC_ABL(10)) {
    if (GS > 0 && cells.nblk <= DPC_WAVE) for_each_record_flat(cells, b, reinterpret_cast<const int*>(red + kRedTab), gather);
    else for_each_record(cells, b, zfirst, min(zfirst + Zs, D), gather);
  }
  if constexpr (kRolls) {
    using Geo = BwdGeo<GS, RB, 2>;
    constexpr int MW1 = GS * (GS / 32), MPT1 = (MW1 + Geo::NT - 1) / Geo::NT;  // clamp-mask words of one plane
    const uint32_t* mask32 = reinterpret_cast<const uint32_t*>(mrow);
    uint32_t* mlds = reinterpret_cast<uint32_t*>(red + kRedMask);
    int* tab = reinterpret_cast<int*>(red + kRedTab);
    const bool flat = cells.nblk <= DPC_WAVE;
    for (int l = 1; l < roll && (down ? zfirst - l >= z0 : z0 + l < D); ++l) {
      __syncthreads();  // the last layer is gathered: the plane it does not share with this one, and the record table, are free
      const int zl = down ? zfirst - l : z0 + l;      // this step's layer: planes zl and zl + 1
      const int znew = down ? zl : zl + 1;            // ... of which this one is not in LDS yet
      const int buf = (znew - zfirst) & 1;            // plane z lives in buffer (z - zfirst) & 1: the freed one
      const bool present = znew < D;
      RecordRange rr{0, 0};
      if (flat) rr = load_record_range(cells, b, zl, zl + 1);
      const ptrdiff_t rel = (ptrdiff_t)znew - zfirst; // planes relative to the pointers of the first step
      uint32_t mreg[MPT1];
#pragma unroll
      for (int it = 0; it < MPT1; ++it) {
        const int w = tid + it * Geo::NT;
        mreg[it] = (w < MW1 && present) ? mask32[rel * MW1 + w] : 0u;
      }
      hpass_global<Geo, GS, RB, 1>(src + rel * HW, present ? 1 : 0, taps_adj, [&](int, int y, int x, f32x2 val) {
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
      else for_each_record(cells, b, zl, zl + 1, gather);
    }
  }
  DPC_STAMP(11);
  if (bk.x == 0 && !shared_points)
    for_each_record(cells, b, D, D + 1, [&](const PointRec&, const int4* aux) {
      const int i = aux->w;
      dcloud[3 * i + 0] = 0.f; dcloud[3 * i + 1] = 0.f; dcloud[3 * i + 2] = 0.f;
    });

  float vals[13];
#pragma unroll
  for (int i = 0; i < 9; ++i) vals[i] = g.m[i];
  vals[9] = g.dt[0]; vals[10] = g.dt[1]; vals[11] = g.dt[2]; vals[12] = g.df;
  if (DPC_ABL(13)) { if (vals[0] == 123.f) dsmall[0] = vals[1]; return; }
  const double tot = block_sum13_fixed(vals, red, tid, nthr, (t != nullptr || f != nullptr) ? 13 : 9);
  DPC_STAMP(12);
  if (tid == 0 && bk.x == 0) {  // the occupancy-scale gradient: the column kernel's per-tile partials, in tile order
    float ds = 0.f;
    for (int i = 0; i < n_ds_part; ++i) ds += ds_part[(size_t)b * n_ds_part + i];
    dsmall[(size_t)DPC_COL_DS * P.B + b] = ds * upstream;
  }
  camgrad_publish(tot, tid, cam_raw, P.B, b, bk.x, bk.nx, cg_part, cg_count, dsmall, t != nullptr, f != nullptr);
  DPC_STAMP(13);
}

// d(point sets) += the fixed-point sums the gather left (shared point sets with several writers); a set that received a
// contribution with no fixed-point value (NaN, Inf, out of range: its poison word is set) gets NaN instead
__global__ __launch_bounds__(256) void k_fixed_to_dpc(const unsigned long long* __restrict__ acc, float* __restrict__ dpc, size_t n,
                                                       size_t per_set) {
  const unsigned int* poison = reinterpret_cast<const unsigned int*>(acc + n);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dpc[i] = poison[i / per_set] != 0u ? __int_as_float(0x7fc00000) : dpc[i] + (float)((double)(long long)acc[i] * kGradFixInv);
}

// several clouds write into one point set's gradient (the kernel's own test, see k_gather_hw)
bool fixed_point_gradients(const DpcParams* p, const LossArgs& la) {
  const bool shared = p->point_replicas > 1 || p->point_index != nullptr;
  const int reps = p->point_replicas > 1 ? p->point_replicas : 1;
  const bool single_writer = winners_only(la) && shared && reps == la.K && p->point_index == nullptr;
  return shared && !single_writer;
}

template <int GS, int ZS, int RB>
int launch_gather_fast(const DpcParams* p, Cells cells, const float* pc, const float* q, const float* t, const float* f,
                       const float* kxy, const TapPlan& pxy, const float* dT, const uint64_t* mask,
                       const float* ds_part, int ntile, float* dpc, float* dsmall, double* cg_part, unsigned int* cg_count,
                       const LossArgs& la, hipStream_t st, unsigned long long* dpc_fixed) {
  using Geo = BwdGeo<GS, RB, ZS + 1>;
  // slab + scratch tail: reduction floats, record table, staged mask words
  constexpr size_t lds = (((Geo::slab_floats(ZS + 1) + 3) / 4) * 4 + kRedMask + (ZS + 1) * GS * (GS / 32)) * sizeof(float);
  static_assert(lds <= kLdsLimit, "backward slab does not fit LDS");
  static_assert(kRedTab >= 13 * (Geo::NT / DPC_WAVE) && kRedMask >= kRedTab + kTabInts, "scratch tail layout");
  auto kern = k_gather_hw<GS, ZS, RB>;
  static LdsLimit limit;
  int rc = set_lds(kern, lds, limit);
  if (rc != DPC_OK) return rc;
  // one-layer slabs roll over several layers per workgroup (see the kernel): as many as still leave a workgroup per CU
  const bool wo = winners_only(la) && (p->point_replicas > 1 || p->point_index != nullptr);  // the kernel's own test
  const int clouds = wo ? p->B / la.K : p->B;  // clouds that get workgroups
  int roll = 1;
  if (ZS == 1 && RB > 0)
    for (int c = 2; c <= 16; c *= 2)
      if (p->D % c == 0 && (size_t)(p->D / c) * clouds >= (size_t)kNumCUs) roll = c;
  const int nslab = (p->D + ZS - 1) / ZS;
  DPC_LAUNCH("k_gather_hw", kern, dim3(((nslab + roll - 1) / roll) * clouds), dim3(Geo::NT), lds, st, *p, cells, pc, q, t, f,
             make_taps<RB>(kxy, pxy, true), ZS == 1 ? roll : ZS, dT, mask, ds_part, ntile, dpc, dsmall, cg_part, cg_count, la, dpc_fixed);
  return launch_ok();
}

template <int RB>
int launch_gather_rb(const DpcParams* p, Cells cells, const float* pc, const float* q, const float* t, const float* f,
                  const float* kxy, const TapPlan& pxy, const float* dT, const uint64_t* mask, const float* ds_part,
                  int ntile, float* dpc, float* dsmall, double* cg_part, unsigned int* cg_count, const LossArgs& la,
                  hipStream_t st, unsigned long long* dpc_fixed) {
  if (p->H == p->W) {
    // 64-wide: thick slabs when they fill the chip, thinner ones when few clouds have backward work (dpc_kernels.h, kBwdZs64)
    const bool wo64 = winners_only(la) && (p->point_replicas > 1 || p->point_index != nullptr);  // the kernel's own test
    const bool thick = (size_t)(wo64 ? p->B / la.K : p->B) * ((p->D + kBwdZs64 - 1) / kBwdZs64) >= (size_t)kNumCUs;
    if constexpr (RB <= 4) {
      if (p->H == 32) return launch_gather_fast<32, 4, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, cg_part, cg_count, la, st, dpc_fixed);
      if (p->H == 128) return launch_gather_fast<128, 1, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, cg_part, cg_count, la, st, dpc_fixed);
      if (p->H == 64 && thick) return launch_gather_fast<64, kBwdZs64, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, cg_part, cg_count, la, st, dpc_fixed);
      if (p->H == 64) return launch_gather_fast<64, kBwdZs64Thin, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, cg_part, cg_count, la, st, dpc_fixed);
    } else if constexpr (RB <= 10) {
      if constexpr (kBwdNarrowPad)
        if (p->H == 64 && thick) return launch_gather_fast<64, kBwdZs64, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, cg_part, cg_count, la, st, dpc_fixed);
      if (p->H == 64) return launch_gather_fast<64, kBwdZs64Wide, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, cg_part, cg_count, la, st, dpc_fixed);
      if (p->H == 128) return launch_gather_fast<128, 1, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, cg_part, cg_count, la, st, dpc_fixed);  // c4: sigma_rel 1.28 -> radius 8
      if (p->H == 32) return launch_gather_fast<32, 4, RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, cg_part, cg_count, la, st, dpc_fixed);
    }
  }
  const int fit = planes_fit(p);
  if (fit < 2) return DPC_ERR_LDS;
  const int Zs = std::min(fit - 1, std::max(1, (p->D + 7) / 8));
  const size_t lds = ((size_t)(Zs + 1) * p->H * (p->W | 1) + kRedFloats) * sizeof(float);
  auto kern = k_gather_hw<0, 0, RB>;
  static LdsLimit limit;
  int rc = set_lds(kern, lds, limit);
  if (rc != DPC_OK) return rc;
  const bool wo = winners_only(la) && (p->point_replicas > 1 || p->point_index != nullptr);  // the kernel's own test
  DPC_LAUNCH("k_gather_hw", kern, dim3(((p->D + Zs - 1) / Zs) * (wo ? p->B / la.K : p->B)), dim3(slab_threads(p)), lds, st, *p, cells, pc, q, t, f,
             make_taps<RB>(kxy, pxy, true), Zs, dT, mask, ds_part, ntile, dpc, dsmall, cg_part, cg_count, la, dpc_fixed);
  return launch_ok();
}

}  // namespace

int launch_gather(int bucket, const DpcParams* p, Cells cells, const float* pc, const float* q, const float* t, const float* f,
                  const float* kxy, const TapPlan& pxy, const float* dT, const uint64_t* mask, const float* ds_part, int ntile,
                  float* dpc, float* dsmall, double* cg_part, unsigned int* cg_count, const LossArgs& la, hipStream_t st,
                  unsigned long long* dpc_fixed) {
  const bool fixed = fixed_point_gradients(p, la) && p->N > 0;
  const int reps = p->point_replicas > 1 ? p->point_replicas : 1;
  const size_t nfix = (size_t)(p->B / reps) * points_per_set(*p) * 3;
  if (fixed) {
    if (dpc_fixed == nullptr) return DPC_ERR_NULL;
    // the sums and the poison words behind them; a kernel, not a memset node (dpc_common.h)
    if (!zero_words_async(dpc_fixed, 2 * nfix + (size_t)(p->B / reps), st)) return DPC_ERR_LAUNCH;
  }
  int rc = DPC_OK;
#define DPC_GATHER(RB) rc = launch_gather_rb<RB>(p, cells, pc, q, t, f, kxy, pxy, dT, mask, ds_part, ntile, dpc, dsmall, cg_part, cg_count, la, st, dpc_fixed)
  DPC_FOR_BUCKET(bucket, DPC_GATHER)
#undef DPC_GATHER
  if (rc != DPC_OK || !fixed) return rc;
  DPC_LAUNCH("k_fixed_to_dpc", k_fixed_to_dpc, dim3((unsigned)std::min<size_t>((nfix + 255) / 256, 2048)), dim3(256), 0, st, dpc_fixed, dpc, nfix, (size_t)points_per_set(*p) * 3);
  return launch_ok();
}

}  // namespace dpck

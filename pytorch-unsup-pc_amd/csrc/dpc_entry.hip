// C ABI of the fused path (include/dpc_render.h): argument checks, tap planning, and the launch sequence
//   forward : k_locate -> k_splat_hw -> k_zcol_fwd | k_zcol_fwdbwd [-> k_loss_finalize]
//   backward: [k_zcol_bwd ->] k_gather_hw
// The kernels live in dpc_slab_fwd.hip, dpc_column.hip and dpc_slab_bwd.hip.
#include "dpc_kernels.h"

using namespace dpck;

#ifdef DPC_ABLATE
extern "C" int dpc_debug_set_ablate_fwd(int), dpc_debug_set_ablate_col(int), dpc_debug_set_ablate_bwd(int), dpc_debug_set_ablate_xl(int);
extern "C" int dpc_debug_set_stamps_fwd(void*), dpc_debug_set_stamps_col(void*), dpc_debug_set_stamps_bwd(void*), dpc_debug_set_stamps_xl(void*);
extern "C" int dpc_debug_set_ablate(int v) { return dpc_debug_set_ablate_fwd(v) | dpc_debug_set_ablate_col(v) | dpc_debug_set_ablate_bwd(v) | dpc_debug_set_ablate_xl(v); }
extern "C" int dpc_debug_set_stamps(void* p) { return dpc_debug_set_stamps_fwd(p) | dpc_debug_set_stamps_col(p) | dpc_debug_set_stamps_bwd(p) | dpc_debug_set_stamps_xl(p); }
#endif

extern "C" {

int dpc_taps_bucket(const float* host_kern, int taps) {
  if (taps < 0 || taps > DPC_MAX_TAPS || (taps > 0 && (taps % 2 == 0 || !host_kern))) return -1;
  return plan_taps(host_kern, taps).bucket;
}

size_t dpc_mask_words_per_plane(const DpcParams* p) { return p ? ((size_t)p->H * p->W + 63) / 64 : 0; }

size_t dpc_cells_bytes(const DpcParams* p) {
  if (validate(p) != DPC_OK) return 0;
  return (size_t)p->B * num_chunks(p->N) * chunk_bytes(p->D);
}

// Can the grid of `p` be served: forward (one H x W plane in LDS), or forward AND backward (a cell layer plus its halo: two
// planes)?  Square 32 / 64 / 128 grids have kernels of their own.  Lets a caller that knows a backward will follow refuse the
// call up front instead of finding out in the backward.
int dpc_check_grid(const DpcParams* p, int with_backward) {
  const int rc = validate(p);
  if (rc != DPC_OK) return rc;
  if (p->H == p->W && (p->H == 32 || p->H == 64 || p->H == 128)) return DPC_OK;
  return planes_fit(p) < (with_backward ? 2 : 1) ? DPC_ERR_LDS : DPC_OK;
}

size_t dpc_workspace_bytes(const DpcParams* p) {
  if (validate(p) != DPC_OK) return 0;
  return ws_total_bytes(p);   // layout: workspace_view() in dpc_kernels.h
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
  if (p->B == 0) return DPC_OK;
  if (!vox || (p->N > 0 && (!tr || !cells))) return DPC_ERR_NULL;
  hipStream_t st = (hipStream_t)stream;
  if ((rc = launch_locate(p, tr_is_f64 ? 2 : 1, tr, nullptr, nullptr, nullptr, nullptr, cells, st)) != DPC_OK) return rc;
  const TapPlan none{0, 0, 0};
  return launch_splat(0, p, cells_view(p, cells), nullptr, none, vox, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st);
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
  if (p->B == 0) return DPC_OK;  // no clouds (an empty shard): every array is empty, its pointer may be NULL
  if (!q || !grid_wh || !mask || !proj) return DPC_ERR_NULL;
  if (p->N > 0 && (!pc || !cells)) return DPC_ERR_NULL;
  if ((p->taps_xy > 0 && !host_kern_xy) || (p->taps_z > 0 && !host_kern_z)) return DPC_ERR_NULL;
  const TapPlan pxy = plan_taps(host_kern_xy, p->taps_xy), pz = plan_taps(host_kern_z, p->taps_z);
  if (pxy.bucket < 0) return DPC_ERR_TAPS;  // in-LDS passes need a radius bucket; caller composes the stage ops
  float* Tbuf = grid_wh;
  if ((rc = launch_locate(p, 0, pc, q, t, f, tr_pc, cells, st)) != DPC_OK) return rc;
  const Cells cv = cells_view(p, cells);
  // fused loss with one candidate per sample and a backward workspace: the ray-march kernel also runs the column
  // backward; its per-cloud sum-and-count words live behind the ds partials and are zeroed by the slab kernel
  const bool fuse_bwd = bwd_workspace != nullptr && bwd_dsmall != nullptr && la.loss_direct != nullptr &&
                        can_fuse_column_backward(p, pz, la.K, Tbuf, proj, la.gt, bwd_workspace);
  Workspace w{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (fuse_bwd) w = workspace_view(p, bwd_workspace);
  float* ds_part = w.ds_part;
  unsigned long long* tickets = w.tickets;  // B per-cloud words + 1 batch word behind the ds partials, 8-byte aligned
  if ((rc = launch_splat(pxy.bucket, p, cv, host_kern_xy, pxy, raw, Tbuf, mask, la.sse, la.loss_direct, la.winner_out, tickets, st)) != DPC_OK)
    return rc;
  if (fuse_bwd)
    return launch_zcol_fwdbwd(p, host_kern_z, pz, Tbuf, s, proj, w.dT, ds_part, col_tiles(p), tickets, bwd_dsmall, w.cg_count,
                              la, st);
  return launch_zcol_fwd(p, host_kern_z, pz, Tbuf, s, smoothed, proj, trans, la, st);
}

int project_bwd_impl(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f, const float* s,
                     const float* host_kern_xy, const float* host_kern_z, const void* cells, const float* grid_wh,
                     const uint64_t* mask, const float* dproj, const float* proj, const float* trans, const LossArgs& la,
                     float* dpc, float* dsmall, void* workspace, const float* dgrid_wh, hipStream_t st) {
  const bool column_done = la.scale_in_gather != 0;  // dT, ds partials and zeroed dsmall come from the forward
  int rc = validate(p);
  if (rc != DPC_OK) return rc;
  if (p->B == 0) return DPC_OK;  // no clouds: nothing to write
  if (!q || !grid_wh || !mask || !dsmall || !workspace) return DPC_ERR_NULL;
  if (!column_done && (la.gt == nullptr ? !dproj : (!proj || !la.winner))) return DPC_ERR_NULL;
  if (p->N > 0 && (!pc || !cells || !dpc)) return DPC_ERR_NULL;
  if ((p->taps_xy > 0 && !host_kern_xy) || (p->taps_z > 0 && !host_kern_z)) return DPC_ERR_NULL;
  const TapPlan pxy = plan_taps(host_kern_xy, p->taps_xy), pz = plan_taps(host_kern_z, p->taps_z);
  if (pxy.bucket < 0) return DPC_ERR_TAPS;
  const Workspace w = workspace_view(p, workspace);
  const int ntile = col_tiles(p);
  if (!column_done &&
      (rc = launch_zcol_bwd(p, host_kern_z, pz, grid_wh, s, dproj, proj, trans, w.dT, w.ds_part, dsmall, w.cg_count, dgrid_wh, la, st)) != DPC_OK)
    return rc;
  return launch_gather(pxy.bucket, p, cells_view(p, cells), pc, q, t, f, host_kern_xy, pxy, w.dT, mask, w.ds_part, ntile, dpc,
                       dsmall, w.cg_part, w.cg_count, la, st, w.dpc_fixed);
}

const LossArgs kNoLoss{nullptr, nullptr, nullptr, nullptr, 1, 1.0f, nullptr, nullptr, 0, nullptr};

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
                    const float* grid_wh, const uint64_t* mask, const float* trans, const float* dproj,
                    const float* dgrid_wh, float* dpc, float* dsmall, void* workspace, void* stream) {
  if (!dproj && !(p && p->B == 0)) return DPC_ERR_NULL;
  return project_bwd_impl(p, pc, q, t, f, s, host_kern_xy, host_kern_z, cells, grid_wh, mask, dproj, nullptr, trans,
                          kNoLoss, dpc, dsmall, workspace, dgrid_wh, (hipStream_t)stream);
}

int dpc_project_loss_fwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                         const float* s, const float* host_kern_xy, const float* host_kern_z, const float* gt,
                         int num_candidates, float* tr_pc, void* cells, float* grid_wh, uint64_t* mask, float* proj,
                         float* trans, float* sse, float* sse_tiles, float* loss, int32_t* winner, void* bwd_workspace,
                         float* bwd_dsmall, int* column_backward_done, void* stream) {
  if (!p || !loss) return DPC_ERR_NULL;
  if (num_candidates < 1 || p->B % num_candidates != 0) return DPC_ERR_SHAPE;
  if (p->B == 0)   // no clouds (an empty shard): the loss of nothing is 0
    return zero_words_async(loss, 1, (hipStream_t)stream) ? DPC_OK : DPC_ERR_LAUNCH;
  if (!gt || !sse || !winner) return DPC_ERR_NULL;
  const int S = p->B / num_candidates;
  // one candidate per sample AND a backward workspace: the fused ray march sums the loss itself (64-bit fixed point)
  const TapPlan pz = plan_taps(host_kern_z, p->taps_z);
  const bool fuse = num_candidates == 1 && bwd_workspace && bwd_dsmall &&
                    can_fuse_column_backward(p, pz, num_candidates, grid_wh, proj, gt, bwd_workspace);
  if (!fuse && !sse_tiles) return DPC_ERR_NULL;  // the unfused ray march leaves per-tile partials for the finalize launch
  const LossArgs la{gt, sse, nullptr, nullptr, num_candidates, S > 0 ? 1.0f / (float)S : 0.f,
                    fuse ? loss : nullptr, fuse ? winner : nullptr, 0, fuse ? nullptr : sse_tiles};
  if (column_backward_done) *column_backward_done = fuse ? 1 : 0;
  int rc = project_fwd_impl(p, pc, q, t, f, s, host_kern_xy, host_kern_z, tr_pc, cells, nullptr, grid_wh, nullptr, mask,
                            proj, trans, la, fuse ? bwd_workspace : nullptr, fuse ? bwd_dsmall : nullptr,
                            (hipStream_t)stream);
  if (rc != DPC_OK || p->B == 0 || fuse) return rc;
  return launch_loss_finalize(sse_tiles, col_tiles(p), sse, S, num_candidates, la.inv_S, loss, winner, (hipStream_t)stream);
}

int dpc_project_loss_step(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                          const float* s, const float* host_kern_xy, const float* host_kern_z, const float* gt,
                          int num_candidates, void* cells, float* grid_wh, uint64_t* mask, float* proj, float* trans,
                          float* sse, float* sse_tiles, float* loss, int32_t* winner, void* workspace, float* fwd_dsmall,
                          const float* dloss, float* dpc, float* dsmall, void* stream) {
  if (!workspace || !fwd_dsmall) return (p && p->B == 0) ? dpc_project_loss_fwd(p, pc, q, t, f, s, host_kern_xy, host_kern_z, gt, num_candidates, nullptr, cells, grid_wh, mask, proj, trans, sse, sse_tiles, loss, winner, nullptr, nullptr, nullptr, stream) : DPC_ERR_NULL;
  const int K = num_candidates;
  // K candidates, a column depth and a z kernel the specialised column backward covers: the min-of-K selection rides in
  // that launch (k_zcol_bwd) instead of taking one of its own between forward and backward
  if (p && K > 1 && K <= kColThreads && p->B > 0 && p->B % K == 0 && gt && sse && sse_tiles && loss && winner &&
      (p->D == 32 || p->D == 64 || p->D == 128) && plan_taps(host_kern_z, p->taps_z).bucket >= 0) {
    const int S = p->B / K;
    const float inv_S = 1.0f / (float)S;
    const LossArgs lf{gt, sse, nullptr, nullptr, K, inv_S, nullptr, nullptr, 0, sse_tiles, nullptr, nullptr, 0};
    int rc = project_fwd_impl(p, pc, q, t, f, s, host_kern_xy, host_kern_z, nullptr, cells, nullptr, grid_wh, nullptr, mask, proj,
                              trans, lf, nullptr, nullptr, (hipStream_t)stream);
    if (rc != DPC_OK) return rc;
    const LossArgs lb{gt, sse, winner, dloss, K, inv_S, nullptr, nullptr, 0, sse_tiles, winner, loss, col_tiles(p)};
    return project_bwd_impl(p, pc, q, t, f, s, host_kern_xy, host_kern_z, cells, grid_wh, mask, nullptr, proj, trans, lb, dpc,
                            dsmall, workspace, nullptr, (hipStream_t)stream);
  }
  int column_done = 0;
  int rc = dpc_project_loss_fwd(p, pc, q, t, f, s, host_kern_xy, host_kern_z, gt, num_candidates, nullptr, cells, grid_wh, mask,
                                proj, trans, sse, sse_tiles, loss, winner, workspace, fwd_dsmall, &column_done, stream);
  if (rc != DPC_OK || p->B == 0) return rc;
  return dpc_project_loss_bwd(p, pc, q, t, f, s, host_kern_xy, host_kern_z, cells, grid_wh, mask, proj, trans, gt, num_candidates,
                              winner, dloss, column_done, dpc, dsmall, workspace, stream);
}

int dpc_project_loss_bwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                         const float* s, const float* host_kern_xy, const float* host_kern_z, const void* cells,
                         const float* grid_wh, const uint64_t* mask, const float* proj, const float* trans,
                         const float* gt, int num_candidates, const int32_t* winner, const float* dloss,
                         int column_backward_done, float* dpc, float* dsmall, void* workspace, void* stream) {
  if (!p) return DPC_ERR_NULL;
  if (num_candidates < 1 || p->B % num_candidates != 0) return DPC_ERR_SHAPE;
  if (p->B == 0) return DPC_OK;  // no clouds: nothing to write
  if (!gt || !winner || !proj) return DPC_ERR_NULL;
  const int S = p->B / num_candidates;
  const LossArgs la{gt, nullptr, winner, dloss, num_candidates, S > 0 ? 1.0f / (float)S : 0.f, nullptr, nullptr,
                    column_backward_done ? 1 : 0, nullptr};
  return project_bwd_impl(p, pc, q, t, f, s, host_kern_xy, host_kern_z, cells, grid_wh, mask, nullptr, proj, trans, la,
                          dpc, dsmall, workspace, nullptr, (hipStream_t)stream);
}

}  // extern "C"

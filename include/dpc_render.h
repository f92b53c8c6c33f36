/*
 * dpc_render.h -- C ABI of libdpc_render.so: the differentiable point-cloud projection of
 * NiteshBharadwaj/pytorch-unsup-pc as hand-written HIP kernels for MI355X (gfx950).
 *
 * The reference has no native code and no FFI: its hot path is Python calling ATen
 * (SURVEY.md section 2b).  The entry points below are therefore what a binding for that path would bind --
 * one call per reference function, plus the fused forward/backward that replaces the whole of
 * pointcloud_project_fast and its autograd graph.  Each entry cites the reference code it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - plain C, no C++/torch types; all pointers are DEVICE pointers unless named host_*;
 *   - every buffer is allocated by the caller; the library never allocates device memory and never
 *     synchronises the stream (so calls can be captured into a hipGraph);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - return value: DPC_OK (0) or a negative DPC_ERR_* code; dpc_strerror() names it;
 *   - B == 0 (no clouds: an empty shard) is valid everywhere: nothing is launched, array pointers may be NULL, and
 *     dpc_project_loss_fwd writes loss = 0;
 *   - re-entrant, no global state; arithmetic type fp32 (the ray-march transmittance product runs in fp64
 *     registers); tensors are dense row-major with the shapes stated per argument;
 *   - optional inputs (t, f, s) and optional outputs are NULL when absent;
 *   - grids, silhouettes and the workspace are read and written with 8- and 16-byte accesses: their base pointers must be
 *     16-byte aligned (anything hipMalloc returns is); the big grids a launch writes and does not read again (the W/H-
 *     filtered grid, its gradient) are stored write-through, which needs nothing from the caller.
 *
 * Grid: D x H x W voxels (D = vox_size_z or vox_size, H = W = vox_size), voxel (z,y,x) of cloud b at
 * [((b*D + z)*H + y)*W + x].  Point clouds are [B,N,3] xyz, quaternions [B,4] (w,x,y,z) unnormalised.
 */
#ifndef DPC_RENDER_H
#define DPC_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPC_ABI_VERSION 13
#define DPC_MAX_TAPS 63 /* longest 1-D smoothing kernel accepted (pc_gauss_kernel_size) */
/* Size limits, checked by every entry point (DPC_ERR_SHAPE): grid sides <= 1024 (10-bit cell indices in a point record),
 * B <= 65535, and N <= DPC_MAX_POINTS points per cloud: a voxel's splat weights are summed in 64-bit fixed point with 44
 * fractional bits, so 2^20 - 1 points of weight 1 in ONE voxel is the most that cannot wrap. */
#define DPC_MAX_POINTS 1048575

enum {
  DPC_OK = 0,
  DPC_ERR_NULL = -1,        /* a required pointer is NULL */
  DPC_ERR_SHAPE = -2,       /* B, N, D, H, W out of range */
  DPC_ERR_TAPS = -3,        /* tap count even, negative or > DPC_MAX_TAPS */
  DPC_ERR_LDS = -4,         /* an H x W plane does not fit the 160 KiB LDS tile (H*W too large) */
  DPC_ERR_LAUNCH = -5,      /* hipLaunchKernel failed (hipGetLastError has the cause) */
  DPC_ERR_UNSUPPORTED = -6  /* configuration the reference itself cannot run (dead branch) */
};

/* Bits of the optional device status word (DpcParams.status). */
enum {
  DPC_STATUS_BAD_INDEX = 1    /* a point_index entry was outside [0, N_src): the point was dropped (the reference's fancy
                               * indexing raises IndexError there, dpc/util/point_cloud_to.py:266-295)                   */
};

/* Geometry and camera constants of one call (dpc/resources/default_config.yaml:77-89 and the cfg fields
 * read at dpc/util/point_cloud_to.py:11-15,128-135; dpc/util/drc.py:52-57,148). */
typedef struct DpcParams {
  int32_t B;               /* clouds in this call (batch_size * step_size * num_candidates)            */
  int32_t N;               /* points per cloud, <= DPC_MAX_POINTS                                       */
  int32_t D, H, W;         /* voxel grid                                                                */
  int32_t taps_xy;         /* length of the x/y Gaussian (odd), 0 = no smoothing (kernel=None / CPU branch) */
  int32_t taps_z;          /* length of the z Gaussian (odd), 0 = no smoothing                          */
  float camera_distance;   /* cfg.camera_distance                                                        */
  float focal_length;      /* cfg.focal_length, used when f == NULL                                      */
  float clip_val;          /* cfg.drc_logsum_clip_val (eps)                                              */
  float max_depth;         /* cfg.max_depth                                                              */
  int32_t point_replicas;  /* R: clouds b*R .. b*R+R-1 share point set b (tf_repeat_0 of the decoded points over views and
                            * pose candidates, dpc/models/model_pc_to.py:302-306) -- then `pc` is [B/R,N,3], read once per
                            * replica instead of being materialised B times, and `dpc` is [B/R,N,3], ZERO-INITIALISED BY
                            * THE CALLER: the replicas' gradients are added into it.  0 or 1: every cloud has its own
                            * points.  Honoured by the fused entry points (dpc_project_*); the stage entry points require
                            * R <= 1. */
  int32_t N_src;           /* points per stored point set when point_index is given (>= 1), otherwise ignored     */
  const int32_t* point_index; /* DEVICE pointer [B,N] int32 | NULL: cloud b projects the points
                            * pc[b/R][point_index[b*N + i]], i < N, of a stored set of N_src points -- every replica
                            * of a point set keeps its own random subset (pc_point_dropout applied after tf_repeat_0,
                            * dpc/models/model_pc_to.py:254-258, 302-306; dpc/util/point_cloud_to.py:269-295) without the
                            * [B,N,3] copies.  Then `pc` is [B/R,N_src,3] and `dpc` is [B/R,N_src,3], ZERO-INITIALISED BY
                            * THE CALLER (the selected points' gradients are added into it; indices may repeat).
                            * An entry outside [0, N_src) never reaches memory: the point is dropped (no contribution, no
                            * gradient) and DPC_STATUS_BAD_INDEX is set in *status when `status` is given.
                            * Honoured by dpc_locate and the fused entry points; the stage entry points require NULL. */
  int32_t* status;         /* DEVICE pointer to one int32 | NULL: DPC_STATUS_* bits are OR-ed into it (never cleared by the
                            * library; the caller zeroes it and reads it at a synchronisation point of its own)           */
  const int32_t* n_live;   /* DEVICE pointer to one int32 | NULL: only the first min(*n_live, N) points of every cloud (of
                            * every point_index row) are live in this call, the rest are skipped without a trace.  N stays the
                            * CAPACITY the buffers and launch grids are sized for -- so a captured HIP graph follows a
                            * scheduled keep-count (dpc/models/model_pc_to.py:68-87, 254-258) from replay to replay.       */
  const float* dev_taps_xy; /* DEVICE pointers to taps_xy / taps_z floats | NULL: when given, the kernels read the tap    */
  const float* dev_taps_z;  /* VALUES from device memory at run time instead of taking them from host_kern_xy / host_kern_z
                            * at launch time.  The host arrays are still required: they select the compiled radius bucket
                            * (dpc_taps_bucket) and must need the same bucket as, or a smaller one than, the device values
                            * ever will.  For captured HIP graphs under a sigma schedule (model_pc_to.py:59-63, 171-179):
                            * dpc_schedule_update rewrites the device values between replays.                              */
} DpcParams;

/* Small per-cloud gradients written by the backward entry points: one buffer of DPC_SMALL_COLS * B floats made of
 * contiguous blocks, so each gradient is a dense tensor of its own: dq [B,4] at float offset DPC_COL_DQ*B,
 * ds [B,1] at DPC_COL_DS*B, dt [B,3] at DPC_COL_DT*B, df [B,1] at DPC_COL_DF*B. */
enum { DPC_COL_DQ = 0, DPC_COL_DS = 4, DPC_COL_DT = 5, DPC_COL_DF = 8, DPC_SMALL_COLS = 12 };

int dpc_abi_version(void);
const char* dpc_strerror(int code);

/* Words of the clamp mask ([B, D, dpc_mask_words_per_plane] uint64; bit i of a plane = voxel y*W+x == i,
 * set where the raw splat value v satisfies 0 <= v <= 1, the pass-through set of torch.clamp's backward). */
size_t dpc_mask_words_per_plane(const DpcParams* p);
/* Binned point records written by the locate kernel and consumed by the slab kernels -- opaque to the caller,
 * dpc_cells_bytes(p) bytes, saved between forward and backward.  Per cloud, ceil(N/256) chunks; each chunk holds
 * its 256 points counting-sorted by z cell: 256 x {int32 code = iz<<20|iy<<10|ix or -1 (out of bounds), 3 x fp32
 * fractional weights encoded so that both r and 1-r keep fp32 relative precision}, 256 x {px, py, pz, int32
 * original point index} sorted alike, (D+2) x uint16 bin offsets (padded to 16 bytes). */
size_t dpc_cells_bytes(const DpcParams* p);
/* First launch of the fused forward on its own: transform (the reference's exact op sequence, see
 * csrc/dpc_common.h) + cell location + per-256-point z sort.  tr_pc [B,N,3] | NULL, cells dpc_cells_bytes(p).
 * Integer/bit-exact against the reference: the parity tests decode `cells` and compare with records computed
 * from the oracle's fp64 tr_pc. */
int dpc_locate(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f, float* tr_pc,
               void* cells, void* stream);
/* Scratch the fused BACKWARD entry points need (one grid-sized fp32 buffer + per-tile partial sums). */
size_t dpc_workspace_bytes(const DpcParams* p);
/* ABI 13.  DPC_OK when the grid of `p` can be served, DPC_ERR_LDS when not: the slab kernels keep whole H x W planes in LDS --
 * one for the forward (planes up to 199 x 199), a cell layer plus its halo for the backward (up to 141 x 141; square 32 / 64 /
 * 128 grids have kernels of their own).  The forward entry points succeed wherever the forward can run; a caller that knows a
 * backward will follow asks with with_backward = 1 and refuses up front (the Python layer does, when gradients are required).
 * The reference takes any vox_size (dpc/util/point_cloud_to.py:11-15); its experiments use 32, 64 and 128. */
int dpc_check_grid(const DpcParams* p, int with_backward);

/* ---------------------------------------------------------------------------------------------------
 * Fused hot path: replaces pointcloud_project_fast (dpc/util/point_cloud_to.py:191-263) =
 * pc_perspective_transform (:118-178) -> pointcloud2voxels3d_fast (:10-87) -> clamp (:201) ->
 * smoothen_voxels3d (:90-103) -> scale+clamp (:218-222) -> drc_projection (dpc/util/drc.py:114-129) ->
 * flip (:242), in three launches (locate + z-sort points -> splat + W/H passes in LDS -> z column pass + DRC).
 *   pc [B,N,3], q [B,4], t [B,3]|NULL, f [B,1]|NULL, s [B,1]|NULL, host_kern_xy[taps_xy], host_kern_z[taps_z]
 *   (HOST pointers: the tap weights travel as kernel arguments)
 * outputs
 *   tr_pc    [B,N,3] (z,y,x) | NULL
 *   cells    dpc_cells_bytes(p) bytes of binned point records (saved for backward)
 *   raw      [B,D,H,W] unclamped splat | NULL (not needed by the backward)
 *   grid_wh  [B,D,H,W] grid after clamp + the W and H passes of the Gaussian (saved for backward, which
 *            recomputes the D pass in registers instead of reading a second grid)
 *   smoothed [B,D,H,W] | NULL: grid after the full Gaussian, before the occupancy scale
 *            (voxels = s ? clamp(s*smoothed,0,1) : smoothed); optional, the hot path does not write it
 *   mask     [B,D,words] uint64 clamp mask (saved for backward)
 *   proj     [B,H,W] silhouette, rows already flipped
 *   trans    [B,H,W] per-ray transmittance prod(1-y) (ray order, not flipped) | NULL; saves the backward a pass
 * ------------------------------------------------------------------------------------------------- */
int dpc_project_fwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                    const float* s, const float* host_kern_xy, const float* host_kern_z, float* tr_pc,
                    void* cells, float* raw, float* grid_wh, float* smoothed, uint64_t* mask, float* proj,
                    float* trans, void* stream);

/* Hand-written backward of the chain above (the reference relies on autograd).
 *   dproj    [B,H,W] gradient w.r.t. the (flipped) silhouette
 *   dgrid_wh [B,D,H,W] | NULL: a gradient arriving at grid_wh itself, added in (callers that derive further outputs --
 *            voxels, drc_probs, proj_depth of the reference's output dict -- from the saved grid_wh instead of running the
 *            chain a second time)
 * outputs
 *   dpc    [B,N,3]
 *   dsmall DPC_SMALL_COLS*B floats in the block layout above (ds/dt/df only meaningful when the matching input
 *          was given); fully overwritten, needs no zeroing by the caller. */
int dpc_project_bwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                    const float* s, const float* host_kern_xy, const float* host_kern_z, const void* cells,
                    const float* grid_wh, const uint64_t* mask, const float* trans /* from fwd, or NULL */,
                    const float* dproj, const float* dgrid_wh, float* dpc, float* dsmall, void* workspace, void* stream);

/* The same chain with the caller's silhouette loss fused in: add_proj_loss /
 * proj_loss_pose_candidates (dpc/models/model_pc_to.py:339-385, 410-440).  Cloud b is pose candidate b % K of
 * sample b / K; gt [B/K, H, W] is the mask already pooled to the silhouette size.  The ray-march kernel
 * accumulates each cloud's sum of squared differences, a one-block finalize picks argmin over K (first minimum)
 * and writes loss = sum_s min_k sse / S.  In the backward dproj = 2 (proj - gt) / S * dloss is formed on the fly
 * (never stored) and losing candidates skip all work -- their gradients are exact zeros.
 *   fwd outputs: proj [B,H,W], trans [B,H,W], sse [B], loss [1], winner [B/K] int32 (+ tr_pc|NULL, cells, grid_wh, mask);
 *                sse_tiles [B, ceil(H*W/256)] scratch: the ray tiles' squared errors, added in tile order by the finalize
 *                launch (no float atomics: sums, winners and loss are the same bits on every run); may be NULL only when
 *                the column backward is fused into the forward (see below), which sums in 64-bit fixed point instead
 *   bwd inputs : dloss = device scalar arriving at `loss` (NULL = 1)
 *
 * Column half of the backward inside the forward (optional): d loss / d proj is linear in dloss, and the forward's
 * ray-march kernel holds each ray's column in registers, so with one candidate per sample (K = 1) it can also run the
 * DRC backward + adjoint D pass for dloss = 1.  Pass bwd_workspace (dpc_workspace_bytes) and bwd_dsmall
 * (DPC_SMALL_COLS*B floats): when the configuration allows it, *column_backward_done is set to 1, the workspace holds
 * dT and the ds partials and bwd_dsmall is zeroed; hand the workspace and the flag to dpc_project_loss_bwd, which then
 * launches the gather kernel only (it multiplies by *dloss).  dpc_project_loss_bwd WRITES dq, ds (and dt, df when t, f
 * are given) into its `dsmall` -- any DPC_SMALL_COLS*B floats, need not be bwd_dsmall; the workspace is left as it was, so
 * the backward may be called again on the same forward.  Pass NULLs / 0 to keep the two halves apart. */
int dpc_project_loss_fwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                         const float* s, const float* host_kern_xy, const float* host_kern_z, const float* gt,
                         int num_candidates, float* tr_pc, void* cells, float* grid_wh, uint64_t* mask, float* proj,
                         float* trans, float* sse, float* sse_tiles, float* loss, int32_t* winner, void* bwd_workspace,
                         float* bwd_dsmall, int* column_backward_done, void* stream);
int dpc_project_loss_bwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                         const float* s, const float* host_kern_xy, const float* host_kern_z, const void* cells,
                         const float* grid_wh, const uint64_t* mask, const float* proj, const float* trans,
                         const float* gt, int num_candidates, const int32_t* winner, const float* dloss,
                         int column_backward_done, float* dpc, float* dsmall, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * The whole step in ONE call: dpc_project_loss_fwd followed by dpc_project_loss_bwd -- four launches (one pose candidate per
 * sample: the column backward is fused into the forward) or six (K candidates: locate, slab, ray march, finalize, column
 * backward and gather of the winners) enqueued back to back by native code.  What a training loop calls once
 * per step instead of replaying a captured HIP graph of the two calls: the same kernels, the same results bit for bit, and
 * no graph (on MI355X / ROCm 7.2 the eager native sequence is 2-3 us per step FASTER than the replayed graph: 55.3 against
 * 57.2-58.2 us at B = 32, N = 8000, 64^3; host cost of the call 18 us, well under the GPU time).
 *   num_candidates, trans, sse_tiles, fwd_dsmall, dsmall, workspace, dloss: as in the two calls above (trans and sse_tiles
 *   are needed when the column backward is not fused into the forward, i.e. always for K > 1; may be NULL for K = 1 on the
 *   32/64/128-deep grids).
 * ------------------------------------------------------------------------------------------------- */
int dpc_project_loss_step(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                          const float* s, const float* host_kern_xy, const float* host_kern_z, const float* gt,
                          int num_candidates, void* cells, float* grid_wh, uint64_t* mask, float* proj, float* trans,
                          float* sse, float* sse_tiles, float* loss, int32_t* winner, void* workspace, float* fwd_dsmall,
                          const float* dloss, float* dpc, float* dsmall, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Stage-level entry points (one per reference function), used for the sub-stage API and to cross-check the
 * fused path.
 * ------------------------------------------------------------------------------------------------- */

/* pc_perspective_transform, quaternion branch (dpc/util/point_cloud_to.py:118-148,169-178;
 * dpc/util/quaternion.py:110-132).  out [B,N,3] in (z,y,x) order. */
int dpc_transform_fwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                      float* out, void* stream);
/* dout [B,N,3] -> dpc [B,N,3], dsmall (DPC_SMALL_COLS*B floats, block layout above: dq, dt, df; overwritten). */
int dpc_transform_bwd(const DpcParams* p, const float* pc, const float* q, const float* t, const float* f,
                      const float* dout, float* dpc, float* dsmall, void* stream);

/* pc_point_dropout's random choice (dpc/util/point_cloud_to.py:269-295: np.random.choice(N, n, replace=False) for every
 * cloud) drawn on the device: out [B, n] int32 = for each of B clouds n DISTINCT indices in [0, N), a uniformly random
 * n-subset, ascending.  seed: TWO int64 words in device memory (any values; equal seeds give equal draws) -- they are read
 * by the kernel, so a captured graph whose seed words are refreshed by a captured RNG node draws anew at every replay.
 * The result is what DpcParams.point_index expects.  No host work, no synchronisation. */
int dpc_point_dropout_indices(int B, int N, int n, const int64_t* seed, int32_t* out, void* stream);

/* Schedules under HIP-graph replay (dpc/models/model_pc_to.py:59-87, 171-179, 254-258: sigma_rel(step) and the dropout
 * keep-probability are recomputed every step).  A captured graph freezes kernel ARGUMENTS, so the per-step values live in
 * device memory instead: dev_taps_xy[taps_xy], dev_taps_z[taps_z] (DpcParams.dev_taps_*) and n_live[1] (DpcParams.n_live,
 * the n_live argument of dpc_point_dropout_indices_live).  dpc_schedule_update writes new values with ONE tiny launch on
 * `stream` (the values travel as kernel arguments: no pinned staging buffer, nothing to keep alive); enqueue it in front of
 * every replay.  Any of the three destinations may be NULL.  dpc_taps_bucket: the compiled radius bucket a 1-D kernel needs
 * (after dropping outer taps that cannot change an fp32 result), -1 when it is beyond the fused kernels -- a captured graph
 * stays valid while the bucket of the new taps is <= the bucket it was captured with, and is fastest when equal. */
int dpc_schedule_update(const float* host_kern_xy, int taps_xy, const float* host_kern_z, int taps_z, int n_live,
                        float* dev_taps_xy, float* dev_taps_z, int32_t* dev_n_live, void* stream);
int dpc_taps_bucket(const float* host_kern, int taps);
/* dpc_point_dropout_indices with the keep-count read on the device: rows of `n` slots (the capacity), the first
 * min(*n_live, n) of each filled, ascending.  n_live NULL = n. */
int dpc_point_dropout_indices_live(int B, int N, int n, const int32_t* n_live, const int64_t* seed, int32_t* out, void* stream);

/* pointcloud2voxels3d_fast (dpc/util/point_cloud_to.py:10-87): trilinear scatter of already-transformed
 * points tr [B,N,3] (z,y,x; fp32, or fp64 when tr_is_f64 -- the reference's direct callers pass fp64) into
 * vox [B,D,H,W] (overwritten).  cells (dpc_cells_bytes(p) bytes) is scratch for the point records. */
int dpc_splat_fwd(const DpcParams* p, const void* tr, int tr_is_f64, void* cells, float* vox, void* stream);
/* backward of the scatter: gather dvox [B,D,H,W] at the 8 corners -> dtr [B,N,3] (fp32). */
int dpc_splat_bwd(const DpcParams* p, const void* tr, int tr_is_f64, const float* dvox, float* dtr, void* stream);

/* smoothen_voxels3d (dpc/util/point_cloud_to.py:90-103): zero-padded separable correlation along W, H, D.
 * `transpose` != 0 applies the adjoint (flipped taps), i.e. the backward.  in/out [B,D,H,W]; tmp same size.
 * p->taps_xy == 0 (or p->taps_z == 0) leaves that group of axes alone: the D pass by itself finishes a grid that already
 * went through the W and H passes (the grid_wh of dpc_project_fwd). */
int dpc_smooth(const DpcParams* p, const float* host_kern_xy, const float* host_kern_z, int transpose,
               const float* in, float* out, float* tmp, void* stream);

/* drc_projection / drc_event_probabilities / drc_depth_projection (dpc/util/drc.py:48-129,145-160).
 * vox [B,D,H,W] -> proj [B,H,W] | NULL, probs [D+1,B,H,W] | NULL, depth [B,H,W] | NULL.
 * No row flip here (the reference flips in pointcloud_project_fast, point_cloud_to.py:239-242). */
int dpc_drc_fwd(const DpcParams* p, const float* vox, float* proj, float* probs, float* depth, void* stream);
/* dproj [B,H,W]|NULL, dprobs [D+1,B,H,W]|NULL, ddepth [B,H,W]|NULL -> dvox [B,D,H,W]. */
int dpc_drc_bwd(const DpcParams* p, const float* vox, const float* dproj, const float* dprobs,
                const float* ddepth, float* dvox, void* stream);

/* Caller-side silhouette loss fused with its gradient: add_proj_loss /
 * proj_loss_pose_candidates (dpc/models/model_pc_to.py:339-385, 410-440).  gt [S, n_pix] is the ground-truth
 * mask already pooled to the silhouette size, pred [S*K, n_pix] the K candidate silhouettes per sample.
 * Per sample the candidate with the smallest sum of squared differences wins (first minimum, like
 * torch.argmin); loss = sum over winners (gt-pred)^2 / S.  Outputs: loss_part [S] (sum them for the loss),
 * winner [S], dpred [S*K, n_pix] = d loss / d pred (zero for losing candidates). */
int dpc_silhouette_loss(const float* gt, const float* pred, int S, int K, int n_pix, float* loss_part, int32_t* winner,
                        float* dpred, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Evaluation side: point_cloud_distance (dpc/util/point_cloud_distance.py:25-40), the kernel of
 * the Chamfer evaluation (dpc/run/eval_chamfer_to.py:24-44).  For every source point vs[i] ([ns,3]) the nearest target
 * vt[j] ([nt,3]): idx[i] = first j minimising dist = sqrt(sum((vt[j]-vs[i])^2)) (int64, like torch.argmin),
 * min_dist[i] that distance, proj[i] = vt[idx[i]].  fp32, or fp64 when is_f64 (the evaluation runs in fp64); proj,
 * min_dist, idx may each be NULL.  workspace: dpc_nearest_workspace_bytes(ns, nt, is_f64) bytes of scratch.
 * ------------------------------------------------------------------------------------------------- */
size_t dpc_nearest_workspace_bytes(int ns, int nt, int is_f64);
int dpc_point_cloud_distance(const void* vs, const void* vt, int ns, int nt, int is_f64, void* proj, void* min_dist,
                             int64_t* idx, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Opt-in measurement aid (nothing in the reference corresponds to it).  After dpc_profile_enable(capacity)
 * every launch of the fused path is bracketed by hipEvents on its stream; synchronise the stream, then read
 * dpc_profile_count() entries with dpc_profile_get(i, &kernel_name, &milliseconds).  Off by default; the only
 * global state in the library; not usable while a hipGraph is being captured.
 * ------------------------------------------------------------------------------------------------- */
int dpc_profile_enable(int capacity);
int dpc_profile_disable(void);
int dpc_profile_count(void);
int dpc_profile_get(int i, const char** name, float* ms);
/* What an EMPTY begin/end event pair reads on `stream` (synchronising; call outside timed regions): subtract it from
 * dpc_profile_get's figures to compare with rocprofv3's kernel durations. */
int dpc_profile_pair_overhead(void* stream, int pairs, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* DPC_RENDER_H */

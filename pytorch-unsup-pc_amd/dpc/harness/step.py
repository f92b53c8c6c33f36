"""One training step of the reference's unsupervised shape-and-pose model around dpc.render.

Follows ModelPointCloud.forward / get_loss (dpc/models/model_pc_to.py:289-336, 339-408, 410-489) and the loop body
of dpc/run/train_to.py:110-134 for the live configuration of the experiments (predict_pose, K pose candidates with a
student, learned occupancy scale, no rgb / depth / drc losses, no translation, fixed focal length):

    images [B*V,3,S,S] -> encoder -> ids [B*V,z]; the first view's id of every object -> decoder -> points [B,N,3]
    pose FC of every image -> K candidate quaternions + 1 student quaternion per image
    points, scales repeated V*K times (candidate-minor), optional point dropout
    renderer + min-of-K silhouette loss in ONE call (dpc.render.pointcloud_project_loss)
    student loss against the winning candidate, (proj + student) * proj_weight, backward, Adam
"""
import numpy as np
import torch
import torch.nn.functional as F

import dpc.render as R

from .nets import StepNets


def pooled_masks(masks, size):
    """[M,1,Hm,Wm] masks -> [M,size,size,1], average-pooled like add_proj_loss (model_pc_to.py:346-356)."""
    if masks.shape[2] < size:
        raise ValueError("GT size should not be smaller than the prediction size")
    if masks.shape[2] > size:
        masks = F.avg_pool2d(masks, masks.shape[2] // size)
    return masks.permute(0, 2, 3, 1).contiguous()


def student_loss(poses, student, winner, num_candidates, weight):
    """add_student_loss (model_pc_to.py:442-489), rotation-difference form: 1 - <teacher, student>_w^2, teachers detached.

    The reference builds diff = normalise(teacher * conj(student)) and reads its w component.  That component is
    <teacher, student> / (|teacher| |student|) (the Hamilton product's norm is the product of the norms), which is what is
    computed here, in float64 like the reference: a dozen launches forward + backward instead of eighty-five."""
    teachers = poses.reshape(-1, num_candidates, 4)
    pick = winner.long().view(-1, 1, 1).expand(-1, 1, 4)
    t = teachers.gather(1, pick).squeeze(1).detach().double()
    s = student.double()
    dot = (t * s).sum(-1)
    norm2 = (t * t).sum(-1) * (s * s).sum(-1)
    return (1.0 - dot * dot / norm2).sum() / winner.shape[0] * weight


def device_point_dropout(points, keep_prob, generator=None):
    """pc_point_dropout (point_cloud_to.py:269-295) without the host: int(N*keep) distinct random points per cloud,
    chosen on the device (dpc.render.point_dropout_indices).  Same distribution as the reference's
    np.random.choice(replace=False), different random stream.  Materialises the kept points; the
    training step below hands the INDICES to the renderer instead and keeps the point sets shared."""
    B, N = points.shape[0], points.shape[1]
    idx = R.point_dropout_indices(B, N, keep_prob, points.device, generator)
    return points.gather(1, idx.long().unsqueeze(-1).expand(B, idx.shape[1], 3))


class TrainStep:
    def __init__(self, cfg, device, lr=1e-4, device_dropout=False, capturable=False, fused_adam=None):
        """capturable: build Adam with its step counters on the device, so that the whole step (networks, renderer, loss,
        backward, optimiser) can be captured into ONE HIP graph with capture().
        fused_adam (default: on a GPU): torch's single-kernel Adam -- the same update rule as the reference's
        torch.optim.Adam(lr, weight_decay) (train_to.py:72), one pass over the parameters instead of a dozen foreach passes
        plus, with capturable=True, six tiny kernels per parameter for the bias corrections (0.7 ms of a 3 ms step)."""
        self.cfg, self.device, self.device_dropout = cfg, device, device_dropout
        self.nets = StepNets(cfg).to(device)
        if fused_adam is None:
            fused_adam = torch.device(device).type == "cuda"
        self.optimizer = torch.optim.Adam(self.nets.parameters(), lr=lr, weight_decay=cfg.weight_decay,
                                          capturable=capturable, fused=bool(fused_adam))  # train_to.py:73-74
        self._graph = None
        self._schedule = None      # dpc.render.DeviceSchedule ONLY while a graph is being captured (record() below): an eager
                                   # loss() / __call__ on this object afterwards computes its values from the step it is asked for
        self._captured_schedule = None   # the schedule the current graph's kernels read; rewritten in front of every replay
        self.recaptures = 0
        self.global_step = 0
        self.grad_sync, self.sync_samples = None, (1, 1)

    def load_reference_state(self, state_dict):
        self.nets.load_state_dict(state_dict)

    def predict(self, images):
        cfg, n = self.cfg, self.nets
        enc = n.encoder(images)
        first_view = enc["ids"][::cfg.step_size]  # pool_single_view(cfg, ids, 0), model_base_to.py:8-10
        out = {"ids": enc["ids"], "points_1": n.decoder(first_view), "scaling_factor": n.scalePred(first_view)}
        out.update(n.poseNet(enc["poses"]))
        return out

    def loss(self, images, masks, global_step=None):
        """Forward of one step; returns (total loss, dict of the pieces the reference's outputs dict would hold)."""
        cfg = self.cfg
        step = self.global_step if global_step is None else global_step
        out = self.predict(images)
        K, V = cfg.pose_predict_num_candidates, cfg.step_size
        all_scales = out["scaling_factor"].repeat_interleave(V * K, dim=0) if cfg.pc_learn_occupancy_scaling else None
        all_points, point_index = out["points_1"], None  # [B,N,3] shared by the V*K clouds of an object, read in place
        sched = self._schedule       # inside a captured step: this step's sigma / keep-count live in device memory
        if cfg.pc_point_dropout != 1:                     # every replica drops its own points (:254-258 after :302-306)
            keep = R.get_dropout_prob(cfg, step)
            clouds = all_points.shape[0] * V * K
            if sched is not None:    # rows of `capacity` slots, the live count is read on the device at every replay
                point_index = R.point_dropout_indices(clouds, all_points.shape[1], (sched.capacity + 0.5) / all_points.shape[1],
                                                      all_points.device, n_live=sched.n_live)
            elif self.device_dropout:
                point_index = R.point_dropout_indices(clouds, all_points.shape[1], keep, all_points.device)
            else:  # the reference's host RNG protocol (one np.random.choice per cloud, in batch order), indices only
                n_out = int(all_points.shape[1] * keep)
                host = np.stack([np.random.choice(all_points.shape[1], n_out, replace=False) for _ in range(clouds)])
                point_index = torch.from_numpy(host.astype(np.int32)).to(all_points.device)
        kernel = R.smoothing_kernel(cfg, R.get_smooth_sigma(cfg, step))
        gt = pooled_masks(masks, cfg.vox_size)
        proj_loss, proj_out, winner = R.pointcloud_project_loss(cfg, all_points, out["poses"], None, None, kernel,
                                                                scaling_factor=all_scales, gt=gt, num_candidates=K,
                                                                point_index=point_index, schedule=sched)
        total = proj_loss.double()
        if K > 1 and cfg.pose_predictor_student:
            out["student_loss"] = student_loss(out["poses"], out["pose_student"], winner, K, cfg.pose_predictor_student_loss_weight)
            total = total + out["student_loss"]
        total = total * cfg.proj_weight
        out.update(projs=proj_out["proj"], min_loss=winner, proj_loss=proj_loss, pooled_masks=gt)
        return total, out

    def __call__(self, images, masks):
        """zero_grad, forward, loss, backward, Adam step (train_to.py:112-131).  Returns the loss tensor (no host sync).
        With `grad_sync` set (dpc.render.parallel.OverlappedGradAllReduce) the ranks' gradients are summed while the
        backward runs; `sync_samples` = (objects of this rank, objects of all ranks)."""
        if self.grad_sync is not None:
            self.grad_sync.prepare(*self.sync_samples)
        else:
            self.optimizer.zero_grad(set_to_none=True)
        total, _ = self.loss(images, masks)
        total.backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.optimizer.step()
        self.global_step += 1
        return total.detach()

    # ---- schedules under graph replay (model_pc_to.py:59-87, 171-179, 254-258: recomputed every step) ----
    def _schedule_values(self, step):
        """(x/y taps, z taps, live points per cloud | None) of `step`."""
        cfg = self.cfg
        kern = R.smoothing_kernel(cfg, R.get_smooth_sigma(cfg, step))
        n_live = None
        if cfg.pc_point_dropout != 1:
            n_live = int(cfg.pc_num_points * R.get_dropout_prob(cfg, step))
        return kern[0].reshape(-1).numpy(), kern[2].reshape(-1).numpy(), n_live

    def _new_schedule(self, step):
        """Device-resident schedule values sized for `step` and a while after it: the tap windows of this step's sigma,
        room for half as many live points again (the keep-probability grows along the schedule)."""
        kxy, kz, n_live = self._schedule_values(step)
        capacity = None if n_live is None else min(self.cfg.pc_num_points, -(-int(n_live * 1.5) // 64) * 64)
        return R.DeviceSchedule(self.device, kxy, kz, n_live=n_live, capacity=capacity)

    def _follow_schedule(self, recapture):
        """In front of a replay: write this step's values into device memory -- or, when they no longer fit what the graph
        was captured for (sigma crossed into another tap window, the live points outgrew the rows), capture again."""
        kxy, kz, n_live = self._schedule_values(self.global_step)
        if not self._captured_schedule.tight(kxy, kz, n_live):
            recapture()
            self.recaptures += 1
        else:
            self._captured_schedule.update(kxy, kz, n_live)

    def capture_compute(self, images, masks, warmup=2):
        """The multi-rank variant of capture(): forward, loss and backward as ONE HIP graph whose backward accumulates
        straight into the flat buckets of `grad_sync` (every .grad is a view into them); the gradient exchange and Adam run
        eagerly after each replay (OverlappedGradAllReduce.reduce_now -- collectives are not captured).  The eager step
        hides the exchange under a launch-bound 4 ms backward; this one has a 2 ms step and exposes the exchange.
        `warmup` eager steps run first: the first one fixes which parameters take part in the exchange.
        Returns replay(images, masks) -> loss tensor."""
        sync = self.grad_sync
        if sync is None:
            raise RuntimeError("capture_compute() is the step with a gradient exchange (set grad_sync); use capture() without")
        if self.cfg.pc_point_dropout != 1 and not self.device_dropout:
            raise RuntimeError("the reference's host-RNG point dropout uploads indices every step: not capturable; "
                               "use device_dropout=True")
        static_images, static_masks = images.clone(), masks.clone()
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self(static_images, static_masks)
        torch.cuda.current_stream(self.device).wait_stream(side)
        for p in sync.params:                 # what finish() left: views into the buckets
            p.grad = sync.views[id(p)]
        state = {}

        def record():
            self._captured_schedule = self._schedule = self._new_schedule(self.global_step)
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    for flat in sync.flat:
                        flat.zero_()
                    total, _ = self.loss(static_images, static_masks)
                    total.backward()              # the hooks are disarmed: gradients simply land in the buckets
            finally:
                self._schedule = None             # the captured kernels keep the pointers; eager calls do not see it
            self._graph, state["graph"], state["loss"] = graph, graph, total.detach()

        record()

        def replay(new_images, new_masks):
            static_images.copy_(new_images)
            static_masks.copy_(new_masks)
            self._follow_schedule(record)
            state["graph"].replay()
            sync.reduce_now(*self.sync_samples)
            self.optimizer.step()
            self.global_step += 1
            return state["loss"]

        return replay

    def capture(self, images, masks, warmup=3):
        """Capture forward + loss + backward + Adam into one HIP graph (the standard whole-step recipe of
        torch.cuda.graphs: warm up on a side stream, capture with gradients set to None, replay on static inputs).

        Everything inside the step is capture-safe: the renderer enqueues on the capturing stream and never synchronises,
        allocates through torch's (graph-private) pool, draws the point dropout on the device, and the optimiser was built
        with capturable=True.  The schedules stay step-exact: the Gaussian's tap values and the number of kept points live in
        device memory (dpc.render.DeviceSchedule) and are rewritten by one tiny launch in front of every replay from
        get_smooth_sigma / get_dropout_prob of the CURRENT global_step, like the reference recomputes them every step
        (model_pc_to.py:59-87, 171-179, 254-258); the graph is captured again, automatically, only when sigma crosses into
        another compiled tap window or the kept points outgrow the captured rows (`recaptures` counts).  Returns
        replay(images, masks) -> loss tensor (static memory, overwritten by the next replay)."""
        if self.grad_sync is not None:
            raise RuntimeError("capture() covers the single-process step; the overlapped gradient exchange runs eagerly")
        if self.cfg.pc_point_dropout != 1 and not self.device_dropout:
            raise RuntimeError("the reference's host-RNG point dropout uploads indices every step: not capturable; "
                               "use device_dropout=True")
        if not all(g["capturable"] for g in self.optimizer.param_groups):
            raise RuntimeError("build the TrainStep with capturable=True")
        static_images, static_masks = images.clone(), masks.clone()
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):          # lazy initialisations (Adam state, allocator, kernel attributes) happen here
                self(static_images, static_masks)
        torch.cuda.current_stream(self.device).wait_stream(side)
        state = {}

        def record():
            self._captured_schedule = self._schedule = self._new_schedule(self.global_step)
            self.optimizer.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    total, _ = self.loss(static_images, static_masks)
                    total.backward()
                    self.optimizer.step()
            finally:
                self._schedule = None             # the captured kernels keep the pointers; eager calls do not see it
            self._graph, state["graph"], state["loss"] = graph, graph, total.detach()

        record()

        def replay(new_images, new_masks):
            static_images.copy_(new_images)
            static_masks.copy_(new_masks)
            self._follow_schedule(record)
            state["graph"].replay()
            self.global_step += 1
            return state["loss"]

        return replay

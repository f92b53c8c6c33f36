"""Data-parallel plumbing for the projection path: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).

The renderer itself needs no collective: clouds are independent, so the batch is sharded by
SAMPLE -- all pose candidates (and all views) of a sample stay on one rank, which keeps the min-over-K selection
rank-local.  What a data-parallel trainer exchanges per step is (i) the scalar loss for logging and (ii) the
gradients of the shared network parameters that sit above the renderer (encoder / decoder / pose net: 133 MB in
chair_unsupervised).  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so gradients are flattened into a
few large buckets (one collective per bucket, all links busy) rather than one all-reduce per tensor.
"""
import time

import torch
import torch.distributed as dist


def shard_samples(num_samples, rank, world_size):
    """[begin, end) of the samples rank `rank` owns; contiguous, sizes differ by at most one."""
    base, extra = divmod(num_samples, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_clouds(num_samples, num_candidates, rank, world_size):
    """Cloud index range of a rank when clouds are laid out sample-major, candidate-minor (tf_repeat_0,
    dpc/models/model_pc_to.py:47-56): candidates of one sample never straddle ranks."""
    b, e = shard_samples(num_samples, rank, world_size)
    return b * num_candidates, e * num_candidates


def global_mean_loss(local_loss, local_samples, group=None):
    """The loss the single-process run would report: local losses are means over the rank's own samples
    (proj_loss / num_samples, model_pc_to.py:437-438), so weight by sample count before summing."""
    buf = torch.stack([local_loss.detach().to(torch.float64) * local_samples,
                       torch.tensor(float(local_samples), dtype=torch.float64, device=local_loss.device)])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return (buf[0] / buf[1]).to(local_loss.dtype)


class BucketedGradAllReduce:
    """Average the gradients of shared parameters across ranks with few, large all-reduces.

    Gradients are copied into flat buckets of about `bucket_mb` (async all-reduce per bucket as soon as it is
    full, so the collective of bucket i overlaps the packing of bucket i+1), divided by `total_samples /
    local_samples` weighting so that the result equals the gradient of the global mean loss, then scattered back.
    """

    def __init__(self, params, bucket_mb=64, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.bucket_elems = max(1, int(bucket_mb * (1 << 20)) // 4)

    def __call__(self, local_samples, total_samples):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        scale = float(local_samples) / float(total_samples)  # local mean -> contribution to the global mean
        pending, bucket, size = [], [], 0

        def flush():
            nonlocal bucket, size
            if not bucket:
                return
            flat = torch.cat([p.grad.reshape(-1) for p in bucket]).mul_(scale)
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            pending.append((work, flat, bucket))
            bucket, size = [], 0

        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)  # parameters without gradient on this rank still take part
            bucket.append(p)
            size += p.numel()
            if size >= self.bucket_elems:
                flush()
        flush()
        for work, flat, ps in pending:
            work.wait()
            off = 0
            for p in ps:
                p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
                off += p.numel()


class OverlappedGradAllReduce:
    """The same exchange started from inside the backward ("bucketed and overlapped with the network
    backward"; reference loop dpc/run/train_to.py:110-134 has a single process and no exchange at all).

    Layout: parameters are packed into flat buckets of about `bucket_mb` in REVERSE registration order -- the order in
    which autograd finishes their gradients, last layer first -- and every parameter's `.grad` is a VIEW into its
    bucket, so nothing is packed or unpacked: autograd accumulates straight into the bucket.  A
    post-accumulate-grad hook per parameter counts arrivals; the hook that completes a bucket scales it to the
    rank's share of the global mean loss and starts its all-reduce asynchronously (RCCL runs it on its own stream,
    ordered after the producing kernels), while autograd carries on with the layers below.  `finish()` starts the
    buckets that never filled (parameters without gradient this step count as zeros, so every rank agrees on the
    layout), waits for all of them and detaches the parameters that received no gradient (`.grad = None`, which is
    what the single-process run hands the optimiser); after the FIRST step those parameters are dropped from the exchange.

        sync = OverlappedGradAllReduce(nets.parameters())
        sync.prepare(local_samples, total_samples); loss.backward(); sync.finish(); optimizer.step()

    xGMI is point-to-point and per-link bound: a few large buckets (default 32 MB) keep every link busy; a bucket per
    tensor would pay one collective latency per layer.
    """

    def __init__(self, params, bucket_mb=32, group=None, overlap=True, single_rank_collectives=False):
        """single_rank_collectives: issue the all-reduces even in a process group of ONE rank (they are identities there).
        A world of one normally skips them; with this flag the very code path of N ranks -- librccl loaded, the collectives
        enqueued from the autograd hooks, the compute stream waiting on them -- runs on a single GPU (the RCCL smoke test)."""
        self.all_params = [p for p in params if p.requires_grad]
        self.group, self.overlap, self._single = group, overlap, bool(single_rank_collectives)
        self._limit = max(1, int(bucket_mb * (1 << 20)) // 4)
        self._events, self._host_exposed, self.steps = [], 0.0, 0
        self.collectives_issued = 0      # all-reduces handed to the backend so far (0 in a world of one without the flag)
        self._scale, self._armed, self._trimmed = 1.0, False, False
        self._layout(self.all_params)
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.all_params]

    def _layout(self, params):
        """Flat buckets over `params` in reverse registration order; .grad views; arrival bookkeeping."""
        self.params = list(params)
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel()
            if size >= self._limit:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.flat, self.views, self._bucket_of = [], {}, {}
        for bi, bucket in enumerate(self.buckets):
            ref = bucket[0]
            flat = torch.zeros(sum(p.numel() for p in bucket), dtype=ref.dtype, device=ref.device)
            off = 0
            for p in bucket:
                if p.dtype != ref.dtype or p.device != ref.device:
                    raise ValueError("parameters of one exchange must share dtype and device")
                self.views[id(p)] = flat[off:off + p.numel()].view_as(p)
                self._bucket_of[id(p)] = bi
                off += p.numel()
            self.flat.append(flat)
        self.num_buckets = len(self.buckets)
        self.nbytes = sum(f.numel() * f.element_size() for f in self.flat)
        self._expected = [len(b) for b in self.buckets]
        self._arrived = [0] * self.num_buckets
        self._seen = set()
        self._work = [None] * self.num_buckets

    def _active(self):
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self._single)

    def prepare(self, local_samples, total_samples):
        """Before the backward: zero the buckets, point every .grad at its slice, reset the arrival counters."""
        self._scale = float(local_samples) / float(total_samples)   # local mean -> contribution to the global mean
        for flat in self.flat:
            flat.zero_()
        for p in self.params:
            p.grad = self.views[id(p)]
        self._arrived = [0] * self.num_buckets
        self._seen.clear()
        self._work = [None] * self.num_buckets
        self._armed = True

    def _launch(self, bi):
        flat = self.flat[bi]
        if self._scale != 1.0:
            flat.mul_(self._scale)
        if self._active():
            self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.collectives_issued += 1
        else:
            self._work[bi] = True

    def _hook(self, p):
        if not self._armed:
            return
        if id(p) not in self._bucket_of:
            raise RuntimeError("a parameter that had no gradient in the first step has one now: the exchange was laid out "
                               "without it (every rank drops the same gradient-less parameters after step one)")
        bi, view = self._bucket_of[id(p)], self.views[id(p)]
        if p.grad.data_ptr() != view.data_ptr():   # autograd replaced the view (it does when .grad was None)
            view.copy_(p.grad)
            p.grad = view
        if self._work[bi] is not None:
            raise RuntimeError("a gradient arrived after its bucket had been sent (the set of parameters with a gradient "
                               "changed between steps)")
        self._seen.add(id(p))
        self._arrived[bi] += 1
        if self.overlap and self._arrived[bi] == self._expected[bi]:
            self._launch(bi)

    def finish(self):
        """After the backward: send what is left, wait, drop the gradients nothing contributed to."""
        for bi in range(self.num_buckets):
            if self._work[bi] is None:
                self._launch(bi)
        on_gpu = self.flat[0].is_cuda
        if on_gpu:   # how long the compute stream stalls on the collectives = the part of the exchange NOT hidden
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        else:
            t0 = time.perf_counter()
        for w in self._work:
            if w is not True:
                w.wait()
        if on_gpu:
            ev[1].record()
            self._events.append(ev)
            if len(self._events) >= 16:
                self._fold_events(wait=False)
        else:
            self._host_exposed += time.perf_counter() - t0
        self.steps += 1
        for p in self.params:
            if id(p) not in self._seen:
                p.grad = None
        # Parameters without a gradient (the reference's colour decoder and focal head are built but unused: 28 M of the 61 M
        # parameters) took part as zeros in this first exchange, so that all ranks agreed on the layout; the autograd graph
        # is the same on every rank and in every step, so from now on they are left out: less than half the bytes go over
        # the wire, and every bucket is sent the moment its last gradient lands.
        if not self._trimmed:
            self._trimmed = True
            alive = [p for p in self.params if id(p) in self._seen]
            if alive and len(alive) < len(self.params):
                grads = {id(p): p.grad.clone() for p in alive}
                self._layout(alive)
                for p in alive:
                    self.views[id(p)].copy_(grads[id(p)])
                    p.grad = self.views[id(p)]
        self._armed = False

    def _fold_events(self, wait):
        """Move finished event pairs into the running total, so that a long run holds a handful of events, not two per step.
        wait: also wait for the newest pair (a read of the total); otherwise only pairs that are done are folded."""
        if wait and self._events:
            self._events[-1][1].synchronize()
        keep = []
        for a, b in self._events:
            if b.query():
                self._host_exposed += 1e-3 * a.elapsed_time(b)
            else:
                keep.append((a, b))
        self._events = keep

    @property
    def exposed_seconds(self):
        """Total time the compute stream (GPU) or the host (CPU tensors) waited for collectives in finish()."""
        self._fold_events(wait=True)
        return self._host_exposed

    def reduce_now(self, local_samples, total_samples):
        """The exchange for gradients that are ALREADY complete in the buckets (a backward that ran without the hooks, e.g.
        replayed from a HIP graph, TrainStep.capture_compute): scale every bucket to this rank's share of the global mean
        loss, all-reduce them all, wait.  Nothing overlaps with the backward here; the buckets still go out back to back."""
        scale = float(local_samples) / float(total_samples)
        on_gpu = self.flat[0].is_cuda
        if on_gpu:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        else:
            t0 = time.perf_counter()
        works = []
        for flat in self.flat:
            if scale != 1.0:
                flat.mul_(scale)
            if self._active():
                works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if on_gpu:
            ev[1].record()
            self._events.append(ev)
            if len(self._events) >= 16:
                self._fold_events(wait=False)
        else:
            self._host_exposed += time.perf_counter() - t0
        self.steps += 1

    def exchange_only(self):
        """The collectives alone on the current bucket contents (timing aid)."""
        if not self._active():
            return
        works = [dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for f in self.flat]
        for w in works:
            w.wait()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []

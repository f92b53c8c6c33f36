"""Data-parallel plumbing for the projection path: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).

The renderer itself needs no collective: clouds are independent, so the batch is sharded by
SAMPLE -- all pose candidates (and all views) of a sample stay on one rank, which keeps the min-over-K selection
rank-local.  What a data-parallel trainer exchanges per step is (i) the scalar loss for logging and (ii) the
gradients of the shared network parameters that sit above the renderer (encoder / decoder / pose net: 133 MB in
chair_unsupervised).  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so gradients are flattened into a
few large buckets (one collective per bucket, all links busy) rather than one all-reduce per tensor.
"""
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

"""world_size-2 gloo test (CPU) of the N>1 path: sample sharding that keeps pose candidates together, the
global loss, and the bucketed gradient all-reduce a data-parallel trainer runs above the renderer.  The renderer
stand-in on CPU is the oracle (test infrastructure); on GPUs each rank calls dpc.render on its own shard."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dpc_oracle as O

S, K, N, G = 5, 2, 120, 16  # 5 samples (uneven over 2 ranks), 2 pose candidates each


def _problem():
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=5)
    pc, q, s, _, _, _ = O.synth_inputs(S * K, N, G, 31)
    gt = O.synth_inputs(S, 1, G, 32)[3]
    torch.manual_seed(5)
    net = torch.nn.Linear(4, 4, dtype=torch.float64)  # stands for the shared pose net above the renderer
    return cfg, pc, q, s, gt, net


def _local_loss(cfg, pc, q, s, gt, net, lo, hi):
    """loss of samples [lo, hi): q goes through the shared net, then projection + min-of-K loss."""
    c0, c1 = lo * K, hi * K
    qq = net(q[c0:c1].double())
    out = O.pointcloud_project_fast(cfg, pc[c0:c1], qq, None, None, O.smoothing_kernel(cfg, 0.8), scaling_factor=s[c0:c1])
    loss, win = O.proj_loss_pose_candidates(gt[lo:hi], out["proj"], K)
    return loss, win


def _worker(rank, world, port, ret):
    from dpc.render.parallel import BucketedGradAllReduce, global_mean_loss, shard_clouds, shard_samples

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, pc, q, s, gt, net = _problem()
        lo, hi = shard_samples(S, rank, world)
        assert shard_clouds(S, K, rank, world) == (lo * K, hi * K)
        loss, win = _local_loss(cfg, pc, q, s, gt, net, lo, hi)
        loss.backward()
        BucketedGradAllReduce(net.parameters(), bucket_mb=1e-5)(hi - lo, S)  # tiny buckets: several collectives
        gl = global_mean_loss(loss, hi - lo)
        ret[rank] = dict(loss=gl.item(), win=win.tolist(), range=(lo, hi),
                         grads=[p.grad.clone() for p in net.parameters()])
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    cfg, pc, q, s, gt, net = _problem()
    loss, win = _local_loss(cfg, pc, q, s, gt, net, 0, S)
    loss.backward()
    assert ret[0]["range"] == (0, 3) and ret[1]["range"] == (3, 5)
    assert ret[0]["win"] + ret[1]["win"] == win.tolist()
    for r in (0, 1):
        assert abs(ret[r]["loss"] - loss.item()) < 1e-12 * max(1.0, abs(loss.item()))
        for g, p in zip(ret[r]["grads"], net.parameters()):
            assert torch.allclose(g, p.grad, rtol=1e-10, atol=1e-12)


def test_shard_ranges_cover_everything():
    from dpc.render.parallel import shard_samples

    for n in (0, 1, 7, 32, 33):
        for w in (1, 2, 3, 8):
            r = [shard_samples(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


# ------------------------------------------------------------------------------------------------------
# The same exchange with the real parameter set: the harness networks (tests/test_harness.py), whose colour and
# focal heads get no gradient (such tensors must be zero-filled, not skipped, or ranks disagree
# on the bucket layout).  The renderer is replaced by the oracle on a tiny grid.
# ------------------------------------------------------------------------------------------------------
def _harness_problem(layout="c3"):
    import json

    from dpc.harness import StepNets, pooled_masks

    class C(dict):
        __getattr__ = dict.__getitem__

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cfg = C(json.load(open(os.path.join(golden, "f10_config.json"))))
    cfg.update(batch_size=4, vox_size=8, pc_gauss_kernel_size=3)
    if layout == "c4":   # BASELINE configs[3] as a full step: one view and one pose per object, the decoder sized by the points
        cfg.update(step_size=1, pose_predict_num_candidates=1, pose_predictor_student=False, pc_num_points=96, batch_size=6)
    torch.manual_seed(11)
    nets = StepNets(cfg)
    g = torch.Generator().manual_seed(12)
    n = cfg.batch_size * cfg.step_size
    images, masks = torch.rand(n, 3, 32, 32, generator=g), (torch.rand(n, 1, 32, 32, generator=g) > 0.5).float()
    return cfg, nets, images, pooled_masks(masks, cfg.vox_size)


def _harness_loss(cfg, nets, images, gt, lo, hi):
    """objects [lo, hi): networks -> oracle renderer -> min-of-K loss (mean over the rank's own samples)."""
    V, K = cfg.step_size, cfg.pose_predict_num_candidates
    enc = nets.encoder(images[lo * V:hi * V])
    first = enc["ids"][::V]
    pts = nets.decoder(first).repeat_interleave(V * K, dim=0)
    scale = nets.scalePred(first).repeat_interleave(V * K, dim=0)
    poses = nets.poseNet(enc["poses"])["poses"]
    ocfg = O.Cfg(vox_size=cfg.vox_size, pc_gauss_kernel_size=cfg.pc_gauss_kernel_size)
    out = O.pointcloud_project_fast(ocfg, pts, poses, None, None, O.smoothing_kernel(ocfg, 0.8), scaling_factor=scale)
    return O.proj_loss_pose_candidates(gt[lo * V:hi * V].double(), out["proj"], K)[0]


def _harness_worker(rank, world, port, ret):
    from dpc.render.parallel import BucketedGradAllReduce, shard_samples

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, nets, images, gt = _harness_problem()
        lo, hi = shard_samples(cfg.batch_size, rank, world)
        _harness_loss(cfg, nets, images, gt, lo, hi).backward()
        missing = sorted(k for k, p in nets.named_parameters() if p.grad is None)
        BucketedGradAllReduce(nets.parameters(), bucket_mb=0.05)((hi - lo) * cfg.step_size, cfg.batch_size * cfg.step_size)
        ret[rank] = dict(missing=missing, grads={k: p.grad.clone() for k, p in nets.named_parameters()})
    finally:
        dist.destroy_process_group()


def test_two_rank_gradients_of_the_real_networks():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_harness_worker, args=(2, port, ret), nprocs=2, join=True)
    cfg, nets, images, gt = _harness_problem()
    _harness_loss(cfg, nets, images, gt, 0, cfg.batch_size).backward()
    assert any(k.startswith("decoder.rgb") for k in ret[0]["missing"]) and "focalPred.fc.weight" in ret[0]["missing"]
    for r in (0, 1):
        for k, p in nets.named_parameters():
            g = ret[r]["grads"][k]
            if p.grad is None:
                assert float(g.abs().max()) == 0.0, k  # zero-filled, took part in the exchange
            else:
                assert torch.allclose(g, p.grad, rtol=1e-4, atol=1e-6 * float(p.grad.abs().max())), k


# ------------------------------------------------------------------------------------------------------
# The exchange started from autograd hooks while the backward is still running (what bench.py --config c3 runs over
# RCCL): two steps, so that the second one goes through the adapted per-bucket arrival counts.
# ------------------------------------------------------------------------------------------------------
def _hooked_worker(rank, world, port, ret, layout="c3"):
    from dpc.render.parallel import OverlappedGradAllReduce, shard_samples

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, nets, images, gt = _harness_problem(layout)
        lo, hi = shard_samples(cfg.batch_size, rank, world)
        sync = OverlappedGradAllReduce(nets.parameters(), bucket_mb=0.05)
        launched_early = []
        for _ in range(2):
            sync.prepare((hi - lo) * cfg.step_size, cfg.batch_size * cfg.step_size)
            _harness_loss(cfg, nets, images, gt, lo, hi).backward()
            launched_early.append(sum(w is not None for w in sync._work))   # buckets already on the wire before finish()
            sync.finish()
        ret[rank] = dict(grads={k: (None if p.grad is None else p.grad.clone()) for k, p in nets.named_parameters()},
                         early=launched_early, buckets=sync.num_buckets)
    finally:
        dist.destroy_process_group()


def test_two_rank_overlapped_exchange_matches_single_process():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_hooked_worker, args=(2, port, ret), nprocs=2, join=True)
    cfg, nets, images, gt = _harness_problem()
    _harness_loss(cfg, nets, images, gt, 0, cfg.batch_size).backward()
    for r in (0, 1):
        assert ret[r]["buckets"] > 3
        # first step: buckets holding a gradient-less parameter wait for finish(); second step: every bucket that gets
        # any gradient is sent from inside the backward
        assert ret[r]["early"][1] >= ret[r]["early"][0] > 0
        for k, p in nets.named_parameters():
            g = ret[r]["grads"][k]
            if p.grad is None:
                assert g is None, k   # the optimiser sees what the single-process run shows it
            else:
                assert torch.allclose(g, p.grad, rtol=1e-4, atol=1e-6 * float(p.grad.abs().max())), k


def test_two_rank_exchange_of_the_config4_full_step_layout():
    """BASELINE configs[3] as worded -- data-parallel objects, one cloud each, WITH the gradient all-reduce -- at a tiny
    shape: one view and one pose candidate per object (`bench.py --config c4 --full-step` per rank: 8 objects of 16000
    points into 128^3), 6 objects over 2 ranks, the hooked exchange against single-process gradients."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_hooked_worker, args=(2, port, ret, "c4"), nprocs=2, join=True)
    cfg, nets, images, gt = _harness_problem("c4")
    assert cfg.pose_predict_num_candidates == 1 and nets.decoder.pts_raw_fc.out_features == 3 * 96
    _harness_loss(cfg, nets, images, gt, 0, cfg.batch_size).backward()
    for r in (0, 1):
        for k, p in nets.named_parameters():
            g = ret[r]["grads"][k]
            if p.grad is None:
                assert g is None, k
            else:
                assert torch.allclose(g, p.grad, rtol=1e-4, atol=1e-6 * float(p.grad.abs().max())), k

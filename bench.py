#!/usr/bin/env python3
"""Benchmark of the differentiable point-cloud projection hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic clouds already resident in HBM:
pointcloud_project_fast (transform -> splat -> clamp -> Gaussian -> scale/clamp -> DRC silhouette), the loss
sum((proj-gt)^2)/B, and the hand-written backward to d(pc), d(q), d(s).  Workload = BASELINE.json configs[1]
(SURVEY.md 8(d) "c2"): B=32 clouds, N=8000 points, 64^3 grid, 21-tap Gaussian, sigma = 0.01 world units
(sigma_rel 0.64).  By default a step is ONE native call (dpc_project_loss_step through dpc.render.project_loss_step) that
enqueues the four kernels on static buffers -- no HIP graph; `--launch graph` captures the autograd path once and replays it
(rounds 1-2).  Weak scaling: every rank runs its own B=32 shard, no data-path collective; the per-step losses are all-reduced
once after the timed loop.

`--gpus N` without a launcher starts the N ranks itself (a child `python -m torch.distributed.run ... bench.py` started
before anything in this process touches a GPU) and relays rank 0's line; under torchrun (WORLD_SIZE set) it is a rank.
`--config c3` runs BASELINE configs[2], the full chair_unsupervised training step (networks + renderer + loss + Adam), one
shard of 8 objects per rank with the parameter gradients all-reduced over RCCL in buckets, overlapped with the backward.

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel (per-kernel HIP-event timing from
the library's opt-in profiler, eager pass after the timed region, minus one event marker's cost);
`roofline_step` prices the whole step with
the contractual algorithmic bytes A(N,G) of SURVEY.md 8(d); `cpu_baseline` (the only part of this file that touches
oracle/) times the CPU oracle (a port of the reference's PyTorch CPU path) on a bounded sample on this host.  Beside `value`,
never as it: `step_us` (median of short windows), `hip_graph_replay`, `two_batches_in_flight`, `forward_only_us`, and the
reference's own call sequence through the drop-in signatures, eager (`plain_eager`) and replayed (`plain_graph_replay`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "pytorch-unsup-pc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# The default step is ONE native call that enqueues the four kernels (dpc_project_loss_step): no HIP graph, no runtime
# setting involved.  Only runs whose VALUE comes from a replayed graph -- `--launch graph` (the round-1/2 way: the autograd
# path captured and replayed) and the captured full training steps -- ask for graph replay
# without the runtime's pre-built AQL packets ("graph packet capture", the default of ROCm 7, costs about 1 us per kernel
# boundary inside a replayed graph on MI355X): read when HIP initialises, so set before torch is
# imported; an explicit DEBUG_CLR_GRAPH_PACKET_CAPTURE in the environment wins.
_argv = sys.argv[1:]
if ("graph" in [a for i, a in enumerate(_argv) if i and _argv[i - 1] == "--launch"] or "--launch=graph" in _argv
        or "c3" in [a for i, a in enumerate(_argv) if i and _argv[i - 1] == "--config"] or "--full-step" in _argv):
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # = dpc.render.prefer_direct_graph_launch()

import torch
import torch.distributed as dist

B, N_PTS, G, KSIZE, SIGMA_REL, K_CAND = 32, 8000, 64, 21, 0.64, 1   # BASELINE configs[1] ("c2"), the default workload
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
# other BASELINE configs, for extra data points only (the contract metric is quoted on c2)
CONFIGS = {"c2": (32, 8000, 64, 0.64, 1),     # B, N, G, sigma_rel, pose candidates
           "c4": (8, 16000, 128, 1.28, 1),    # per-GPU shard of B=64 over 8 GPUs, sigma = 0.01
           "c5": (128, 8000, 64, 0.64, 8)}    # 16 samples x 8 candidate rotations, min-of-K loss


def algorithmic_bytes_per_cloud(n, g):
    """A(N,G) = 36N + min(32N, 4G^3) + 36G^3 + 8G^2 (fwd+bwd, fp32)."""
    return 36 * n + min(32 * n, 4 * g ** 3) + 36 * g ** 3 + 8 * g ** 2


def kernel_bytes_per_cloud(n, g):
    """Bytes each launch of THIS implementation must move per cloud (kernel table): the tensors a launch has to
    read or write once, whatever the cache does.  Chunk of 256 points = 256 x (16 B record + 16 B {point, index}) + bin
    offsets; optional outputs (`smoothed`, `trans`) are not written on the hot path and not counted."""
    g3, g2 = g ** 3, g ** 2
    cells = 32 * n + ((n + 255) // 256) * (((g + 2) * 2 + 15) // 16) * 16   # binned records + per-chunk bin offsets
    return {
        "k_locate": 12 * n + cells,                       # read pc, write the binned records
        "k_splat_hw": 16 * n + 4 * g3 + g3 // 8,          # read records, write T (after W/H passes) + clamp mask
        "k_splat_xl": 16 * n + 4 * g3 + g3 // 8,          # the same work, x-in-lanes kernel (64-wide grids, radius <= 6)
        "k_zcol_fwd": 4 * g3 + 4 * g2,                    # read T, write silhouette
        "k_zcol_bwd": 4 * g3 + 4 * g3 + 8 * g2,           # read T + dproj (or proj, gt), write dT
        "k_zcol_fwdbwd": 4 * g3 + 4 * g3 + 8 * g2,        # read T + gt, write dT + silhouette (column backward fused)
        "k_gather_hw": 4 * g3 + g3 // 8 + 32 * n + 12 * n,  # read dT + mask + records with their points, write dpc
    }


def step_bytes_per_cloud(n, g, k_cand):
    """Contractual bytes of one cloud of the step that is actually run: A(N,G) for one pose candidate per sample; with K
    candidates only the winning cloud of a sample runs a backward, so the backward share counts 1/K."""
    fwd = 12 * n + 16 * g ** 3 + 4 * g ** 2
    bwd = algorithmic_bytes_per_cloud(n, g) - fwd
    return fwd + bwd / k_cand


PROFILE_SUMMARIES = ("r04_rocprof_summary.json", "r04_rocprof_summary_c4.json", "r03_rocprof_summary.json", "r03_rocprof_summary_c4.json", "r02_rocprof_summary.json",
                     "r02_rocprof_summary_c4.json", "r01_rocprof_summary.json")   # newest first


def profile_summary(config="c2"):
    """The newest committed rocprofv3 summary of THIS config (tools/profile_gpu.sh: kernel trace + FETCH_SIZE / WRITE_SIZE
    in separate passes, corrected with the calibration kernels as MI355X_MICROARCH.md prescribes), or None."""
    for name in PROFILE_SUMMARIES:
        try:
            summary = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if summary.get("config", "c2") == config:
            summary["file"] = "profiles/" + name
            return summary
    return None


def measured_traffic(kernel, config="c2"):
    """HBM bytes per launch of `kernel` from that summary; None when no profile of this kernel at this config is on file."""
    summary = profile_summary(config)
    for kname, rec in (summary or {}).get("kernels", {}).items():
        if kname.split("<")[0] == kernel and rec.get("hbm_bytes_per_launch"):
            return rec["hbm_bytes_per_launch"]
    return None


def dominant_kernel(kern_ms, config="c2"):
    """The kernel the `roofline` object describes.  Two kernels of the step are within a microsecond of each other and swap
    places from run to run, so the choice is made ONCE per profile: the kernel with the largest average duration in the
    committed rocprofv3 summary of this config (when it is among the kernels this run launched), else the longest of this
    run's own event timings."""
    summary = profile_summary(config)
    if summary:
        ranked = sorted(summary.get("kernels", {}).items(), key=lambda kv: kv[1].get("avg_us", 0.0), reverse=True)
        for kname, _ in ranked:
            if kname.split("<")[0] in kern_ms:
                return kname.split("<")[0], summary["file"]
    return (max(kern_ms, key=kern_ms.get), "this run's event timings") if kern_ms else (None, None)


def synthetic_inputs(b, n, g, seed):
    """Synthetic inputs of SURVEY.md 8(d): points like the decoder's tanh/2 output, unnormalised quaternions, occupancy
    scales in (0.5, 1), and a random 2G x 2G mask average-pooled to the silhouette size (the reference's add_proj_loss)."""
    gen = torch.Generator().manual_seed(seed)
    pc = (torch.tanh(0.5 * torch.randn(b, n, 3, generator=gen)) / 2).float()
    q = torch.randn(b, 4, generator=gen).float()
    s = (0.5 + 0.5 * torch.rand(b, 1, generator=gen)).float()
    mask = (torch.rand(b, 1, 2 * g, 2 * g, generator=gen) > 0.5).float()
    gt = torch.nn.functional.avg_pool2d(mask, 2).permute(0, 2, 3, 1).contiguous()
    return pc, q, s, gt


def make_inputs(device, seed):
    return [x.to(device=device, dtype=torch.float32) for x in synthetic_inputs(B, N_PTS, G, seed)]


def cpu_baseline():
    """Oracle (op-for-op torch-CPU fp64 port of the reference path) fwd+bwd on the same workload, bounded to ~15 s: a
    thread-count sweep on 4 of the 32 clouds, then the full 32 clouds, best of 3, at the fastest count, with the Gaussian
    (the reference's CUDA-branch semantics = what the GPU path computes).  Beside it, as SURVEY 8(d) asks: the literal CPU call (the reference skips the Gaussian on CPU)
    and a single-thread run."""
    from oracle import dpc_oracle as O

    nb = 4
    cfg = O.Cfg(vox_size=G, pc_gauss_kernel_size=KSIZE)
    pc, q, s, gt, _, _ = O.synth_inputs(B, N_PTS, G, 1234)
    kern = O.smoothing_kernel(cfg, SIGMA_REL)

    def rate(n, reps, smooth):
        best = float("inf")
        for _ in range(reps):
            a, b_, c = (x[:n].clone().requires_grad_(True) for x in (pc, q, s))
            t0 = time.perf_counter()
            out = O.pointcloud_project_fast(cfg, a, b_, None, None, kern, scaling_factor=c, smooth=smooth)
            (((out["proj"] - gt[:n]) ** 2).sum() / n).backward()
            best = min(best, time.perf_counter() - t0)
        return n / best

    # torch-CPU oversubscribes on this chain of small ops: try a few thread counts, report the fastest as `value`
    threads = torch.get_num_threads()
    by_threads = {}
    try:
        for nt in sorted({1, 8, 32, threads}):
            if nt > threads:
                continue
            torch.set_num_threads(nt)
            by_threads[nt] = rate(nb, 2 if nt > 1 else 1, True)
        # the FULL workload (all 32 clouds in one call), best of 3, at the two fastest thread counts of the sweep (a batch
        # of 32 clouds does not always prefer the count the 4-cloud sample preferred)
        value, best_nt = 0.0, 1
        ranked = sorted(by_threads, key=by_threads.get, reverse=True)
        for nt in [n for n in ranked[:2] if by_threads[n] >= 0.75 * by_threads[ranked[0]]]:
            torch.set_num_threads(nt)
            v = rate(B, 3, True)
            if v > value:
                value, best_nt = v, nt
        torch.set_num_threads(best_nt)
        literal = rate(nb, 2, False)
    finally:
        torch.set_num_threads(threads)
    single = by_threads.get(1)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": value, "unit": "point-clouds/sec", "cores": best_nt, "kind": "port",
            "sample": "the full workload (%d clouds in one call), fwd+bwd with Gaussian (reference CUDA-branch semantics), "
                      "fp64, best of 3 at the faster of the two best thread counts of a sweep on %d clouds (`cores` = that thread count)"
                      % (B, nb),
            "value_by_threads": {str(k): v for k, v in by_threads.items()}, "value_no_smoothing": literal,
            "value_1thread": single, "cpu_model": model, "host_cpus": os.cpu_count()}


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as `python -m torch.distributed.run`
    children (one process per GPU, RCCL over xGMI) and relay their output.  Nothing in THIS process has initialised a GPU
    (device_count() does not), so no process that has touched a GPU ever re-execs.  Returns the exit code."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < args.gpus and not args.rehearse_on_one_gpu:
        sys.stderr.write("bench.py: --gpus %d but this node shows %d device(s); nothing was run\n" % (args.gpus, have))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def full_step_main(args, rank, world, local):
    """BASELINE configs[2]: the chair_unsupervised training step per rank (8 objects x 4 views = 32 images 128x128, K = 4
    pose candidates -> 128 clouds of 8000 points into 64^3; networks + renderer + loss + backward + Adam, fp32, eager),
    weak scaling: every rank owns 8 objects, the parameter gradients (133 MB) are summed over RCCL in buckets launched from
    autograd hooks while the backward is still running (dpc.render.parallel.OverlappedGradAllReduce)."""
    dist_on = world > 1 or args.rccl_at_one_rank   # one rank: the collectives of the N-rank line over RCCL, for a one-GPU box
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.rehearse_on_one_gpu else "nccl",
                                **({} if args.rehearse_on_one_gpu else {"device_id": device}))
    from dpc.harness import TrainStep, chair_unsupervised
    from dpc.render.parallel import OverlappedGradAllReduce

    if args.config == "c4":
        # BASELINE configs[3] as worded: "16000 pts, 128^3 grid, batch=64 data-parallel over 8 GPUs with RCCL grad all-reduce"
        # = 8 objects per rank, one view and one pose each (8 clouds per rank), the decoder sized by the
        # points (1024 -> 48000: 57.9 M parameters with a gradient, 232 MB exchanged), sigma = 0.01 world units
        cfg = chair_unsupervised(pc_point_dropout=args.keep, pc_num_points=16000, vox_size=128, batch_size=8, step_size=1,
                                 pose_predict_num_candidates=1, pose_predictor_student=False, pc_relative_sigma=1.28,
                                 pc_relative_sigma_end=1.28)
    else:
        cfg = chair_unsupervised(pc_point_dropout=args.keep)
    torch.manual_seed(0)                      # same initial weights on every rank
    if args.captured_compute:
        args.captured = True
    step = TrainStep(cfg, device, device_dropout=True, capturable=args.captured and world == 1 and not args.captured_compute)
    torch.cuda.manual_seed(4321 + rank)      # ... but every rank draws its OWN dropout subsets (device generator)
    sync = None
    if dist_on or args.captured_compute:
        sync = OverlappedGradAllReduce(step.nets.parameters(), bucket_mb=32, overlap=not args.no_overlap)
        step.grad_sync, step.sync_samples = sync, (cfg.batch_size, cfg.batch_size * world)
    nimg = cfg.batch_size * cfg.step_size
    gen = torch.Generator().manual_seed(1234 + rank)
    images = torch.rand(nimg, 3, 128, 128, generator=gen).to(device)
    masks = (torch.rand(nimg, 1, 128, 128, generator=gen) > 0.5).float().to(device)
    trainer = step
    if args.captured and world == 1 and not args.captured_compute:
        step = trainer.capture(images, masks)   # replay(images, masks): copies the batch into the graph's inputs, one launch
    elif args.captured:
        step = trainer.capture_compute(images, masks)   # forward + backward as one graph; exchange and Adam eager
    for _ in range(args.warmup):
        step(images, masks)
    torch.cuda.synchronize(device)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(images, masks)
    torch.cuda.synchronize(device)
    mine = time.perf_counter() - t0
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    times = torch.tensor([wall, mine], device=device, dtype=torch.float64)
    per_rank = [times.clone() for _ in range(world)]
    if dist_on:
        dist.all_gather(per_rank, times)
    wall = max(float(t[0]) for t in per_rank)
    comm = None
    if sync is not None:   # the exchange alone, nothing else on the GPU: what overlapping has to hide
        torch.cuda.synchronize(device)
        c0 = time.perf_counter()
        for _ in range(5):
            sync.exchange_only()
        torch.cuda.synchronize(device)
        comm = (time.perf_counter() - c0) / 5
    if rank == 0:
        clouds = nimg * cfg.pose_predict_num_candidates
        print(json.dumps({
            "metric": "point-clouds/sec rendered inside the full chair_unsupervised train step (fwd + loss + bwd + grad all-reduce + Adam)",
            "value": world * clouds * args.steps / wall, "unit": "point-clouds/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[3] (c4) as a full training step: per rank 8 objects x 1 view (8 images "
                                    "128x128x3), 1 pose -> 8 clouds x %d of 16000 pts -> 128^3, 21 taps sigma_rel 1.28, %s"
                                    if args.config == "c4" else
                                    "BASELINE configs[2] (c3): per rank 8 objects x 4 views (32 images 128x128x3), K=4 pose "
                                    "candidates -> 128 clouds x %d of 8000 pts -> 64^3, 21 taps sigma_rel 3.0, %s")
                                   % (int(cfg.pc_num_points * args.keep), ("whole step as one HIP graph" if world == 1 else "forward + backward as one HIP graph, exchange + Adam eager") if args.captured else "eager launches"),
                       "parameters": sum(p.numel() for p in trainer.nets.parameters()),
                       "gradient_exchange": None if sync is None else
                       {"backend": dist.get_backend() if dist.is_initialized() else "none (one rank)", "buckets": sync.num_buckets, "bytes": sync.nbytes,
                        "overlapped_with_backward": not args.no_overlap and not args.captured}},
            "train_steps_per_sec": world * args.steps / wall,
            "ms_per_step_by_rank": [1e3 * float(t[1]) / args.steps for t in per_rank],
            "allreduce_alone_ms": None if comm is None else 1e3 * comm,
            "allreduce_exposed_ms": None if sync is None else 1e3 * sync.exposed_seconds / max(1, sync.steps),
            "loss": float(loss)}))
    if dist_on:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP graph replay")
    ap.add_argument("--launch", choices=["plan", "graph"], default="plan",
                    help="plan (default, one pose candidate per sample): one native call per step enqueues the four kernels "
                         "(dpc.render.project_loss_step); graph: the autograd path captured into a HIP graph and replayed")
    ap.add_argument("--streams", type=int, default=1, help="split the batch over this many HIP streams inside the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sigma-rel", type=float, default=None,
                    help="kernel experiments: another smoothing width than the config's (the line then says so in config.workload)")
    ap.add_argument("--split", type=int, default=1,
                    help="the batch of a step as this many independent sub-batches, each its own HIP graph on its own stream, "
                         "joined at the end of every step (steps stay sequential)")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="experiment: this many INDEPENDENT batches in flight (each its own buffers, HIP graph and stream, "
                         "replayed round-robin); the default 1 is the contract's sequence of dependent-looking steps")
    ap.add_argument("--no-extras", action="store_true", help="skip the median/best windows and the forward-only timing "
                    "(profiling runs: keeps every traced kernel inside the contract's fwd+bwd step)")
    ap.add_argument("--config", choices=sorted(CONFIGS) + ["c3"], default="c2",
                    help="BASELINE config (default c2 = the metric's); c3 = the full training step with the RCCL gradient exchange")
    ap.add_argument("--keep", type=float, default=1.0, help="c3: point keep-probability of the dropout (1.0 = all 8000 points)")
    ap.add_argument("--full-step", action="store_true",
                    help="c4: BASELINE configs[3] as a full training step (networks sized for 16000 points, renderer at 128^3, "
                         "loss, backward, RCCL gradient all-reduce, Adam) instead of the renderer shard alone")
    ap.add_argument("--no-overlap", action="store_true", help="c3: all-reduce after the backward instead of inside it")
    ap.add_argument("--captured-compute", action="store_true",
                    help="c3: forward + backward as one HIP graph accumulating into the gradient buckets, exchange + Adam eager "
                         "(what --captured does on several ranks), also on one rank")
    ap.add_argument("--captured", action="store_true",
                    help="c3, one rank: the whole step (networks, renderer, loss, backward, Adam) replayed as ONE HIP graph")
    ap.add_argument("--api", choices=["fused", "plain"], default="fused",
                    help="fused: pointcloud_project_loss (renderer + loss in one autograd node, the default and the contract "
                         "line); plain: the reference's own call sequence, pointcloud_project_fast then the loss in torch")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a single-GPU box: every rank uses cuda:0 and the collectives run over gloo")
    ap.add_argument("--rccl-at-one-rank", action="store_true",
                    help="with one rank under torch.distributed.run: initialise RCCL anyway and run the barriers and all-reduces "
                         "of the N-rank line (what a one-GPU box can execute of the --gpus N code)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))   # this process never touches a GPU: the N ranks are its children
    global B, N_PTS, G, SIGMA_REL, K_CAND
    if args.config != "c3" and not args.full_step:
        B, N_PTS, G, SIGMA_REL, K_CAND = CONFIGS[args.config]
        if args.sigma_rel is not None:
            SIGMA_REL = args.sigma_rel
    if args.full_step and args.config != "c4":
        raise SystemExit("--full-step is the full-training-step form of --config c4 (c3 is a full step already)")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start the ranks with `python bench.py --gpus N` or with "
                         "`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path in dpc.render)")
    if not args.rehearse_on_one_gpu and torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d but this node shows %d device(s)" % (world, torch.cuda.device_count()))
    if args.config == "c3" or args.full_step:
        return full_step_main(args, rank, world, local)
    dist_on = world > 1 or args.rccl_at_one_rank   # as in full_step_main
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI

    import dpc.render as R
    from dpc.harness import chair_unsupervised
    from dpc.render import _native

    cfg = chair_unsupervised(vox_size=G, pc_gauss_kernel_size=KSIZE)  # the experiment's renderer settings; no oracle here
    kern = R.smoothing_kernel(cfg, SIGMA_REL)
    pc, q, s, gt = make_inputs(device, 1234 + rank)
    if K_CAND > 1:  # candidates of a sample share its cloud, scale and mask (tf_repeat_0); quaternions differ
        S = B // K_CAND
        pc = pc[:S].contiguous()  # shared point sets: [S,N,3] read in place by the K clouds of a sample (no tf_repeat_0 copy)
        s = s[:S].repeat_interleave(K_CAND, dim=0).contiguous()
        gt = gt[:S].contiguous()
    pc.requires_grad_(True), q.requires_grad_(True), s.requires_grad_(True)

    one = torch.ones((), device=device)
    ns = max(1, args.streams)
    if B % ns:
        raise SystemExit("--streams must divide %d" % B)
    if ns > 1 and K_CAND > 1:
        raise SystemExit("--streams is an experiment of the one-candidate configs")
    # per-stream shards (leaves of their own, so each shard's backward accumulates into its own .grad)
    shards = [[x[i * (x.shape[0] // ns):(i + 1) * (x.shape[0] // ns)].detach().clone().requires_grad_(x.requires_grad)
               for x in (pc, q, s, gt)] for i in range(ns)]
    extra = [torch.cuda.Stream(device) for _ in range(ns - 1)]
    share = torch.full((), 1.0 / ns, device=device)

    def step():
        # projection + sum((proj-gt)^2)/B in one autograd node (loss folded into the ray-march kernels)
        if ns == 1 and args.api == "plain":  # what ModelPointCloud does: compute_projection, then add_proj_loss in torch
            pc.grad = q.grad = s.grad = None
            proj = R.pointcloud_project_fast(cfg, pc, q, None, None, kern, scaling_factor=s)["proj"]
            if K_CAND > 1:
                loss, _ = R.silhouette_loss(proj, gt, K_CAND)
            else:
                loss = ((proj - gt) ** 2).sum() / B
            loss.backward(gradient=one)
            return loss
        if ns == 1:
            pc.grad = q.grad = s.grad = None
            loss, _, _ = R.pointcloud_project_loss(cfg, pc, q, None, None, kern, scaling_factor=s, gt=gt, num_candidates=K_CAND)
            loss.backward(gradient=one)  # preallocated d(loss)=1: no ones-fill launch per step
            return loss
        main = torch.cuda.current_stream(device)
        losses = []
        for i, (spc, sq, ss, sgt) in enumerate(shards):
            st = main if i == 0 else extra[i - 1]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                spc.grad = sq.grad = ss.grad = None
                l, _, _ = R.pointcloud_project_loss(cfg, spc, sq, None, None, kern, scaling_factor=ss, gt=sgt, num_candidates=K_CAND)
                l.backward(gradient=share)  # each shard's loss is a mean over B/ns clouds
                losses.append(l)
        for st in extra:
            main.wait_stream(st)
        return losses[0]

    side = torch.cuda.Stream(device)
    graph = None
    with torch.cuda.stream(side):
        for _ in range(3):  # settle allocator / lazy init before capture
            step()
        side.synchronize()
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may make HIP calls of its own while this thread captures
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                static_loss = step()

        last = [static_loss if graph is not None else None]

        # the default step: forward + backward as ONE native call on static buffers (no graph, no Python between launches)
        plan = None
        if (args.launch == "plan" and ns == 1 and args.api == "fused" and not args.no_graph
                and args.in_flight == 1 and args.split == 1):
            plan = R.project_loss_step(cfg, kern, B, N_PTS, device, num_candidates=K_CAND, point_replicas=B // pc.shape[0])
            plan.bind(pc.detach(), q.detach(), s.detach(), gt)
            for _ in range(3):
                plan.run()
            side.synchronize()
            last[0] = plan.loss

        # --in-flight n > 1: n - 1 more lanes, each with its own inputs, gradients, workspace pool, graph and stream
        def new_lane():
            """One more independent batch: its own inputs, gradients, buffers, HIP graph and stream."""
            lpc, lq, ls = (x.detach().clone().requires_grad_(True) for x in (pc, q, s))
            lgt, lst = gt.clone(), torch.cuda.Stream(device)

            def lane_step():
                lpc.grad = lq.grad = ls.grad = None
                loss, _, _ = R.pointcloud_project_loss(cfg, lpc, lq, None, None, kern, scaling_factor=ls, gt=lgt, num_candidates=K_CAND)
                loss.backward(gradient=one)
                return loss
            with torch.cuda.stream(lst):
                for _ in range(3):
                    lane_step()
                lst.synchronize()
                lg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(lg, stream=lst, capture_error_mode="thread_local"):
                    lane_step()
            return lg, lst

        lanes = []
        if args.in_flight > 1:
            if graph is None or ns != 1 or args.api != "fused":
                raise SystemExit("--in-flight is an experiment of the graph-replayed fused step")
            lanes = [new_lane() for _ in range(1, args.in_flight)]

        # --split n: the step's batch as n sub-batches of B/n clouds: n graphs on n streams, fork and join on `side`
        parts = []
        if args.split > 1:
            if graph is None or ns != 1 or args.api != "fused" or K_CAND != 1 or B % args.split or args.in_flight > 1:
                raise SystemExit("--split is an experiment of the graph-replayed fused one-candidate step")
            nb = B // args.split
            frac = torch.full((), 1.0 / args.split, device=device)
            for j in range(args.split):
                lpc, lq, ls = (x[j * nb:(j + 1) * nb].detach().clone().requires_grad_(True) for x in (pc, q, s))
                lgt, lst = gt[j * nb:(j + 1) * nb].clone(), torch.cuda.Stream(device)

                def part_step(lpc=lpc, lq=lq, ls=ls, lgt=lgt):
                    lpc.grad = lq.grad = ls.grad = None
                    loss, _, _ = R.pointcloud_project_loss(cfg, lpc, lq, None, None, kern, scaling_factor=ls, gt=lgt, num_candidates=1)
                    loss.backward(gradient=frac)   # every part's loss is a mean over B/n clouds
                    return loss
                with torch.cuda.stream(lst):
                    for _ in range(3):
                        part_step()
                    lst.synchronize()
                    lg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(lg, stream=lst, capture_error_mode="thread_local"):
                        part_step()
                parts.append((lg, lst))

        def run_step(i):
            if parts:
                for lg, lst in parts:
                    lst.wait_stream(side)      # fork: the step starts when the previous one has been joined
                    with torch.cuda.stream(lst):
                        lg.replay()
                for lg, lst in parts:
                    side.wait_stream(lst)      # join
                return
            if plan is not None:
                plan.run()
            elif lanes and i % args.in_flight:
                lg, lst = lanes[i % args.in_flight - 1]
                with torch.cuda.stream(lst):
                    lg.replay()
            elif graph is not None:
                graph.replay()
            else:
                last[0] = step()

        for i in range(args.warmup):
            run_step(i)
        torch.cuda.synchronize(device)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(device)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(side)
        for i in range(args.steps):
            run_step(i)
        ev1.record(side)
        torch.cuda.synchronize(device)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(device)
        wall = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1) if not lanes else wall * 1e3   # several streams: the events of one do not bracket the others

        tt = torch.tensor([wall], device=device, dtype=torch.float64)
        if dist_on:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        loss_t = last[0].detach().clone().reshape(1) if last[0] is not None else torch.zeros(1, device=device)
        if dist_on:
            dist.all_reduce(loss_t, op=dist.ReduceOp.SUM)  # the only exchange: the final step's loss, once
        wall = tt.item()

        # per-kernel durations (eager pass, library's event profiler) -> dominant kernel
        kern_ms = {}
        if rank == 0:
            prof = _native.profile_kernels(lambda: [(plan.run() if plan is not None else step()) for _ in range(30)], device)
            # An empty event pair executes two markers back to back; a bracketed kernel exposes one of them, so
            # half the empty-pair reading is subtracted (reproduces rocprofv3's kernel durations to ~0.3 us here).
            floor_ms = 0.5 * _native.event_pair_overhead_ms(device)
            kern_ms = {k: max(sum(v[len(v) // 3:]) / len(v[len(v) // 3:]) - floor_ms, 1e-6) for k, v in prof.items()}

        # SURVEY 8(d) extras on rank 0, outside the contract's timed region: median / best step over short windows, and
        # the forward alone (no_grad: locate, slab kernel, ray march with the loss)
        extras = {}
        if rank == 0 and graph is not None and not args.no_extras:
            def window_us(replay, n):
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(side)
                for _ in range(n):
                    replay()
                b_.record(side)
                b_.synchronize()
                return 1e3 * a.elapsed_time(b_) / n
            wins = sorted(window_us(plan.run if plan is not None else graph.replay, 20) for _ in range(15))
            extras["step_us"] = {"median": wins[len(wins) // 2], "best": wins[0], "windows": "15 x 20 steps"}
            if plan is not None:   # the same kernels as a replayed HIP graph of the autograd path (rounds 1-2), for comparison
                gw = sorted(window_us(graph.replay, 20) for _ in range(15))
                extras["hip_graph_replay"] = {"median_us": gw[len(gw) // 2], "best_us": gw[0], "point_clouds_per_sec": B / (gw[len(gw) // 2] * 1e-6),
                                              "packet_capture": os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0",
                                              "note": "autograd path captured once and replayed; not the value of this line"}
            if args.api == "fused" and ns == 1 and not lanes and not parts:
                # NOT the contract's number: two INDEPENDENT batches in flight (second graph, second stream, no join between
                # steps) -- how much of the one-chain step is latency and kernel tails that a second chain fills
                lg, lst = new_lane()
                torch.cuda.synchronize(device)
                n2 = 200
                for phase in range(2):   # first pass warms up
                    t2 = time.perf_counter()
                    for i in range(n2):
                        if i & 1:
                            with torch.cuda.stream(lst):
                                lg.replay()
                        else:
                            graph.replay()
                    torch.cuda.synchronize(device)
                    t2 = time.perf_counter() - t2
                extras["two_batches_in_flight"] = {"point_clouds_per_sec": B * n2 / t2, "us_per_batch": 1e6 * t2 / n2,
                                                   "note": "independent batches on two streams; not the value of this line"}

            if args.api == "fused" and ns == 1 and K_CAND == 1 and not lanes and not parts:
                # NOT the contract's number either: the reference's OWN call sequence through the drop-in signatures
                # (ModelPointCloud.compute_projection + add_proj_loss, dpc/models/model_pc_to.py:262-269, 339-385) --
                # smoothing_kernel, pointcloud_project_fast, the loss as torch ops, backward -- launched EAGERLY, no graph:
                # what a caller that changes nothing but the import gets; host-bound, so this is a figure about the Python
                # layer (round 3: 63 k clouds/s on the same box type), and the same sequence captured and replayed
                try:
                    def plain_step():
                        k2 = R.smoothing_kernel(cfg, SIGMA_REL)
                        pc.grad = q.grad = s.grad = None
                        proj = R.pointcloud_project_fast(cfg, pc, q, None, None, k2, scaling_factor=s)["proj"]
                        l = ((proj - gt) ** 2).sum() / B
                        l.backward()
                        return l
                    for _ in range(20):
                        plain_step()
                    torch.cuda.synchronize(device)
                    n3 = 200
                    t3 = time.perf_counter()
                    for _ in range(n3):
                        plain_step()
                    torch.cuda.synchronize(device)
                    t3 = time.perf_counter() - t3
                    extras["plain_eager"] = {"point_clouds_per_sec": B * n3 / t3, "us_per_step": 1e6 * t3 / n3,
                                             "note": "pointcloud_project_fast + torch loss + backward, eager launches (host-bound); "
                                                     "not the value of this line"}
                    pgraph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(pgraph, stream=side, capture_error_mode="thread_local"):
                        plain_step()
                    pw = sorted(window_us(pgraph.replay, 20) for _ in range(15))
                    extras["plain_graph_replay"] = {"median_us": pw[len(pw) // 2], "best_us": pw[0],
                                                    "point_clouds_per_sec": B / (pw[len(pw) // 2] * 1e-6),
                                                    "note": "the same call sequence captured once and replayed; not the value of this line"}

                except Exception as exc:   # an extra beside the value must never cost the line itself
                    extras["plain_eager"] = {"error": repr(exc)[:200]}

            def fwd_only():
                with torch.no_grad():
                    return R.pointcloud_project_loss(cfg, pc, q, None, None, kern, scaling_factor=s, gt=gt, num_candidates=K_CAND)[0]
            for _ in range(3):
                fwd_only()
            side.synchronize()
            fgraph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(fgraph, stream=side, capture_error_mode="thread_local"):
                fwd_only()
            fw = sorted(window_us(fgraph.replay, 20) for _ in range(15))
            extras["forward_only_us"] = {"median": fw[len(fw) // 2], "best": fw[0]}

    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return

    clouds_per_s = world * B * args.steps / wall
    ms_per_step = 1e3 * wall / args.steps
    a_bytes = step_bytes_per_cloud(N_PTS, G, K_CAND)   # = A(N,G) of SURVEY 8(d) when every cloud runs its backward
    kb = kernel_bytes_per_cloud(N_PTS, G)
    dom, dom_from = dominant_kernel({k: v for k, v in kern_ms.items() if k in kb}, args.config)
    # clouds a launch really works on: the backward kernels skip the losing pose candidates
    live = {k: (B // K_CAND if k in ("k_zcol_bwd", "k_gather_hw") else B) for k in kb}
    roofline = None
    if dom is not None:
        ach = live[dom] * kb[dom] / (kern_ms[dom] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": measured_traffic(dom, args.config), "avg_launch_us": 1e3 * kern_ms[dom],
                    "algorithmic_bytes_per_launch": live[dom] * kb[dom], "chosen_by": dom_from,
                    # every kernel of the step the same way (the two slab kernels are level and used to swap places here)
                    "all_kernels": {k: {"frac": live[k] * kb[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_us": 1e3 * v,
                                        "algorithmic_bytes_per_launch": live[k] * kb[k], "traffic": measured_traffic(k, args.config)}
                                    for k, v in sorted(kern_ms.items()) if k in kb}}
    step_ach = (B * a_bytes) / (dev_ms * 1e-3 / args.steps) / 1e9
    # what the step's launches really move (sum of the committed PMC figures) against the same step time
    own = [measured_traffic(k, args.config) for k in kern_ms if k in kb]
    own_bytes = sum(own) if own and all(x is not None for x in own) else None
    out = {
        "metric": "point-clouds/sec (8000 pts->64^3->128^2 proj) fwd+bwd", "value": clouds_per_s,
        "unit": "point-clouds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE %s: B=%d clouds x %d pts -> %d^3 grid, 21-tap Gaussian sigma_rel=%g "
                               "(sigma=0.01), DRC silhouette %dx%d vs mean-pooled %dx%d mask, %s, backward to pc, q, s"
                               % ("configs[1] (c2)" if args.config == "c2" else args.config, B, N_PTS, G, SIGMA_REL, G, G,
                                  2 * G, 2 * G, "loss sum((proj-gt)^2)/B" if K_CAND == 1 else
                                  "min-of-%d pose-candidate loss" % K_CAND),
                   "clouds_per_gpu": B, "points": N_PTS, "grid": G, "taps": KSIZE, "sigma_rel": SIGMA_REL,
                   "launch": "one native call per step (dpc_project_loss_step), no HIP graph" if plan is not None else
                             ("eager" if graph is None else "hip-graph replay"), "streams": ns, "batches_in_flight": args.in_flight, "split": args.split,
                   "hip_graph_packet_capture": None if plan is not None else os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0",
                   "api": args.api,
                   "sharding": "clouds, no collective"},
        "roofline": roofline,
        "roofline_step": {"bound": "hbm", "achieved": step_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": step_ach / HBM_PEAK_GBS, "algorithmic_bytes_per_cloud": a_bytes,
                          "device_ms_per_step": dev_ms / args.steps,
                          "own_traffic_bytes_per_step": own_bytes,
                          "own_traffic_frac": None if own_bytes is None else own_bytes / (dev_ms * 1e-3 / args.steps) / 1e9 / HBM_PEAK_GBS,
                          "note": "whole fwd+bwd step on rank 0 (HIP events on the launch stream) priced with A(N,G) of SURVEY 8(d)"
                                  + ("" if K_CAND == 1 else "; losing pose candidates run no backward, their backward bytes are not counted")},
        "kernels_us": {k: 1e3 * v for k, v in sorted(kern_ms.items())},
        "kernels_gbs": {k: live[k] * kb[k] / (v * 1e-3) / 1e9 for k, v in sorted(kern_ms.items()) if k in kb},
        "loss_mean": float(loss_t.item() / world),
    }
    out.update(extras)
    if not args.no_cpu_baseline and world == 1:  # the CPU leg is a single-GPU-run item (rank 0 at N=1 only)
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Full-step harness against fixture F10: the reference's own ModelPointCloud run on a tiny
configuration (tests/golden/make_golden.py::f10_full_step).  CPU part: networks and the loss pieces that need no
renderer.  GPU part: the whole step through the HIP renderer -- silhouettes, winners, loss, every parameter gradient."""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Cfg(dict):
    __getattr__ = dict.__getitem__


@pytest.fixture(scope="module")
def f10():
    cfg = Cfg(json.load(open(os.path.join(GOLDEN, "f10_config.json"))))
    g = np.load(os.path.join(GOLDEN, "f10_full_step.npz"))
    state = {k[len("state/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
    grads = {k[len("grad/"):]: g[k] for k in g.files if k.startswith("grad/")}
    return cfg, g, state, grads


def test_networks_match_reference(f10):
    """Same parameter names as the reference's checkpoint, and bit-identical activations on CPU."""
    from dpc.harness import StepNets

    cfg, g, state, _ = f10
    nets = StepNets(cfg)
    assert set(nets.state_dict()) == set(state)
    nets.load_state_dict(state)
    enc = nets.encoder(torch.from_numpy(g["images"]))
    first = enc["ids"][::cfg.step_size]
    assert torch.equal(nets.decoder(first), torch.from_numpy(g["points_1"]))
    assert torch.equal(nets.scalePred(first), torch.from_numpy(g["scaling_factor"]))
    pose = nets.poseNet(enc["poses"])
    assert torch.equal(pose["poses"], torch.from_numpy(g["poses"]))
    assert torch.equal(pose["pose_student"], torch.from_numpy(g["pose_student"]))


def test_fresh_initialisation():
    """Xavier-uniform weights, bias 0.01 (nets/*.py init_weights)."""
    from dpc.harness import StepNets

    cfg = Cfg(json.load(open(os.path.join(GOLDEN, "f10_config.json"))))
    nets = StepNets(cfg)
    w = nets.encoder.fc1[0].weight
    bound = (6.0 / (w.shape[0] + w.shape[1])) ** 0.5
    assert float(w.detach().abs().max()) <= bound and float(w.detach().abs().max()) > 0.8 * bound
    assert all(bool((m.bias == 0.01).all()) for m in nets.encoder.modules() if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)))


def test_loss_pieces_cpu(f10):
    """Mask pooling, min-of-K projection loss (oracle) and the student loss add up to the reference's total."""
    from dpc.harness import pooled_masks, student_loss
    from oracle import dpc_oracle as O

    cfg, g, _, _ = f10
    gt = pooled_masks(torch.from_numpy(g["masks"]), cfg.vox_size)
    assert torch.equal(gt, torch.from_numpy(g["pooled_masks"]))
    K = cfg.pose_predict_num_candidates
    proj_loss, winner = O.proj_loss_pose_candidates(gt.double(), torch.from_numpy(g["projs"]), K)
    assert np.array_equal(winner.numpy(), g["min_loss"])
    stud = student_loss(torch.from_numpy(g["poses"]), torch.from_numpy(g["pose_student"]), winner, K,
                        cfg.pose_predictor_student_loss_weight)
    assert abs(float(proj_loss + stud) * cfg.proj_weight - float(g["loss"])) <= 1e-9 * abs(float(g["loss"]))


@pytest.mark.gpu
def test_full_step_matches_reference(f10):
    """Forward + backward of one step on the GPU against the reference model's outputs and parameter gradients."""
    from dpc.harness import TrainStep

    cfg, g, state, grads = f10
    dev = torch.device("cuda")
    step = TrainStep(cfg, dev)
    step.load_reference_state(state)
    total, out = step.loss(torch.from_numpy(g["images"]).to(dev), torch.from_numpy(g["masks"]).to(dev), global_step=0)
    total.backward()

    def close(a, ref, tol, what):
        a = a.detach().double().cpu().numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        err = float(np.abs(a - ref).max())
        assert err <= tol * scale, "%s: %.3e > %.1e * %.3g" % (what, err, tol, scale)

    close(out["points_1"], g["points_1"], 1e-5, "points")
    close(out["poses"], g["poses"], 1e-5, "poses")
    close(out["projs"], g["projs"], 1e-5, "silhouettes")
    assert np.array_equal(out["min_loss"].cpu().numpy(), g["min_loss"])
    assert abs(float(total.detach()) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    named = dict(step.nets.named_parameters())
    assert {k for k, p in named.items() if p.grad is None} == set(g["no_grad"].tolist())
    for name, ref in grads.items():
        # network gradients are sums of the renderer's 1e-5-accurate gradients over thousands of points
        close(named[name].grad, ref, 1e-4, "grad " + name)


@pytest.mark.gpu
def test_training_reduces_loss(f10):
    """A few Adam steps on one fixed batch make the loss go down (the step is wired end to end)."""
    from dpc.harness import TrainStep

    cfg, g, state, _ = f10
    dev = torch.device("cuda")
    step = TrainStep(cfg, dev, lr=1e-3)
    step.load_reference_state(state)
    images, masks = torch.from_numpy(g["images"]).to(dev), torch.from_numpy(g["masks"]).to(dev)
    losses = [float(step(images, masks)) for _ in range(12)]
    assert step.global_step == 12 and losses[-1] < losses[0]

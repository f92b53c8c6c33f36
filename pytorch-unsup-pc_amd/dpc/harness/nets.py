"""The four networks of the reference's ModelPointCloud, restated from their published structure.

    reference                                    here
    dpc/nets/img_encoder_to.py:15-67   Encoder   strided 5x5 conv, then (stride-2 3x3, 3x3) blocks down to 4x4, three FCs, pose FC
    dpc/nets/pc_decoder_to.py:15-56    Decoder   one FC to N*3, tanh (/2 for the unit cube); rgb heads kept for checkpoint parity
    dpc/nets/pose_net_to.py:15-86      PoseNet   K candidate MLPs + a student MLP (or a single FC), optional translation FC
    dpc/models/model_pc_to.py:113-130  ScalePredictor  FC + sigmoid * pc_occupancy_scaling_maximum

Every Linear/Conv2d starts from Xavier-uniform weights and bias 0.01, like the reference.  Module and parameter names
(encoder.conv_layers.0.weight, poseNet.candidate_fcs.2.layers.4.bias, ...) match the reference's state_dict.
"""
import math

import torch
import torch.nn as nn


def _fresh(module):
    for m in module.modules():
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            nn.init.xavier_uniform_(m.weight)
            nn.init.constant_(m.bias, 0.01)
    return module


def _stack(widths, last_activation):
    """Linear layers widths[0] -> ... -> widths[-1] with LeakyReLU between them (Linear at even indices)."""
    mods = []
    for i, (a, b) in enumerate(zip(widths[:-1], widths[1:])):
        mods.append(nn.Linear(a, b))
        if last_activation or i + 2 < len(widths):
            mods.append(nn.LeakyReLU())
    return nn.Sequential(*mods)


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        size, channels = cfg.input_shape[0], cfg.input_shape[2]
        width = cfg.f_dim
        convs = [nn.Conv2d(channels, width, 5, stride=2, padding=2), nn.LeakyReLU()]
        for _ in range(int(math.log2(size / 4) - 1)):  # halve the resolution until 4 x 4 remains
            convs += [nn.Conv2d(width, 2 * width, 3, stride=2, padding=1), nn.LeakyReLU(),
                      nn.Conv2d(2 * width, 2 * width, 3, stride=1, padding=1), nn.LeakyReLU()]
            width *= 2
        self.conv_layers = nn.Sequential(*convs)
        self.fc1 = _stack([width * 16, cfg.fc_dim], True)
        self.fc2 = _stack([cfg.fc_dim, cfg.fc_dim], True)
        self.fc3 = _stack([cfg.fc_dim, cfg.z_dim], True)
        self.pose_fc = nn.Linear(cfg.fc_dim, cfg.z_dim) if cfg.predict_pose else None
        _fresh(self)

    def forward(self, images):
        feat = self.conv_layers(images * 2 - 1).flatten(1)
        h1 = self.fc1(feat)
        h2 = self.fc2(h1)
        out = {"conv_features": feat, "z_latent": h1, "ids": self.fc3(h2)}
        if self.pose_fc is not None:
            out["poses"] = self.pose_fc(h2)
        return out


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.num_points, self.unit_cube = cfg.pc_num_points, bool(cfg.pc_unit_cube)
        d = cfg.fc_dim
        self.pts_raw_fc = nn.Linear(d, 3 * self.num_points)
        # colour heads: dead in the reference's live configuration (pc_rgb: false) but part of its checkpoints
        self.rgb_deep_decoder = _stack([d, d, d, d], True)
        self.rgb_raw_dec = nn.Linear(d, 3 * self.num_points)
        _fresh(self)

    def forward(self, code):
        xyz = torch.tanh(self.pts_raw_fc(code).reshape(-1, self.num_points, 3))
        return xyz / 2.0 if self.unit_cube else xyz


class _PoseBranch(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        depth = cfg.pose_candidates_num_layers
        self.layers = _stack([cfg.z_dim] + [32] * (depth - 1) + [4], False)

    def forward(self, x):
        return self.layers(x)


class PoseNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.num_candidates = cfg.pose_predict_num_candidates
        self.with_student = bool(cfg.pose_predictor_student)
        if self.num_candidates > 1:
            self.candidate_fcs = nn.ModuleList(_PoseBranch(cfg) for _ in range(self.num_candidates))
            self.student_fc = _PoseBranch(cfg)
        else:
            self.single_candidate_fc = nn.Linear(cfg.z_dim, 4)
        self.trans_fc = nn.Linear(cfg.z_dim, 3) if cfg.predict_translation else None
        self.trans_tanh, self.trans_scale = bool(cfg.predict_translation_tanh), cfg.predict_translation_scaling_factor
        _fresh(self)

    def forward(self, code):
        out = {"pose_student": None}
        if self.num_candidates > 1:
            # candidate-minor order: rows s*K + k, what tf_repeat_0 of the per-sample tensors lines up with
            out["poses"] = torch.cat([branch(code) for branch in self.candidate_fcs], dim=1).reshape(-1, 4)
            if self.with_student:
                out["pose_student"] = self.student_fc(code)
        else:
            out["poses"] = self.single_candidate_fc(code)
        t = None
        if self.trans_fc is not None:
            t = self.trans_fc(code)
            if self.trans_tanh:
                t = torch.tanh(t) * self.trans_scale
        out["predicted_translation"] = t
        return out


class ScalePredictor(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fc = _fresh(nn.Linear(cfg.z_dim, 1))
        self.maximum = cfg.pc_occupancy_scaling_maximum

    def forward(self, code):
        return torch.sigmoid(self.fc(code)) * self.maximum


class _FocalHead(nn.Module):  # present in the reference's checkpoints; unused unless learn_focal_length
    def __init__(self, cfg):
        super().__init__()
        self.fc = nn.Linear(cfg.z_dim, 1)
        self.mean, self.range = getattr(cfg, "focal_length_mean", 0.0), getattr(cfg, "focal_length_range", 1.0)

    def forward(self, code):
        return self.mean + torch.sigmoid(self.fc(code)) * self.range


class StepNets(nn.Module):
    """Container with the reference's attribute names: load_state_dict() accepts its 'model_state_dict'."""

    def __init__(self, cfg):
        super().__init__()
        self.encoder, self.decoder, self.poseNet = Encoder(cfg), Decoder(cfg), PoseNet(cfg)
        self.scalePred, self.focalPred = ScalePredictor(cfg), _FocalHead(cfg)

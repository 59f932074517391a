"""NeRF field module with the reference's parameter names, shapes and attributes (models/nerf.py:5-78), so
state_dicts interchange with the reference and Lightning checkpoints load unchanged.

When called through upnerf_amd.render_rays the module is evaluated by the fused HIP kernels from its packed
parameter buffer (`packed()`); `forward()` keeps the reference's per-sample calling convention
(nerf.py:80-124) and runs layer by layer on the HIP linear kernel."""
from __future__ import annotations

import struct

import torch
from torch import nn

from .ops import hip_linear
from .packing import NerfPacker


def fp32_round(x: float) -> float:
    return struct.unpack("f", struct.pack("f", float(x)))[0]


class NeRF(nn.Module):
    def __init__(self, typ, D=8, W=256, skips=[4], encode_feat=True, feat_dim=384, xyz_L=10, dir_L=8,
                 appearance_dim=48, candidate_dim=16, c2f=None):
        super().__init__()
        self.typ, self.D, self.W, self.skips = typ, D, W, list(skips)
        self.xyz_L, self.dir_L = xyz_L, dir_L
        self.in_channels_xyz, self.in_channels_dir = 6 * xyz_L + 3, 6 * dir_L + 3
        self.feat_dim, self.appearance_dim, self.candidate_dim = feat_dim, appearance_dim, candidate_dim
        self.encode_feat = encode_feat
        self.encode_appearance = appearance_dim > 0
        self.encode_candidate = candidate_dim > 0
        self.c2f = c2f
        self.progress = nn.Parameter(torch.tensor(0.0))  # written through .data only
        # Host mirror of `progress`: render_rays needs the value on the host (band weights are kernel arguments) and
        # reading the device parameter would drain the HIP queue twice per step.  None = unknown (read the parameter).
        self.host_progress = None
        for i in range(D):
            k = self.in_channels_xyz if i == 0 else (W + self.in_channels_xyz if i in self.skips else W)
            setattr(self, f"xyz_encoding_{i + 1}", nn.Sequential(nn.Linear(k, W), nn.ReLU(True)))
        self.xyz_encoding_final = nn.Linear(W, W)
        self.share_sigma = nn.Sequential(nn.Linear(W, 1), nn.Softplus())
        # encode_feat = False (nerf.py:52-56, 75-78): no feature layer -- the colour head reads xyz_encoding_final itself and
        # the candidate head ends in a 3-wide colour layer instead of the feature one
        if encode_feat:
            self.feat_share_layer = nn.Linear(W, feat_dim)
        in_rgb = (feat_dim if encode_feat else W) + self.in_channels_dir + (appearance_dim if self.encode_appearance else 0)
        self.rgb_share_layer = nn.Sequential(nn.Linear(in_rgb, W // 2), nn.ReLU(True), nn.Linear(W // 2, 3),
                                             nn.Sigmoid())
        if self.encode_candidate:
            self.candidate_encoding = nn.Sequential(nn.Linear(W + candidate_dim, W // 2), nn.ReLU(True),
                                                    nn.Linear(W // 2, W // 2), nn.ReLU(True))
            self.candidate_sigma = nn.Sequential(nn.Linear(W // 2, 1), nn.Softplus())
            if encode_feat:
                self.feat_candidate_layer = nn.Linear(W // 2, feat_dim)
            else:
                self.rgb_candidate_layer = nn.Linear(W // 2, 3)
        self.packer = NerfPacker(W, D, self.skips, self.in_channels_xyz, self.in_channels_dir, feat_dim,
                                 appearance_dim, candidate_dim, encode_feat=encode_feat)

    def set_progress(self, progress: float):
        """Write `progress` (device parameter, as the reference does through .data) and its host mirror.  The mirror
        holds the fp32-rounded value -- what the reference reads back with `progress.data.item()` (nerf.py:94,
        nerf_system.py:180) and what a checkpoint restores."""
        self.host_progress = fp32_round(progress)
        # The device parameter is written ON DEMAND (flush_progress): every reader inside this package uses the host mirror, so a
        # fill launch per model and step bought nothing; state_dict() / checkpoints / the nn.Module forward flush first.
        self._progress_stale = True

    def flush_progress(self):
        """Bring the device parameter `progress` up to date with the host mirror (state_dict, checkpoints, module forward)."""
        if getattr(self, "_progress_stale", False) and self.host_progress is not None:
            self.progress.data.fill_(float(self.host_progress))
        self._progress_stale = False

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_progress()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.host_progress = None  # a checkpoint may carry another progress value
        self._progress_stale = False

    def packed(self, override=None) -> torch.Tensor:
        """Flat kernel-layout parameter buffer, differentiable w.r.t. the parameters.  override: name -> tensor standing in for the
        parameter of that name (aliases from ops.fanout for parameters that have a second consumer in the step)."""
        p = dict(self.named_parameters())
        if override:
            p.update(override)
        if p["xyz_encoding_1.0.weight"].is_cuda:
            return self.packer.pack_hip(p)  # 4 HIP launches; hand-written backward
        return self.packer.pack(p)  # torch restatement (CPU tests)

    # ---- reference-compatible per-sample call (nerf.py:80-124); not used by render_rays
    def positional_encoding(self, x, L):
        freq = (2 ** torch.arange(L, dtype=torch.float32, device=x.device)) * torch.pi
        arg = x[..., None] * freq
        enc = torch.stack([arg.sin(), arg.cos()], dim=-2)
        if self.c2f is not None:
            start, end = self.c2f
            self.flush_progress()
            alpha = (self.progress.data - start) / (end - start) * L
            k = torch.arange(L, dtype=torch.float32, device=x.device)
            enc = enc * ((1 - ((alpha - k).clamp(min=0, max=1) * torch.pi).cos()) / 2)
        return torch.cat([x, enc.reshape(*x.shape[:-1], -1)], -1)

    def forward(self, inputs, sched_mult, sigma_only=False):
        lin = lambda m, x, relu=False: hip_linear(x, m.weight, m.bias, relu)
        ret = {}
        x0 = self.positional_encoding(inputs["input_xyz"], self.xyz_L)
        h = x0
        for i in range(self.D):
            if i in self.skips:
                h = torch.cat([x0, h], 1)
            h = lin(getattr(self, f"xyz_encoding_{i + 1}")[0], h, True)
        ret["s_sigma"] = torch.nn.functional.softplus(lin(self.share_sigma[0], h))
        if sigma_only:
            return ret
        e = lin(self.xyz_encoding_final, h)
        if self.encode_feat:
            ret["s_feat"] = lin(self.feat_share_layer, e)
        if sched_mult < 1 and (self.encode_candidate or not self.encode_feat):  # (nerf.py:97 / 119: no flag check without features)
            g = lin(self.candidate_encoding[0], torch.cat([e, inputs["input_c"]], 1), True)
            g = lin(self.candidate_encoding[2], g, True)
            ret["c_sigma"] = torch.nn.functional.softplus(lin(self.candidate_sigma[0], g))
            if self.encode_feat:
                ret["c_feat"] = lin(self.feat_candidate_layer, g)
            else:
                ret["c_rgb"] = lin(self.rgb_candidate_layer, g)
        if sched_mult > 0 or not self.encode_feat:  # (nerf.py:110-117: always, without features)
            parts = [ret["s_feat"] if self.encode_feat else e, self.positional_encoding(inputs["input_dir"], self.dir_L)]
            if self.encode_appearance:
                parts.append(inputs["input_a"])
            r = lin(self.rgb_share_layer[0], torch.cat(parts, 1), True)
            ret["s_rgb"] = torch.sigmoid(lin(self.rgb_share_layer[2], r))
        return ret

"""UPNeRFLoss (losses.py:13-64): schedule-weighted depth / feature / colour / uncertainty terms on per-ray maps."""
from __future__ import annotations

import torch
from torch import nn


class UPNeRFLoss(nn.Module):
    def __init__(self, depth_mult=1e-4, alpha_reg=1.0, encode_feat=True, fine=True):
        super().__init__()
        self.depth_mult, self.alpha_reg, self.encode_feat, self.fine = depth_mult, alpha_reg, encode_feat, fine
        if not encode_feat:
            raise NotImplementedError("nerf.feat_dim = 0 is not implemented on the HIP path")

    def forward(self, inputs, rgb_targets, feat_targets, depth_targets, schedule_mult):
        m, ret = schedule_mult, {}
        for typ, tag in (("coarse", "c"), ("fine", "f")):
            if typ == "fine" and not self.fine:
                break
            if m < 1:
                l = (inputs[f"s_depth_{typ}"] - depth_targets).abs()
                if f"t_weight_{typ}" in inputs:
                    l = l * (1 - inputs[f"t_weight_{typ}"].detach())
                ret[f"l_depth_{tag}"] = l.mean() * self.depth_mult * (1 - m)
                ret[f"l_feat_{tag}"] = ((inputs[f"feat_{typ}"] - feat_targets) ** 2).mean() * (1 - m)
            if m > 0:
                sq = (inputs[f"s_rgb_{typ}"] - rgb_targets) ** 2
                if typ == "coarse":
                    ret["l_rgb_c"] = sq.mean() * m / 2
                else:
                    ret["l_rgb_f"] = (sq / (2 * inputs["t_beta"] ** 2)).mean() * m
                    ret["l_beta"] = torch.log(inputs["t_beta"]).mean() * m
                    ret["l_alpha"] = inputs["t_alpha"].mean() * self.alpha_reg * m
        return ret

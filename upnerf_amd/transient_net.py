"""TransientNet (models/transient_net.py:5-38) with the reference's parameter names; every Linear runs on the
fp32 MFMA kernel (upnerf_linear / upnerf_wgrad).  One row per RAY (M = batch size), ~0.1 % of the step's FLOPs."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ops import embed_rows, hip_linear


class TransientNet(nn.Module):
    def __init__(self, N_images, beta_min=0.1, trasient_dim=128, feat_dim=384):
        super().__init__()
        self.beta_min, self.trasient_dim = beta_min, trasient_dim
        self.embedding_t = nn.Embedding(N_images, trasient_dim)
        self.feat_encoder = nn.Sequential(nn.Linear(feat_dim, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                                          nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU())
        self.final_encoder = nn.Linear(256, 256)
        self.t_encoder = nn.Sequential(nn.Linear(256 + trasient_dim, 128), nn.ReLU())
        self.alpha_layer = nn.Sequential(nn.Linear(256, 1), nn.Sigmoid())
        self.beta_layer = nn.Sequential(nn.Linear(128, 1), nn.Softplus())
        self.rgb_layer = nn.Sequential(nn.Linear(128, 3), nn.Sigmoid())

    def forward(self, feat, ts):
        lin = lambda m, x, relu=False: hip_linear(x, m.weight, m.bias, relu, defer_wgrad=True)  # sole consumers
        h = feat
        for i in (0, 2, 4, 6):
            h = lin(self.feat_encoder[i], h, True)
        e = lin(self.final_encoder, h)
        t = lin(self.t_encoder[0], torch.cat([e, embed_rows(self.embedding_t, ts, defer_grad=True)], -1), True)
        alpha = torch.sigmoid(lin(self.alpha_layer[0], h))
        rgb = torch.sigmoid(lin(self.rgb_layer[0], t))
        beta = F.softplus(lin(self.beta_layer[0], t)) * alpha + self.beta_min
        return {"alpha": alpha, "rgb": rgb, "beta": beta}

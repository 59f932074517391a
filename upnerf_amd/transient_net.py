"""TransientNet (models/transient_net.py:5-38) with the reference's parameter names.  One row per RAY (M = batch size),
~0.1 % of the step's FLOPs -- and, as one launch per layer, ~0.5 ms of a 17.7 ms step: with the reference's widths (feat 384,
hidden 256, embedding 128) the whole network runs as ONE forward and ONE backward launch (csrc/transient.hip) plus one grouped
weight-gradient launch; any other shape, and CPU tensors, take the per-layer path (fp32 MFMA kernel per nn.Linear)."""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import check, lib, ptr, stream
from .ops import _DeferredWgrads, embed_rows, hip_linear, workspace

FUSED = os.environ.get("UPNERF_TRANSIENT_FUSED", "1") != "0"


class _TransientFn(torch.autograd.Function):
    """alpha, rgb, beta = TransientNet(feat, t_emb) through upnerf_transient_fwd / upnerf_transient_bwd."""

    NAMES = ("w0", "b0", "w1", "b1", "w2", "b2", "w3", "b3", "wf", "bf", "wt", "bt", "wa", "ba", "wb", "bb", "wr", "br")

    @staticmethod
    def forward(ctx, beta_min, feat, t_emb, *params):
        R, dev = feat.shape[0], feat.device
        feat, t_emb = feat.detach().contiguous(), t_emb.detach().contiguous()
        p = [q.detach().contiguous() for q in params]
        e = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        h, ee, t = e(4, R, 256), e(R, 256), e(R, 128)
        alpha, rgb, beta, spre = e(R, 1), e(R, 3), e(R, 1), e(R)
        a = _lib.TransientArgs(R=R, beta_min=float(beta_min), feat=ptr(feat), t_emb=ptr(t_emb), h=ptr(h), e=ptr(ee), t=ptr(t),
                               alpha=ptr(alpha), rgb=ptr(rgb), beta=ptr(beta), spre=ptr(spre),
                               **{n: ptr(q) for n, q in zip(_TransientFn.NAMES, p)})
        check(lib.upnerf_transient_fwd(C.byref(a), stream()), "upnerf_transient_fwd")
        ctx.args, ctx.keep = a, (feat, t_emb, p, h, ee, t, alpha, rgb, beta, spre)
        ctx.set_materialize_grads(False)  # the training loss reads alpha and beta only: no zero-filled d_rgb (a fill launch)
        return alpha, rgb, beta

    @staticmethod
    def backward(ctx, d_alpha, d_rgb, d_beta):
        a = ctx.args
        feat, t_emb, p, h, ee, t, alpha, rgb, beta, spre = ctx.keep
        R, dev = feat.shape[0], feat.device
        e = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        c = lambda x: None if x is None else x.contiguous()
        d_alpha, d_rgb, d_beta = c(d_alpha), c(d_rgb), c(d_beta)
        dz, gz_t, gz_e, gz_h = e(R, 8), e(R, 128), e(R, 256), e(4, R, 256)
        g_feat = e(R, 384) if ctx.needs_input_grad[1] else None
        g_temb = e(R, 128) if ctx.needs_input_grad[2] else None
        g = _lib.TransientGrads(d_alpha=ptr(d_alpha), d_rgb=ptr(d_rgb), d_beta=ptr(d_beta), dz_heads=ptr(dz), gz_t=ptr(gz_t),
                                gz_e=ptr(gz_e), gz_h=ptr(gz_h), g_temb=ptr(g_temb), g_feat=ptr(g_feat))
        check(lib.upnerf_transient_bwd(C.byref(a), C.byref(g), stream()), "upnerf_transient_bwd")
        # weight gradients: one grouped launch + one fixed-order reduction over (pre-activation gradient, stored input) pairs
        gw = [e(*q.shape) for q in p[:12]]              # w0 b0 .. wt bt
        hw, ht, hb = e(8, 256), e(8, 128), e(8)          # the three heads: rows of dz_heads against h4 and t
        G = _lib.WgradGroup
        grp = lambda A, lda, N, B, ldb, K, dW, ldo, db: G(A=A, B=B, dW=dW, db=db, M=R, N=N, K=K, lda=lda, ldb=ldb, ldo=ldo)
        groups = [grp(ptr(gz_h[0]), 256, 256, ptr(feat), 384, 384, ptr(gw[0]), 384, ptr(gw[1]))]
        for l in (1, 2, 3):
            groups.append(grp(ptr(gz_h[l]), 256, 256, ptr(h[l - 1]), 256, 256, ptr(gw[2 * l]), 256, ptr(gw[2 * l + 1])))
        groups += [grp(ptr(gz_e), 256, 256, ptr(h[3]), 256, 256, ptr(gw[8]), 256, ptr(gw[9])),
                   grp(ptr(gz_t), 128, 128, ptr(ee), 256, 256, ptr(gw[10]), 384, ptr(gw[11])),
                   grp(ptr(gz_t), 128, 128, ptr(t_emb), 128, 128, gw[10].data_ptr() + 4 * 256, 384, None),
                   grp(ptr(dz), 8, 8, ptr(h[3]), 256, 256, ptr(hw), 256, ptr(hb)),
                   grp(ptr(dz), 8, 8, ptr(t), 128, 128, ptr(ht), 128, None)]
        arr = (G * len(groups))(*groups)
        ns = max(1, min(_DeferredWgrads.NSPLIT, R // 64))
        n = lib.upnerf_wgrad_grouped_scratch(arr, len(groups), ns)
        if n < 0:
            check(n, "upnerf_wgrad_grouped_scratch")
        ws = workspace("wgrad_transient", n, dev)
        check(lib.upnerf_wgrad_grouped(arr, len(groups), ptr(ws), ns, stream()), "upnerf_wgrad_grouped")
        grads = gw + [hw[0:1], hb[0:1], ht[1:2], hb[1:2], ht[2:5], hb[2:5]]   # wa ba wb bb wr br
        return (None, g_feat, g_temb) + tuple(grads)


class TransientNet(nn.Module):
    def __init__(self, N_images, beta_min=0.1, trasient_dim=128, feat_dim=384):
        super().__init__()
        self.beta_min, self.trasient_dim, self.feat_dim = beta_min, trasient_dim, feat_dim
        self.embedding_t = nn.Embedding(N_images, trasient_dim)
        self.feat_encoder = nn.Sequential(nn.Linear(feat_dim, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                                          nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU())
        self.final_encoder = nn.Linear(256, 256)
        self.t_encoder = nn.Sequential(nn.Linear(256 + trasient_dim, 128), nn.ReLU())
        self.alpha_layer = nn.Sequential(nn.Linear(256, 1), nn.Sigmoid())
        self.beta_layer = nn.Sequential(nn.Linear(128, 1), nn.Softplus())
        self.rgb_layer = nn.Sequential(nn.Linear(128, 3), nn.Sigmoid())

    def _fused_params(self):
        mods = (self.feat_encoder[0], self.feat_encoder[2], self.feat_encoder[4], self.feat_encoder[6], self.final_encoder,
                self.t_encoder[0], self.alpha_layer[0], self.beta_layer[0], self.rgb_layer[0])
        return [q for m in mods for q in (m.weight, m.bias)]

    def forward(self, feat, ts):
        if FUSED and feat.is_cuda and self.feat_dim == 384 and self.trasient_dim == 128 and feat.dtype == torch.float32:
            t_emb = embed_rows(self.embedding_t, ts, defer_grad=True)
            alpha, rgb, beta = _TransientFn.apply(self.beta_min, feat, t_emb, *self._fused_params())
            return {"alpha": alpha, "rgb": rgb, "beta": beta}
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

"""UPNeRFLoss (losses.py:13-64) fused with the depth-prior affine (models/nerf_system.py:169-177) on the GPU:
one HIP reduction kernel forward (upnerf_loss_fwd) and one elementwise kernel backward (upnerf_loss_bwd) instead of
~40 ATen launches.  Same constructor, call signature and term names as the reference."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import step_scalars, zero_pool
from ._lib import LossArgs, LossGrads, check, lib, ptr, stream

TERMS = ("l_depth_c", "l_feat_c", "l_rgb_c", "l_depth_f", "l_feat_f", "l_rgb_f", "l_beta", "l_alpha")


class _LossFn(torch.autograd.Function):
    """inputs (any may be None): depth_direct, inv_depth, scale_rows, s_depth_c, s_depth_f, t_weight_c, t_weight_f,
    feat_c, feat_f, feat_gt, rgb_c, rgb_f, rgb_gt, beta, alpha; returns (terms[8], depth_targets[R], total): `total` = the
    sum of the terms the phase uses (`sum(loss_d.values())`, nerf_system.py:183), written by the same two launches -- and its
    gradient enters the backward kernel directly (no select + reduce forward, no mask product backward)."""

    @staticmethod
    def forward(ctx, cfg, *tensors):
        m, depth_mult, alpha_reg, near, far, fine = cfg
        on = [m < 1, m < 1, m > 0, fine and m < 1, fine and m < 1, fine and m > 0, fine and m > 0, fine and m > 0]
        t = [None if x is None else x.detach().contiguous().float() for x in tensors]
        (dd, inv, rows, sdc, sdf, twc, twf, fc, ff, fg, rc, rf, rg, beta, alpha) = t
        ref = next(x for x in (dd, inv) if x is not None)
        R, dev = ref.shape[0], ref.device
        F = fg.shape[1] if fg is not None else 0
        dyn = step_scalars.current()  # graph capture: the multiplier follows the schedule through device memory
        a = LossArgs(R=R, F=F, fine=int(fine), has_tw=int(twc is not None), sched=float(m), depth_mult=depth_mult,
                     alpha_reg=alpha_reg, near=near, far=far, depth_direct=ptr(dd), inv_depth=ptr(inv),
                     depth_scale_rows=ptr(rows), s_depth_c=ptr(sdc), s_depth_f=ptr(sdf), t_weight_c=ptr(twc),
                     t_weight_f=ptr(twf), feat_c=ptr(fc), feat_f=ptr(ff), feat_gt=ptr(fg), rgb_c=ptr(rc), rgb_f=ptr(rf),
                     rgb_gt=ptr(rg), beta=ptr(beta), alpha=ptr(alpha),
                     sched_dev=dyn.ptr_named("sched", 1) if dyn else None)
        depth = torch.empty(R, device=dev)
        terms, total = torch.empty(8, device=dev), torch.empty((), device=dev)
        a.term_mask = sum(1 << k for k, b in enumerate(on) if b)
        a.total = ptr(total)
        scratch = torch.empty(64 * 8, device=dev)
        check(lib.upnerf_loss_fwd(C.byref(a), ptr(depth), ptr(terms), ptr(scratch), stream()), "upnerf_loss_fwd")
        ctx.args, ctx.keep = a, t  # `t` keeps the device buffers referenced by `a` alive
        ctx.mark_non_differentiable(depth)
        ctx.set_materialize_grads(False)  # (no zero tensor -- a fill launch -- for the output nothing differentiates)
        return terms, depth, total

    @staticmethod
    def backward(ctx, g_terms, _g_depth, g_total):
        if g_terms is None and g_total is None:
            return (None,) * 16
        a, t = ctx.args, ctx.keep
        (dd, inv, rows, sdc, sdf, twc, twf, fc, ff, fg, rc, rf, rg, beta, alpha) = t
        need = ctx.needs_input_grad[1:]
        # one zero fill for all gradient buffers (slices of one arena, each starting on a 256-byte boundary)
        want = [(x, ok) for x, ok in ((dd, need[0]), (rows, need[2]), (sdc, need[3]), (sdf, need[4]), (fc, need[7]), (ff, need[8]),
                                      (rc, need[10]), (rf, need[11]), (beta, need[13]), (alpha, need[14]))]
        pad = lambda n: (n + 63) // 64 * 64
        arena = zero_pool.zeros(sum(pad(x.numel()) for x, ok in want if x is not None and ok) or 1, (g_terms if g_terms is not None else g_total).device)
        cursor = [0]

        def new(x, ok):
            if x is None or not ok:
                return None
            o = cursor[0]
            cursor[0] += pad(x.numel())
            return arena[o:o + x.numel()].view(x.shape)

        d_dd, d_rows = new(dd, need[0]), new(rows, need[2])
        d_sdc, d_sdf, d_fc, d_ff = new(sdc, need[3]), new(sdf, need[4]), new(fc, need[7]), new(ff, need[8])
        d_rc, d_rf, d_beta, d_alpha = new(rc, need[10]), new(rf, need[11]), new(beta, need[13]), new(alpha, need[14])
        g = LossGrads(d_depth_scale_rows=ptr(d_rows), d_depth=ptr(d_dd), d_s_depth_c=ptr(d_sdc), d_s_depth_f=ptr(d_sdf),
                      d_feat_c=ptr(d_fc), d_feat_f=ptr(d_ff), d_rgb_c=ptr(d_rc), d_rgb_f=ptr(d_rf), d_beta=ptr(d_beta),
                      d_alpha=ptr(d_alpha))
        gt = g_terms.contiguous().float() if g_terms is not None else None
        gtot = g_total.contiguous().float() if g_total is not None else None
        a.g_total = ptr(gtot)
        try:
            check(lib.upnerf_loss_bwd(C.byref(a), ptr(gt), C.byref(g), stream()), "upnerf_loss_bwd")
        finally:
            a.g_total = None
        shp = lambda d, x: None if d is None else d.view_as(x)
        return (None, d_dd, None, d_rows, d_sdc, d_sdf, None, None, d_fc, d_ff, None, d_rc, d_rf, None,
                shp(d_beta, tensors_like(ctx, 13)), shp(d_alpha, tensors_like(ctx, 14)))


def tensors_like(ctx, i):
    return ctx.keep[i]


_CONSTS = {}


def _const(value: float, device):
    """A cached 0-dim fp32 constant on `device` (allocated once, outside any graph capture after the first eager step)."""
    key = (float(value), torch.device(device))
    if key not in _CONSTS:
        t = torch.full((), float(value), device=device, dtype=torch.float32)
        if t.is_cuda and torch.cuda.is_current_stream_capturing():
            return t  # memory of the capturing graph's pool: not to be kept beyond it
        _CONSTS[key] = t
    return _CONSTS[key]


def _terms_to_dict(terms, m, fine, encode_feat=True):
    out = {}
    fc, ff = ("l_feat_c", "l_feat_f") if encode_feat else ("l_c_rgb_c", "l_c_rgb_f")  # losses.py:30-35, 51-56
    if m < 1:
        out["l_depth_c"], out[fc] = terms[0], terms[1]
    if m > 0:
        out["l_rgb_c"] = terms[2]
    if fine:
        if m < 1:
            out["l_depth_f"], out[ff] = terms[3], terms[4]
        if m > 0:
            out["l_rgb_f"], out["l_beta"], out["l_alpha"] = terms[5], terms[6], terms[7]
    # the reference's dict order (losses.py:24-63): coarse terms, then fine terms
    order = ["l_depth_c", fc, "l_rgb_c", "l_depth_f", ff, "l_rgb_f", "l_beta", "l_alpha"]
    return {k: out[k] for k in order if k in out}


class UPNeRFLoss(nn.Module):
    def __init__(self, depth_mult=1e-4, alpha_reg=1.0, encode_feat=True, fine=True, near=0.1, far=5.0):
        super().__init__()
        self.depth_mult, self.alpha_reg, self.encode_feat, self.fine = depth_mult, alpha_reg, encode_feat, fine
        self.near, self.far = near, far

    def _run(self, inputs, rgb, feat, m, depth_direct=None, inv_depth=None, scale_rows=None):
        g = inputs.get
        beta = g("t_beta")
        alpha = g("t_alpha")
        fine = self.fine
        # encode_feat = False (losses.py:33-35, 54-56): the candidate colour map against the colour targets takes the place of
        # the feature map against the feature targets -- the same mean squared difference, three columns wide
        fk, ft = ("feat_", feat) if self.encode_feat else ("c_rgb_", rgb)
        cfg = (m, self.depth_mult, self.alpha_reg, self.near, self.far, fine)
        terms, depth, total = _LossFn.apply(
            cfg, depth_direct, inv_depth, scale_rows,
            g("s_depth_coarse") if m < 1 else None, g("s_depth_fine") if (m < 1 and fine) else None,
            g("t_weight_coarse") if m < 1 else None, g("t_weight_fine") if (m < 1 and fine) else None,
            g(fk + "coarse") if m < 1 else None, g(fk + "fine") if (m < 1 and fine) else None, ft if m < 1 else None,
            g("s_rgb_coarse") if m > 0 else None, g("s_rgb_fine") if (m > 0 and fine) else None, rgb if m > 0 else None,
            beta.reshape(-1) if (beta is not None and m > 0 and fine) else None,
            alpha.reshape(-1) if (alpha is not None and m > 0 and fine) else None)
        self.last_terms, self.last_total = (terms, m, fine), total
        return _terms_to_dict(terms, m, fine, self.encode_feat), depth

    def total(self):
        """Sum of the terms of the last call as one autograd node (what `sum(loss_d.values())` computes; the summation order,
        hence the last bit of the value, may differ -- the gradients do not)."""
        return self.last_total  # (written by the loss launches themselves: _LossFn)

    def forward(self, inputs, rgb_targets, feat_targets, depth_targets, schedule_mult):
        """Reference calling convention (losses.py:21): depth targets already computed by the caller."""
        return self._run(inputs, rgb_targets, feat_targets, schedule_mult, depth_direct=depth_targets)[0]

    def forward_with_prior(self, inputs, rgb_targets, feat_targets, inv_depths, depth_scale_rows, schedule_mult):
        """Fused form used by NeRFSystem.training_step: the depth-prior affine (nerf_system.py:169-177) is evaluated
        inside the kernel and its gradient reaches depth_scale through `depth_scale_rows`."""
        return self._run(inputs, rgb_targets, feat_targets, schedule_mult, inv_depth=inv_depths,
                         scale_rows=depth_scale_rows)

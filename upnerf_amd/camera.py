"""Pose refinement + ray generation on the GPU: se(3) exponential (utils/camera.py:87-98), pose composition
(camera.py:51-58) and get_rays (utils/ray.py:44-56) fused into one HIP kernel with an analytic backward, so the
per-image se(3) table stays differentiable through the renderer."""
from __future__ import annotations

import torch

from ._lib import check, lib, ptr, stream


class _PoseRays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, se3_rows, c2w, directions):
        R = directions.shape[0]
        dev = directions.device
        se3 = se3_rows.detach().contiguous() if se3_rows is not None else None
        c2w, directions = c2w.detach().contiguous(), directions.detach().contiguous()
        o = torch.empty(R, 3, device=dev, dtype=torch.float32)
        d = torch.empty(R, 3, device=dev, dtype=torch.float32)
        check(lib.upnerf_pose_rays_fwd(R, ptr(se3), ptr(c2w), ptr(directions), ptr(o), ptr(d), stream()),
              "upnerf_pose_rays_fwd")
        ctx.save_for_backward(se3, c2w, directions)
        return o, d

    @staticmethod
    def backward(ctx, g_o, g_d):
        se3, c2w, directions = ctx.saved_tensors
        if se3 is None or not ctx.needs_input_grad[0]:
            return None, None, None
        R = directions.shape[0]
        z = lambda g: torch.zeros(R, 3, device=directions.device) if g is None else g.contiguous()
        # keep the contiguous copies alive until the launch has been enqueued (a temporary would be returned to the
        # caching allocator, and re-used by the next copy, before the kernel reads it)
        g_o, g_d = z(g_o), z(g_d)
        g = torch.empty(R, 6, device=directions.device, dtype=torch.float32)
        check(lib.upnerf_pose_rays_bwd(R, ptr(se3), ptr(c2w), ptr(directions), ptr(g_o), ptr(g_d), ptr(g), stream()),
              "upnerf_pose_rays_bwd")
        return g, None, None


def refine_and_get_rays(se3_rows, c2w, directions):
    """rays_o, rays_d [R,3] for per-ray poses c2w [R,3,4] refined by se3_rows [R,6] (None = no refinement).
    Equivalent to get_rays(directions, compose([se3_to_SE3(se3_rows), c2w]))  (nerf_system.py:158-165)."""
    if c2w.dim() == 2:
        c2w = c2w[None].expand(directions.shape[0], 3, 4)
    return _PoseRays.apply(se3_rows, c2w, directions)


def get_rays(directions, c2w):
    """utils/ray.py:30-67 (both branches) without refinement."""
    return refine_and_get_rays(None, c2w, directions.reshape(-1, 3))

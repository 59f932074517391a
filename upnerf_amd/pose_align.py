"""Pose bookkeeping around the evaluation path: what eval.py:13-42 and NeRFSystemOptimize.model_setup
(models/nerf_system_optmize.py:267-317) do with the TRAINED per-image se(3) refinements before any ray is rendered.

* `refined_poses`      -- se3_to_SE3 + compose for every training image (eval.py:33-34), evaluated by the SAME HIP kernel
                          the training step uses (three basis-vector rays per image through upnerf_pose_rays_fwd), so the
                          evaluated poses are bit for bit the ones the fields were trained with.
* `pose_metric`        -- rotation / translation error after Sim(3) alignment to the ground truth (utils/metric.py:65-78).
* `init_test_poses`    -- the held-out images' ground-truth poses carried into the frame the model was trained in: the
                          initial poses of test-time optimisation (nerf_system_optmize.py:271-317).

Everything except `refined_poses` is host logic on [N,3,4] tensors (a 3x3 SVD in double): a few microseconds of
arithmetic next to a render, it stays in torch.  Conventions follow the reference: a pose is [R|t], `compose_pair(a, b)`
applies a first (utils/camera.py:51-58)."""
from __future__ import annotations

from typing import NamedTuple, Optional, Tuple

import torch
from torch import Tensor

__all__ = ["Sim3", "refined_poses", "to_eval_frame", "camera_centers", "procrustes", "prealign_cameras",
           "rotation_distance", "pose_metric", "init_test_poses"]


class Sim3(NamedTuple):
    t0: Tensor  # [3] centroid of the target point set
    t1: Tensor  # [3] centroid of the source point set
    s0: Tensor  # RMS radius of the target set
    s1: Tensor  # RMS radius of the source set
    R: Tensor   # [3,3] rotation; source -> target: (X1 - t1) / s1 @ R.T * s0 + t0


def _pose(R: Tensor, t: Tensor) -> Tensor:
    return torch.cat([R.float(), t.float()[..., None]], -1)


def _invert(p: Tensor) -> Tensor:  # utils/camera.py:35-41 (rotation inverse by transpose)
    Rt = p[..., :3].transpose(-1, -2)
    return _pose(Rt, (-Rt @ p[..., 3:])[..., 0])


def refined_poses(se3_weight: Tensor, poses: Tensor) -> Tensor:
    """[N,6] trained se(3) rows + [N,3,4] (noised) training poses -> [N,3,4] refined poses, through the HIP pose kernel."""
    from .camera import refine_and_get_rays
    if not se3_weight.is_cuda:
        raise RuntimeError("refined_poses runs the HIP pose kernel: pass device tensors")
    N = se3_weight.shape[0]
    basis = torch.eye(3, device=se3_weight.device).repeat(N, 1)  # rays e_x, e_y, e_z of every image
    rows = se3_weight.detach().float().repeat_interleave(3, 0).contiguous()
    c2w = poses.to(se3_weight.device).float().repeat_interleave(3, 0).contiguous()
    o, d = refine_and_get_rays(rows, c2w, basis)  # d = R e_j (unit length already), o = t
    R = d.view(N, 3, 3).transpose(1, 2)  # columns
    return torch.cat([R, o.view(N, 3, 3)[:, 0, :, None]], -1)


def to_eval_frame(pose_raw: Tensor) -> Tensor:
    """utils/metric.py:34-39: flip the y/z axes, invert, flip again.  Closed form: R' = F R^T F, t' = -F R^T t."""
    F = torch.diag(torch.tensor([1.0, -1.0, -1.0], device=pose_raw.device))
    p = pose_raw[..., :3, :].float()
    Rt = p[..., :3].transpose(-1, -2)
    return _pose(F @ Rt @ F, (-(F @ Rt) @ p[..., 3:])[..., 0])


def camera_centers(pose: Tensor) -> Tensor:
    """cam2world of the origin (utils/camera.py:282-285): -R^T t."""
    return _invert(pose)[..., 3]


def procrustes(X0: Tensor, X1: Tensor) -> Sim3:
    """Similarity transform that carries X1 onto X0 (utils/camera.py:364-382): centroids, RMS radii, rotation from the
    SVD (in double) of the normalised cross-covariance, reflection fixed by negating the last row."""
    t0, t1 = X0.mean(0), X1.mean(0)
    A, B = X0 - t0, X1 - t1
    s0, s1 = (A ** 2).sum(-1).mean().sqrt(), (B ** 2).sum(-1).mean().sqrt()
    U, _, Vh = torch.linalg.svd(((A / s0).t() @ (B / s1)).double(), full_matrices=False)
    R = (U @ Vh).float()
    if torch.linalg.det(R) < 0:
        R[2] = -R[2]
    return Sim3(t0, t1, s0, s1, R)


def prealign_cameras(pose: Tensor, pose_gt: Tensor) -> Tuple[Tensor, Sim3]:
    """utils/metric.py:42-52: align the predicted camera centres to the GT ones, carry the rotations along."""
    pose, pose_gt = pose.float(), pose_gt.float()
    c, c_gt = camera_centers(pose), camera_centers(pose_gt)
    s = procrustes(c_gt, c)
    c_al = (c - s.t1) / s.s1 @ s.R.t() * s.s0 + s.t0
    R_al = pose[..., :3] @ s.R.t()
    return _pose(R_al, (-R_al @ c_al[..., None])[..., 0]), s


def rotation_distance(R1: Tensor, R2: Tensor, eps: float = 1e-7) -> Tensor:
    d = R1 @ R2.transpose(-2, -1)  # utils/camera.py:354-361
    tr = d[..., 0, 0] + d[..., 1, 1] + d[..., 2, 2]
    return ((tr - 1) / 2).clamp(-1 + eps, 1 - eps).acos()


def pose_metric(refine_poses: Tensor, gt_poses: Tensor) -> Tuple[Optional[dict], Tensor, Tensor]:
    """eval.py:36-40 -> utils/metric.py:65-78.  Returns ({"R": [N] radians, "t": [N]}, aligned poses, GT poses), all in
    the evaluation frame; the error is None when the alignment fails (the reference prints and carries on)."""
    pr, gt = to_eval_frame(refine_poses.float().cpu()), to_eval_frame(gt_poses.float().cpu())
    try:
        al, _ = prealign_cameras(pr, gt)
    except RuntimeError:  # SVD did not converge
        return None, pr, gt
    err = {"R": rotation_distance(al[..., :3], gt[..., :3]), "t": (al[..., 3] - gt[..., 3]).norm(dim=-1)}
    return err, al, gt


def init_test_poses(train_refined: Tensor, gt_train_poses: Tensor, gt_test_poses: Tensor) -> Tensor:
    """Initial poses of the held-out images for test-time optimisation (nerf_system_optmize.py:279-317): Sim(3) between the
    trained frame and the GT frame from the TRAINING cameras, applied in the GT -> trained direction to the test cameras.

    train_refined: what the model was trained with for the training images.  The reference composes the trained se(3)
    with identity poses there (line 286: `noise_poses = eye`), i.e. pass refined_poses(se3_weight, eye(3,4) x N) to
    reproduce it."""
    pr = to_eval_frame(train_refined.float().cpu())
    gt = to_eval_frame(gt_train_poses.float().cpu())
    _, s = prealign_cameras(pr, gt)
    te = to_eval_frame(gt_test_poses.float().cpu())
    c = (camera_centers(te) - s.t0) / s.s0 @ s.R * s.s1 + s.t1
    R = te[..., :3] @ s.R
    return to_eval_frame(_pose(R, (-R @ c[..., None])[..., 0]))

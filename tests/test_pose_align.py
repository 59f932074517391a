"""Pose evaluation / test-time-optimisation pose initialisation (eval.py:28-40, nerf_system_optmize.py:279-317) against
vectors the reference's own camera functions produced (tools/make_goldens.py pose_align)."""
import os

import numpy as np
import pytest
import torch

from upnerf_amd import pose_align as pa

G = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_align.npz")).items()}


def close(a, b, tol=2e-6):
    assert a.shape == b.shape
    assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), float((a - b).abs().max())


def test_eval_frame_and_alignment_match_the_reference_vectors():
    close(pa.to_eval_frame(G["refined"]), G["eval_pred"])
    close(pa.to_eval_frame(G["gt_train"]), G["eval_gt"])
    al, s = pa.prealign_cameras(G["eval_pred"], G["eval_gt"])
    close(s.R, G["sim_R"]), close(s.t0, G["sim_t0"]), close(s.t1, G["sim_t1"])
    close(s.s0.reshape(()), G["sim_s0"].reshape(())), close(s.s1.reshape(()), G["sim_s1"].reshape(()))
    close(al, G["aligned"], 1e-5)


def test_pose_metric_reports_the_reference_errors():
    err, al, gt = pa.pose_metric(G["refined"], G["gt_train"])
    close(err["R"], G["R_err"], 2e-4)  # acos near 0: 1e-7 in the trace is 3e-4 rad at 0.05 rad
    close(err["t"], G["t_err"], 1e-5)
    # a similarity of the prediction is invisible to the metric
    R = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(2)))[0]
    R = R * torch.sign(torch.linalg.det(R))
    moved = torch.cat([R @ G["refined"][..., :3], 2.5 * (R @ G["refined"][..., 3:]) + torch.tensor([[0.3], [-1.0], [2.0]])], -1)
    err2, _, _ = pa.pose_metric(moved, G["gt_train"])
    close(err2["t"], err["t"], 1e-4), close(err2["R"], err["R"], 1e-3)


def test_procrustes_fixes_reflections_and_recovers_a_known_similarity():
    g = torch.Generator().manual_seed(0)
    X1 = torch.randn(20, 3, generator=g)
    R = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    R = R * torch.sign(torch.linalg.det(R))
    X0 = 3.0 * X1 @ R.t() + torch.tensor([1.0, -2.0, 0.5])
    s = pa.procrustes(X0, X1)
    close(s.R, R, 1e-5), close((s.s0 / s.s1).reshape(()), torch.tensor(3.0), 1e-5)
    close((X1 - s.t1) / s.s1 @ s.R.t() * s.s0 + s.t0, X0, 1e-5)
    flat = torch.cat([torch.randn(20, 2, generator=g), torch.zeros(20, 1)], 1)  # planar set: the SVD may return a reflection
    assert float(torch.linalg.det(pa.procrustes(flat, flat * torch.tensor([1.0, 1.0, -1.0])).R)) > 0


def test_initial_test_poses_match_the_reference_vectors():
    close(pa.init_test_poses(G["refined_identity"], G["gt_train"], G["gt_test"]), G["init_test"], 1e-5)


@pytest.mark.gpu
def test_refined_poses_come_from_the_training_kernel():
    out = pa.refined_poses(G["se3"].cuda(), G["noised"].cuda()).cpu()
    close(out, G["refined"], 2e-6)
    ident = torch.eye(3, 4).repeat(G["se3"].shape[0], 1, 1)
    close(pa.refined_poses(G["se3"].cuda(), ident.cuda()).cpu(), G["refined_identity"], 2e-6)
    with pytest.raises(RuntimeError):
        pa.refined_poses(G["se3"], G["noised"])

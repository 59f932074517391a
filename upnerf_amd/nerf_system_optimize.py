"""Test-time optimisation (TTO) on the HIP path: the callers of render_rays in models/nerf_system_optmize.py
(forward 84-111, training_step 113-150, validation_step 152-188) for ONE held-out image -- frozen NeRF weights, a
fresh appearance embedding and (pose stage) the image's se(3) refinement are optimised against the colour loss.

`eval_train_poses` and `tto_from_checkpoint` are the two things eval.py / tto.py do with a trained checkpoint before
rendering: the pose error of the training cameras after Sim(3) alignment (eval.py:13-42) and a TTO system whose fields
come from the checkpoint and whose held-out cameras start from the aligned ground truth
(nerf_system_optmize.py:254-317); the pose algebra is in pose_align.py.

Differences by design: the reference leaves the NeRF weights trainable-but-unused (it computes and discards all
weight gradients); here they are frozen, so the backward pass skips every weight-gradient
kernel.  SSIM/LPIPS and the pickle
bookkeeping are outside the accelerated path."""
from __future__ import annotations

import torch
from torch import nn

from . import zero_pool
from .camera import refine_and_get_rays
from .losses import _const
from .nerf_system import NeRFSystem
from .ops import EMBED_PREFETCH, embed_rows
from .optim import get_optimizer
from .rendering import join_rays, render_rays


class NeRFSystemOptimize(NeRFSystem):
    supports_graph_step = True  # the step splits like NeRFSystem's (graph_step.py): one graph replay per TTO step

    def __init__(self, hparams, train_dataset=None, val_dataset=None, pose_optimize=True):
        super().__init__(hparams, train_dataset, val_dataset)
        self.pose_optimize = pose_optimize
        # only s_rgb_fine is ever read here, so the coarse field stops at its density head (set False to get the
        # reference's full set of coarse maps back)
        self.coarse_sigma_only = True

    def model_setup(self, trained_state=None, n_test_images: int = 1):
        super().model_setup()
        if trained_state is not None:
            self.load_state_dict(trained_state, strict=False)
        hp = self.hparams
        # only trainable appearance table: one row per test image (nerf_system_optmize.py:254-256)
        self.embedding_fine_a = nn.Embedding(n_test_images, hp["nerf.appearance_dim"])
        self.embeddings["fine_a"] = self.embedding_fine_a
        self.se3_refine = nn.Embedding(n_test_images, 6)
        nn.init.zeros_(self.se3_refine.weight)
        for m in (self.nerf_coarse, self.nerf_fine):
            m.encode_candidate = False  # nerf_system_optmize.py:265-266
            for p in m.parameters():
                p.requires_grad_(False)
        for k in ("coarse_a", "coarse_c", "fine_c"):
            if k in self.embeddings:
                self.embeddings[k].weight.requires_grad_(False)
        self.set_progress(1.0)  # all encoding bands on, schedule finished

    def configure_optimizers(self):
        if self.pose_optimize:  # nerf_system_optmize.py:48-58
            opts = [get_optimizer("adam", 5e-3, [self.embedding_fine_a]), get_optimizer("adam", 1e-4, [self.se3_refine])]
        else:  # appearance stage: AdamW(1e-1) (lines 59-64); capturable: its step counter lives on the device, so the update
            # can sit inside the replayed graph (same update rule; on a CPU system the flag is not available)
            on_gpu = self.embedding_fine_a.weight.is_cuda
            opts = [torch.optim.AdamW(self.embedding_fine_a.parameters(), lr=1e-1, **({"capturable": True} if on_gpu else {}))]
        scheds = [{"scheduler": torch.optim.lr_scheduler.ConstantLR(o, factor=1.0, total_iters=0), "interval": "step"}
                  for o in opts]
        return opts, scheds

    def forward(self, rays, img_idx, train=True, u_list=None):
        hp = self.hparams
        chunk = rays.shape[0] if train else hp["val.chunk_size"]
        outs = []
        for i in range(0, rays.shape[0], chunk):
            outs.append(render_rays(models=self.models, embeddings=self.embeddings,
                                    rays=rays if chunk >= rays.shape[0] else rays[i:i + chunk],
                                    img_idx=img_idx[i:i + chunk], sched_mult=1.0, N_samples=hp["nerf.N_samples"],
                                    use_disp=hp["nerf.use_disp"], perturb=hp["nerf.perturb"] if train else 0,
                                    N_importance=hp["nerf.N_importance"], encode_feat=hp["nerf.feat_dim"] > 0, u_list=u_list,
                                    coarse_sigma_only=self.coarse_sigma_only))
        return {k: (outs[0][k] if len(outs) == 1 else torch.cat([o[k] for o in outs], 0)) for k in outs[0]}

    def rays_from_batch(self, batch):
        se3 = embed_rows(self.se3_refine, batch["img_idx"], defer_grad=True) if self.pose_optimize else None
        o, d = refine_and_get_rays(se3, batch["c2w"], batch["directions"])
        return join_rays(o, d, batch["ray_infos"])

    def compute_loss(self, batch, u_list=None):
        rays = self.rays_from_batch(batch)
        self._last_rays = rays  # kept for tests (gradient w.r.t. the rays), as NeRFSystem.compute_loss does
        res = self(rays, batch["img_idx"], u_list=u_list)
        loss = ((res["s_rgb_fine"] - batch["rgbs"]) ** 2).mean()  # nerf_system_optmize.py:129
        return loss, {"rgb": loss}, res

    # the three pieces of a step (NeRFSystem.training_step composes them; graph_step.GraphedTrainingStep captures the first two)
    def _step_backward(self, batch, u_list=None):
        with zero_pool.step(batch["img_idx"].device), EMBED_PREFETCH.scope(self._per_image_tables()):
            loss, loss_d, _ = self.compute_loss(batch, u_list=u_list)
            for o in self._opts_scheds()[0]:
                o.zero_grad()
            self.manual_backward(loss, gradient=_const(1.0, loss.device))
        return loss, loss_d

    def _step_host(self, loss, loss_d, done):
        opts, _ = self._opts_scheds()
        for o, runs in zip(opts, done):
            if runs is not None:
                o.step_host(runs)
        self.global_step += len(opts)

    def training_step(self, batch, batch_nb=0, u_list=None):
        loss, loss_d = self._step_backward(batch, u_list=u_list)
        self._step_host(loss, loss_d, self._step_update())
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_nb=0):
        """Full-image render in val.chunk_size chunks, perturb = 0; returns the PSNR on s_rgb_fine."""
        res = self(self.rays_from_batch(batch), batch["img_idx"], train=False)
        mse = ((res["s_rgb_fine"] - batch["rgbs"]) ** 2).mean()
        return {"val_psnr": -10.0 * torch.log10(mse), "s_rgb_fine": res["s_rgb_fine"], "s_depth_fine": res["s_depth_fine"]}

    def validation_epoch_end(self, outputs):
        """Mean PSNR over the validation images (nerf_system_optmize.py:190-196)."""
        if not outputs:
            return None
        out = {"val/psnr": torch.stack([x["val_psnr"].reshape(()) for x in outputs]).mean()}
        self.log("val/psnr", out["val/psnr"])
        return out


def run_stage(system: NeRFSystemOptimize, train_batches, n_batches_per_epoch: int, max_epochs: int, val_batches=(),
              graph: bool = True):
    """One test-time-optimisation stage the way tto.py:56-91 runs it: `max_epochs` passes over the held-out image's rays
    (50 for the pose stage, 20 for the appearance stage), a validation render after every epoch, no checkpoints.
    Returns the Trainer (its `history` holds val/psnr per epoch)."""
    from .trainer import Trainer
    opts = system.optimizers()
    n_opt = len(opts) if isinstance(opts, (list, tuple)) else 1
    budget = int(system.global_step) + max_epochs * n_batches_per_epoch * n_opt
    return Trainer(budget, val_check_interval=1.0, dirpath=None, graph=graph).fit(system, train_batches, n_batches_per_epoch,
                                                                                   val_batches)


def eval_train_poses(checkpoint, noised_poses, gt_poses, device="cuda") -> dict:
    """eval.py:13-42: trained se(3) refinements composed with the (noised) training poses, Sim(3)-aligned to the ground
    truth; mean rotation error in degrees and mean translation error, as eval.py prints them, plus the per-image values."""
    import math
    from .checkpoint import read_checkpoint
    from .pose_align import pose_metric, refined_poses
    se3 = read_checkpoint(checkpoint)["state_dict"]["se3_refine.weight"]
    refined = refined_poses(se3.to(device), noised_poses.to(device)).cpu()
    err, aligned, gt = pose_metric(refined, gt_poses)
    if err is None:
        return {"train/pose_R": None, "train/pose_t": None, "refined": refined, "aligned": aligned}
    return {"train/pose_R": float(err["R"].mean()) * 180.0 / math.pi, "train/pose_t": float(err["t"].mean()),
            "R": err["R"], "t": err["t"], "refined": refined, "aligned": aligned}


def tto_from_checkpoint(checkpoint, pose_optimize: bool, n_test_images: int = 1, gt_train_poses=None, gt_test_poses=None,
                        device="cuda", **overrides):
    """NeRFSystemOptimize for the held-out images of a trained run (nerf_system_optmize.py:254-317): hyper-parameters and
    fields from the checkpoint, a fresh appearance row per test image and, when ground-truth poses are given, their
    initial cameras in the frame the model was trained in.  Returns (system, initial test poses or None)."""
    from .checkpoint import read_checkpoint
    from .pose_align import init_test_poses, refined_poses
    ck = read_checkpoint(checkpoint)
    hp = dict(ck["hyper_parameters"])
    hp.update(overrides)
    sd = ck["state_dict"]
    n_train = sd["se3_refine.weight"].shape[0] if "se3_refine.weight" in sd else sd["embedding_fine_a.weight"].shape[0]
    from .nerf_system import SyntheticDataset
    system = NeRFSystemOptimize(hp, SyntheticDataset(n_train), pose_optimize=pose_optimize)
    keep = {k: v for k, v in sd.items() if not k.startswith(("embedding_fine_a.", "se3_refine."))}
    system.model_setup(trained_state=keep, n_test_images=n_test_images)
    system.to(device)
    init = None
    if gt_train_poses is not None and gt_test_poses is not None:
        ident = torch.eye(3, 4).repeat(n_train, 1, 1)  # line 286: the trained refinements over identity poses
        init = init_test_poses(refined_poses(sd["se3_refine.weight"].to(device), ident.to(device)).cpu(),
                               gt_train_poses, gt_test_poses)
    return system, init

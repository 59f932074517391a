"""NeRFSystem: host-side orchestration of one UP-NeRF training step with the reference's module surface
(models/nerf_system.py:22-461): same constructor (flat dotted hparams), attribute names, state_dict keys, hook
names and schedule semantics, so train.py / tto.py style drivers and Lightning checkpoints carry over.

pytorch_lightning is optional (it is not installed in this image): when importable, NeRFSystem subclasses
LightningModule; otherwise a minimal base provides the few members the hooks use (optimizers(), lr_schedulers(),
manual_backward, log, global_step) and `fit_steps()` drives training.

Differences by design: poses/rays, rendering, transient net run on the HIP path; the per-step host sync of the
reference (progress.item(), nerf_system.py:180) is replaced by a host-side step counter; with world_size > 1 the
gradient all-reduce is explicit (upnerf_amd.parallel.GradSync) instead of DDP hooks."""
from __future__ import annotations

import math
from collections import defaultdict

import torch
from torch import nn

from . import step_scalars, zero_pool
from .camera import refine_and_get_rays
from .losses import UPNeRFLoss, _const
from .nerf import NeRF, fp32_round
from .ops import EMBED_PREFETCH, embed_rows, reset_deferred
from .optim import get_learning_rate, get_optimizer, get_scheduler
from .parallel import GradSync
from .rendering import join_rays, render_rays
from .transient_net import TransientNet

try:  # pragma: no cover - Lightning is absent in the build image
    from pytorch_lightning import LightningModule as _Base
    _HAVE_PL = True
except Exception:
    _HAVE_PL = False

    class _Base(nn.Module):
        """The members of LightningModule that NeRFSystem's hooks touch."""

        def __init__(self):
            super().__init__()
            self.global_step = 0
            self._opts, self._scheds, self.logged = None, None, {}
            self.automatic_optimization = True

        def save_hyperparameters(self, hp):
            self.hparams = dict(hp)

        def log(self, name, value, **kw):
            self.logged[name] = value

        def _ensure_optim(self):
            if self._opts is None:
                opts, scheds = self.configure_optimizers()
                self._opts, self._scheds = list(opts), [s["scheduler"] for s in scheds]

        def optimizers(self):
            self._ensure_optim()
            return self._opts if len(self._opts) > 1 else self._opts[0]

        def lr_schedulers(self):
            self._ensure_optim()
            return self._scheds if len(self._scheds) > 1 else self._scheds[0]

        def manual_backward(self, loss, *args, **kwargs):
            loss.backward(*args, **kwargs)


class _LazyResults(dict):
    """The results dict of NeRFSystem.forward with entries that are computed on first access (`lazy(key, thunk)`): present
    for `in`, `keys()`, `get`, `[]`, `items()` like any other entry."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._thunks = {}

    def lazy(self, key, thunk):
        self._thunks[key] = thunk

    def _force(self, key=None):
        for k in ([key] if key is not None else list(self._thunks)):
            if k in self._thunks:
                super().__setitem__(k, self._thunks.pop(k)())

    def __contains__(self, key):
        return key in self._thunks or super().__contains__(key)

    def __getitem__(self, key):
        self._force(key)
        return super().__getitem__(key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __setitem__(self, key, value):
        self._thunks.pop(key, None)
        super().__setitem__(key, value)

    def __iter__(self):
        self._force()
        return super().__iter__()

    # the rest of the dict protocol sees every entry too (r3 ADVICE: pop / copy / update / setdefault / == / pickling bypassed
    # the thunks): all pending entries are evaluated first
    def pop(self, key, *default):
        self._force(key)
        return super().pop(key, *default)

    def setdefault(self, key, default=None):
        self._force(key)
        return super().setdefault(key, default)

    def copy(self):
        self._force()
        return dict(self)

    def update(self, *a, **k):
        for key, value in dict(*a, **k).items():
            self[key] = value  # (an explicit value replaces a pending thunk)

    def keys(self):
        self._force()
        return super().keys()

    def values(self):
        self._force()
        return super().values()

    def items(self):
        self._force()
        return super().items()

    def __len__(self):
        return dict.__len__(self) + sum(1 for k in self._thunks if not dict.__contains__(self, k))

    def __eq__(self, other):
        self._force()
        return dict(self) == other

    __hash__ = None

    def __reduce__(self):  # pickles (and torch.saves) as the plain dict it stands for
        self._force()
        return (dict, (dict(self),))



class NeRFSystem(_Base):
    supports_graph_step = True  # training_step splits into _step_backward / _step_update / _step_host (graph_step.py)

    def __init__(self, hparams, train_dataset=None, val_dataset=None):
        super().__init__()
        self.save_hyperparameters(hparams)
        self.automatic_optimization = False
        self.candidate_schedule = self.hparams["candidate_schedule"]
        self.fine = self.hparams["nerf.N_importance"] > 0
        self.loss = UPNeRFLoss(depth_mult=self.hparams["loss.depth_mult"], alpha_reg=self.hparams["loss.alpha_reg"],
                               encode_feat=self.hparams["nerf.feat_dim"] > 0, fine=self.fine,
                               near=self.hparams["nerf.near"], far=self.hparams["nerf.far"])
        self.val_log_N = 0
        self.train_dataset, self.val_dataset = train_dataset, val_dataset
        self._host_progress = 0.0
        self.grad_sync = None

    # ---- hooks -----------------------------------------------------------------------------------
    def setup(self, stage=None):
        if self.train_dataset is None:
            self.dataset_setup()
        self.model_setup()

    def dataset_setup(self):
        raise NotImplementedError(
            "dataset construction (datasets/phototourism.py) is outside the accelerated path; pass train_dataset / "
            "val_dataset objects (anything with N_images_train and white_back) to NeRFSystem(...)")

    def configure_optimizers(self):
        hp = self.hparams
        self.optimizer = get_optimizer(hp["optimizer.type"], hp["optimizer.lr"], self.models_to_train)
        sched = get_scheduler(hp["optimizer.scheduler.type"], hp["optimizer.lr"], hp["optimizer.scheduler.lr_end"],
                              hp["max_steps"], self.optimizer)
        optimizer, scheduler = [self.optimizer], [{"scheduler": sched, "interval": "step"}]
        if hp["pose.optimize"]:
            self.optimizer_pose = get_optimizer(hp["optimizer_pose.type"], hp["optimizer_pose.lr"],
                                                [self.depth_scale, self.se3_refine])
            sched_p = get_scheduler(hp["optimizer_pose.scheduler.type"], hp["optimizer_pose.lr"],
                                    hp["optimizer_pose.scheduler.lr_end"], hp["max_steps"], self.optimizer_pose)
            optimizer += [self.optimizer_pose]
            scheduler += [{"scheduler": sched_p, "interval": "step"}]
        return optimizer, scheduler

    def model_setup(self):
        hp = self.hparams
        N_images = self.train_dataset.N_images_train
        self.embeddings, self.models_to_train = {}, []
        for kind, dim in (("a", hp["nerf.appearance_dim"]), ("c", hp["nerf.candidate_dim"])):
            if dim > 0:
                for typ in ("coarse", "fine") if self.fine else ("coarse",):
                    emb = nn.Embedding(N_images, dim)
                    setattr(self, f"embedding_{typ}_{kind}", emb)
                    self.embeddings[f"{typ}_{kind}"] = emb
                    self.models_to_train += [emb]
        kw = dict(encode_feat=hp["nerf.feat_dim"] > 0, feat_dim=hp["nerf.feat_dim"], xyz_L=hp["nerf.N_emb_xyz"],
                  dir_L=hp["nerf.N_emb_dir"], appearance_dim=hp["nerf.appearance_dim"],
                  candidate_dim=hp["nerf.candidate_dim"], c2f=hp["pose.c2f"],
                  D=hp.get("nerf.D", 8), W=hp.get("nerf.W", 256))  # D/W: build extension
        self.nerf_coarse = NeRF("coarse", **kw)
        self.models = {"nerf_coarse": self.nerf_coarse}
        if self.fine:
            self.nerf_fine = NeRF("fine", **kw)
            self.models["nerf_fine"] = self.nerf_fine
        self.transient_net = TransientNet(N_images=N_images, beta_min=hp["t_net.beta_min"],
                                          trasient_dim=hp["t_net.transient_dim"], feat_dim=hp["t_net.feat_dim"])
        self.models["transient_network"] = self.transient_net
        self.models_to_train += [self.models]
        self.se3_refine = nn.Embedding(N_images, 6)
        nn.init.zeros_(self.se3_refine.weight)
        self.depth_scale = nn.Embedding(N_images, 2)
        nn.init.zeros_(self.depth_scale.weight)

    # ---- forward (nerf_system.py:93-148) -------------------------------------------------------------
    def forward(self, rays, feats, img_idx, sched_mult, train=True, u_list=None, keep=None, z_fine=None):
        hp = self.hparams
        rng = None
        if train and u_list is None and hp.get("rng.keyed", True) and hp["nerf.perturb"] > 0:
            # stratified-sampling draws keyed by (seed, optimisation step, GLOBAL row of the ray): the reference draws from the
            # process-wide generator (rendering.py:248, 29), whose numbers depend on the rank count
            row0, stride = self.rng_rows(rays.shape[0])
            rng = {"seed": int(hp.get("seed", 0)), "step": int(self.global_step), "row0": row0, "stride": stride}
        sched_phase = 0 if sched_mult == 0 else (2 if sched_mult == 1 else 1)
        B = rays.shape[0]
        results = defaultdict(list)
        chunk = B if train else hp["val.chunk_size"]
        # TransientNet (models/nerf_system.py:128-146) depends on the batch only, not on the rendering: with
        # hparams["hip.side_stream"] its ~20 small launches (and, through autograd's stream bookkeeping, their backward)
        # run on a side stream BESIDE the field kernels; the blend below waits for it.  Off by default: measured 19.89 vs
        # 20.0 ms per step (inside the noise), and a captured graph with a fork / join costs the host 9 ms per replay
        # instead of 0.3 (hipGraphLaunch walks the branches synchronously).
        t_side = None
        if (train and sched_mult > 0 and self.transient_net is not None and rays.is_cuda
                and hp.get("hip.side_stream", False)):
            cur = torch.cuda.current_stream(rays.device)
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream(rays.device)
            self._side_stream.wait_stream(cur)
            with torch.cuda.stream(self._side_stream):
                t_side = self.transient_net(feats, img_idx)
            for v in t_side.values():
                v.record_stream(cur)  # consumed by the blend / loss on the main stream
        for i in range(0, B, chunk):
            out = render_rays(models=self.models, embeddings=self.embeddings, rays=rays if chunk >= B else rays[i:i + chunk],
                              img_idx=img_idx[i:i + chunk], sched_mult=sched_mult, sched_phase=sched_phase,
                              N_samples=hp["nerf.N_samples"], use_disp=hp["nerf.use_disp"],
                              perturb=hp["nerf.perturb"] if train else 0, N_importance=hp["nerf.N_importance"],
                              white_back=getattr(self.train_dataset, "white_back", False),
                              encode_feat=hp["nerf.feat_dim"] > 0, validation=not train, u_list=u_list, keep=keep,
                              z_fine=None if z_fine is None else z_fine[i:i + chunk],
                              rng=None if rng is None else dict(rng, row0=rng["row0"] + i * rng["stride"]))
            for k, v in out.items():
                results[k] += [v]
        results = {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in results.items()}
        if sched_mult > 0:
            if self.transient_net is not None:
                if t_side is not None:
                    torch.cuda.current_stream(rays.device).wait_stream(self._side_stream)
                    t = t_side
                else:
                    t = self.transient_net(feats, img_idx)
                t_rgbs, t_alphas, t_betas = t["rgb"], t["alpha"], t["beta"]
                # the blended colours (models/nerf_system.py:136-142) have one reader, the validation PSNR (:268): they are
                # evaluated when somebody asks for them -- the training step never does (its loss reads s_rgb_*, losses.py:38,59)
                results = _LazyResults(results)
                s_c = results["s_rgb_coarse"]
                results.lazy("rgb_coarse", lambda: s_c * (1 - t_alphas.detach()) + t_rgbs.detach() * t_alphas.detach())
                if "s_rgb_fine" in results:
                    s_f = results["s_rgb_fine"]
                    results.lazy("rgb_fine", lambda: s_f * (1 - t_alphas) + t_rgbs * t_alphas)
                results["t_beta"], results["t_alpha"] = t_betas, t_alphas
            else:
                results["rgb_coarse"] = results["s_rgb_coarse"]
        return results

    def rng_rows(self, rows: int):
        """(row0, stride): global row of local ray r in the data-parallel batch = row0 + r * stride.  The ray sampler deals
        the global batch like DistributedSampler (ray_sampler.py: perm[rank::world]), so local ray r of rank k is global ray
        r * world + k -> (rank, world); `_rng_row0` / `_rng_stride` pin other layouts (tests: contiguous halves)."""
        if getattr(self, "_rng_row0", None) is not None:
            return int(self._rng_row0), int(getattr(self, "_rng_stride", 1))
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    # ---- pieces of training_step, exposed for tests and the benchmark ------------------------------------
    def rays_from_batch(self, batch):
        idx = batch["img_idx"]
        se3 = embed_rows(self.se3_refine, idx, defer_grad=True) if self.hparams["pose.optimize"] else None
        rays_o, rays_d = refine_and_get_rays(se3, batch["c2w"], batch["directions"])
        return join_rays(rays_o, rays_d, batch["ray_infos"])

    def depth_targets(self, batch):
        """Affine-corrected mono-depth prior (nerf_system.py:169-177)."""
        near, far = self.hparams["nerf.near"], self.hparams["nerf.far"]
        scale, shift = torch.unbind(embed_rows(self.depth_scale, batch["img_idx"]), 1)
        p = batch["inv_depths"] * torch.exp(scale) + shift
        p = torch.where(p < 1 / far, torch.full_like(p, 1 / far), p)
        d = 1.0 / p
        return torch.where(d < near, torch.full_like(d, near), d)

    def compute_loss(self, batch, u_list=None, keep=None, z_fine=None):
        reset_deferred()  # nothing pending from a backward pass that raised
        rays = self.rays_from_batch(batch)
        self._last_rays = rays  # kept for tests (gradient w.r.t. the rays)
        sched_mult = self.get_schedule_mult(self._host_progress)
        results = self(rays, batch["feats"], batch["img_idx"], sched_mult, u_list=u_list, keep=keep, z_fine=z_fine)
        loss_d, _depth = self.loss.forward_with_prior(results, batch["rgbs"], batch["feats"], batch["inv_depths"],
                                                      embed_rows(self.depth_scale, batch["img_idx"], defer_grad=True), sched_mult)
        return self.loss.total(), loss_d, results

    def set_progress(self, progress: float):
        """Host-side copy of NeRF.progress (avoids the reference's per-step .item() sync)."""
        self._host_progress = fp32_round(progress)  # the value `.item()` on the fp32 parameter would give
        self.nerf_coarse.set_progress(progress)
        if self.fine:
            self.nerf_fine.set_progress(progress)

    def _opts_scheds(self):
        opts, scheds = self.optimizers(), self.lr_schedulers()
        opts = opts if isinstance(opts, (list, tuple)) else [opts]
        scheds = scheds if isinstance(scheds, (list, tuple)) else [scheds]
        return list(opts), list(scheds)

    # training_step = three pieces, so that graph_step.GraphedTrainingStep can capture the two that only enqueue device work
    # (and put the gradient all-reduce between them) and replay them with the host bookkeeping done beside the replays
    def _step_backward(self, batch, u_list=None):
        """Forward, loss and backward; gradients are left in the parameters' .grad."""
        # the step's zeroed buffers are slices of one arena (one fill launch), its per-image table rows come from one gather launch
        with zero_pool.step(batch["img_idx"].device), EMBED_PREFETCH.scope(self._per_image_tables()):
            loss, loss_d, _ = self.compute_loss(batch, u_list=u_list)
            for o in self._opts_scheds()[0]:
                o.zero_grad()
            if self.grad_sync is not None and step_scalars.current() is None:
                # eager step: the fine field's gradients are all-reduced while the rest of backward runs.  (Under graph
                # capture the collective stays between the two graphs of a step, graph_step.py.)
                sm = self.get_schedule_mult(self._host_progress)
                self.grad_sync.begin(0 if sm == 0 else (2 if sm == 1 else 1))
            self.manual_backward(loss, gradient=_const(1.0, loss.device))  # (the default seed is a ones_like fill launch)
        return loss, loss_d

    def _per_image_tables(self):
        """nn.Embedding modules a training step gathers with batch["img_idx"] (models/nerf_system.py:79-91)."""
        t = [getattr(self, "se3_refine", None), getattr(self, "depth_scale", None)] + list(self.embeddings.values())
        tn = getattr(self, "transient_net", None)
        return t + [getattr(tn, "embedding_t", None)]

    def _step_update(self):
        """Both optimiser updates (device work only for FlatAdam; returns what _step_host needs to advance its counters)."""
        done = []
        for o in self._opts_scheds()[0]:
            if hasattr(o, "step_device"):
                done.append(o.step_device())
            else:
                o.step()
                done.append(None)
        return done

    def _step_host(self, loss, loss_d, done):
        """Host bookkeeping of one step: optimiser step counts, LR schedulers, global step, logs, progress."""
        hp = self.hparams
        opts, scheds = self._opts_scheds()
        for o, s, runs in zip(opts, scheds, done):
            if runs is not None:
                o.step_host(runs)
            s.step()
        if not _HAVE_PL:
            self.global_step += len(opts)  # Lightning 1.9 counts one global step per optimizer.step
        self.log("lr", get_learning_rate(opts[0]))
        self.log("train/loss", loss.detach())
        for k, v in loss_d.items():
            self.log(f"train/{k}", v.detach(), prog_bar=True)
        if hp["pose.optimize"]:  # progress only advances with pose optimisation
            self.set_progress(self.global_step / (hp["max_steps"] * 2))

    def training_step(self, batch, batch_nb=0, u_list=None):
        loss, loss_d = self._step_backward(batch, u_list=u_list)
        if self.grad_sync is not None:
            self.grad_sync()
        self._step_host(loss, loss_d, self._step_update())
        return loss

    # ---- validation (nerf_system.py:231-269 without the image logging; 318-324) -------------------------------------
    @torch.no_grad()
    def validation_step(self, batch, batch_nb=0):
        """One full validation image (DataLoader batch_size 1: every tensor has a leading 1): chunked render with
        perturb = 0 and no gradient (the field kernels then skip all stores the backward would need), the loss terms
        of the current phase and the PSNR of the blended colour, both on device."""
        b = {k: (v[0] if torch.is_tensor(v) else v) for k, v in batch.items()}
        if b["c2w"].dim() == 2:
            b["c2w"] = b["c2w"][None].expand(b["directions"].shape[0], 3, 4)
        rays = self.rays_from_batch(b)
        sched_mult = self.get_schedule_mult(self._host_progress)
        results = self(rays, b["feats"], b["img_idx"], sched_mult, train=False)
        row = embed_rows(self.depth_scale, b["img_idx"][:1])  # the reference uses the first ray's image (nerf_system.py:249)
        loss_d, _ = self.loss.forward_with_prior(results, b["rgbs"], b["feats"], b["inv_depths"],
                                                 row.expand(b["img_idx"].shape[0], 2).contiguous(), sched_mult)
        log = {"val_loss": sum(l for l in loss_d.values())}
        typ = "fine" if "rgb_fine" in results else "coarse"
        if f"rgb_{typ}" in results:
            log["val_psnr"] = -10.0 * torch.log10(((results[f"rgb_{typ}"] - b["rgbs"]) ** 2).mean())  # utils/metric.py:9-20
        else:
            log["val_psnr"] = torch.zeros(1, device=rays.device)
        log["results"] = results
        return log

    def validation_epoch_end(self, outputs):
        if not outputs:
            return None
        out = {"val/loss": torch.stack([x["val_loss"] for x in outputs]).mean(),
               "val/psnr": torch.stack([x["val_psnr"].reshape(()) for x in outputs]).mean()}
        for k, v in out.items():
            self.log(k, v)
        return out

    def enable_data_parallel(self, check=False, overlap=True):
        """Average gradients over torch.distributed ranks after every backward.  overlap: the fine field's gradients (the
        first the backward pass completes) are all-reduced on a side stream while the rest of the backward pass runs
        (parallel.GradSync); the remaining gradients follow in one flat all-reduce at the end."""
        early = list(self.nerf_fine.parameters()) if (overlap and self.fine) else []
        self.grad_sync = GradSync([p for p in self.parameters()], check=check, early=early)

    def get_schedule_mult(self, progress):
        s, e = self.candidate_schedule
        if progress < s:
            return 0
        if progress > e:
            return 1
        return (1 - math.cos(math.pi * (progress - s) / (e - s))) / 2


class SyntheticDataset:
    """Stand-in for datasets/phototourism.py in benchmarks and tests: only the members the system reads."""

    def __init__(self, N_images_train: int, white_back: bool = False):
        self.N_images_train, self.white_back = N_images_train, white_back


def default_hparams(**over):
    """configs/default.yaml as the flat dotted dict configs/config.py:12-31 produces."""
    hp = {"seed": 42, "num_gpus": 1, "debug": False, "nerf.N_samples": 128, "nerf.N_importance": 128,
          "nerf.N_emb_xyz": 10, "nerf.N_emb_dir": 4, "nerf.near": 0.1, "nerf.far": 5.0, "nerf.appearance_dim": 48,
          "nerf.candidate_dim": 16, "nerf.feat_dim": 384, "nerf.use_disp": False, "nerf.perturb": 1.0,
          "t_net.beta_min": 0.1, "t_net.transient_dim": 128, "t_net.feat_dim": 384, "loss.depth_mult": 1e-3,
          "loss.alpha_reg": 1.0, "optimizer.type": "adam", "optimizer.lr": 5e-4,
          "optimizer.scheduler.type": "ExponentialLR", "optimizer.scheduler.lr_end": 5e-5,
          "optimizer_pose.type": "adam", "optimizer_pose.lr": 2e-3, "optimizer_pose.scheduler.type": "ExponentialLR",
          "optimizer_pose.scheduler.lr_end": 1e-5, "max_steps": 600000, "train.batch_size": 2048,
          "val.chunk_size": 4096, "pose.optimize": True, "pose.c2f": (0.1, 0.5), "pose.noise": -1,
          "candidate_schedule": (0.1, 0.5)}
    hp.update(over)
    return hp

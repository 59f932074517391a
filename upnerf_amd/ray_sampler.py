"""GPU-resident train-split ray sampler: replaces the reference's per-ray Python
`PhototourismDataset.__getitem__` + DataLoader workers (datasets/phototourism.py:420-454, models/nerf_system.py:75-82).

All per-ray buffers (`all_ray_infos`, `all_directions`, `all_rgbs`, `all_pxl_coords`, `all_inv_depths`), the per-image
feature maps and the poses live in HBM (Brandenburg Gate at img_downscale 2: ~30 M rays x 48 B + 763 x 64 x 64 x 384 x 4 B
= 1.4 GB + 4.8 GB, a fraction of the 288 GB); one HIP launch (`upnerf_gather_rays`) produces a batch with exactly the keys,
shapes and values the reference's collated batch has -- including the bilinear feature interpolation and its quirk on
the last row / column.  The shuffle is a device-side `torch.randperm` per epoch (the reference: DataLoader(shuffle=True)).

With world_size > 1 every rank draws the same permutation (same seed), pads it by wrap-around to a multiple of the
world size and takes every world_size-th index, exactly like the DistributedSampler the reference gets from Lightning
(train.py:70-72): all ranks run the same number of batches per epoch."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterator, Optional

import torch

from ._lib import GatherRaysArgs, check, lib, ptr, stream


def epoch_indices(N: int, seed: int, epoch: int, rank: int = 0, world_size: int = 1, device="cuda") -> torch.Tensor:
    """Ray indices rank `rank` trains on in epoch `epoch`, in order: a permutation of range(N) seeded by (seed + epoch)
    -- the same on every rank -- padded by wrap-around to ceil(N / world) * world entries and strided by the world size
    (torch.utils.data.DistributedSampler's rule), so every rank gets exactly ceil(N / world) indices."""
    g = torch.Generator(device=device)
    g.manual_seed(seed + epoch)
    perm = torch.randperm(N, device=device, generator=g)
    if world_size > 1:
        pad = -(-N // world_size) * world_size - N
        if pad:
            perm = torch.cat([perm, perm[:pad]])
        perm = perm[rank::world_size].contiguous()
    return perm


class GpuRaySampler:
    def __init__(self, all_ray_infos, all_directions, all_rgbs, poses, all_pxl_coords=None, feat_maps=None,
                 all_inv_depths=None, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("GpuRaySampler keeps its buffers in HBM (no CPU path)")
        f = lambda t: None if t is None else torch.as_tensor(t, dtype=torch.float32).to(dev).contiguous()
        self.ray_infos, self.directions, self.rgbs = f(all_ray_infos), f(all_directions), f(all_rgbs)
        self.poses, self.pxl, self.feat_maps, self.inv_depths = f(poses), f(all_pxl_coords), f(feat_maps), f(all_inv_depths)
        self.N = self.ray_infos.shape[0]
        if self.ray_infos.shape != (self.N, 3) or self.directions.shape != (self.N, 3) or self.rgbs.shape != (self.N, 3):
            raise ValueError("all_ray_infos / all_directions / all_rgbs must be [N,3]")
        if self.poses.dim() != 3 or tuple(self.poses.shape[1:]) != (3, 4):
            raise ValueError("poses must be [N_images,3,4]")
        if self.feat_maps is not None:
            if self.feat_maps.dim() != 4 or self.feat_maps.shape[1] != self.feat_maps.shape[2]:
                raise ValueError("feat_maps must be [N_images,h,h,C] (the reference asserts h == w)")
            if self.pxl is None or tuple(self.pxl.shape) != (self.N, 2):
                raise ValueError("all_pxl_coords [N,2] is required with feat_maps")
        self.device = dev

    @classmethod
    def from_dataset(cls, ds, device="cuda") -> "GpuRaySampler":
        """From a train-split dataset object with the reference's buffer attributes (datasets/phototourism.py:213-323:
        all_ray_infos, all_directions, all_rgbs, all_pxl_coords, all_inv_depths, feat_maps; the per-image poses are
        `poses_dict[img_ids_train[i]]`, or a ready `poses` array of N_images_train rows)."""
        import numpy as np
        if hasattr(ds, "poses_dict") and hasattr(ds, "img_ids_train"):
            poses = np.stack([np.asarray(ds.poses_dict[i], dtype=np.float32) for i in ds.img_ids_train])
        else:
            poses = ds.poses
        g = lambda name: getattr(ds, name, None)
        return cls(ds.all_ray_infos, ds.all_directions, ds.all_rgbs, poses, all_pxl_coords=g("all_pxl_coords"),
                   feat_maps=g("feat_maps"), all_inv_depths=g("all_inv_depths"), device=device)

    def __len__(self) -> int:
        return self.N

    def sample(self, idx: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Batch for the ray indices `idx` (int64, on the device): the reference's collated `__getitem__` results."""
        idx = idx.to(self.device, torch.int64).contiguous()
        R = idx.numel()
        dev = self.device
        e = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        out = {"ray_infos": e(R, 2), "directions": e(R, 3), "img_idx": torch.empty(R, device=dev, dtype=torch.int64),
               "c2w": e(R, 3, 4), "rgbs": e(R, 3)}
        h = Cc = 0
        if self.feat_maps is not None:
            h, Cc = self.feat_maps.shape[1], self.feat_maps.shape[3]
            out["feats"] = e(R, Cc)
            if self.inv_depths is not None:  # the reference reads inv_depths inside the feature branch (phototourism.py:452)
                out["inv_depths"] = e(R)
        a = GatherRaysArgs(R=R, h=h, C=Cc, idx=ptr(idx), all_ray_infos=ptr(self.ray_infos),
                           all_directions=ptr(self.directions), all_rgbs=ptr(self.rgbs), all_pxl_coords=ptr(self.pxl),
                           all_inv_depths=ptr(self.inv_depths), feat_maps=ptr(self.feat_maps), poses=ptr(self.poses),
                           ray_infos=ptr(out["ray_infos"]), directions=ptr(out["directions"]), img_idx=ptr(out["img_idx"]),
                           c2w=ptr(out["c2w"]), rgbs=ptr(out["rgbs"]), feats=ptr(out.get("feats")),
                           inv_depths=ptr(out.get("inv_depths")))
        check(lib.upnerf_gather_rays(C.byref(a), stream()), "upnerf_gather_rays")
        return out

    def n_batches(self, batch_size: int, world_size: int = 1, drop_last: bool = False) -> int:
        """Batches per epoch, identical on every rank."""
        if world_size == 1:
            return self.N // batch_size if drop_last else -(-self.N // batch_size)
        per_rank = -(-self.N // world_size)  # DistributedSampler: ceil(N / world) samples per rank after padding
        return per_rank // batch_size if drop_last else -(-per_rank // batch_size)

    def batches(self, batch_size: int, seed: int = 0, epoch: int = 0, rank: int = 0, world_size: int = 1,
                drop_last: bool = False, start: int = 0) -> Iterator[Dict[str, torch.Tensor]]:
        """One shuffled epoch of batches of `batch_size` rays PER RANK (Lightning semantics: batch_size is per rank).
        `start` = number of leading batches to leave out (a run resumed inside the epoch: same permutation, nothing
        gathered for the batches already consumed).

        world_size > 1 follows torch.utils.data.DistributedSampler (what Lightning wraps the reference's DataLoader in,
        train.py:70-72): the permutation is padded by wrapping around to ceil(N / world) * world indices and rank k
        takes indices k, k + world, k + 2 world, ... of it, so EVERY rank sees the same number of rays and therefore
        the same number of batches per epoch (the last batch may be short, on all ranks alike)."""
        perm = epoch_indices(self.N, seed, epoch, rank, world_size, self.device)
        n = perm.numel()
        for lo in range(start * batch_size, n, batch_size):
            sl = perm[lo: lo + batch_size]
            if drop_last and sl.numel() < batch_size:
                return
            yield self.sample(sl)

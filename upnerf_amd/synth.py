"""Deterministic synthetic weights and ray batches (no dataset, no checkpoint, no RNG library).

Every tensor is a closed-form function of (name, shape, seed): a splitmix64 hash of the flat element
index, mapped to uniform [-1,1) and scaled like nn.Linear's default init (bound 1/sqrt(fan_in)).
Integer arithmetic only, so the same bytes come out on every machine; the golden fixtures under
tests/golden/ therefore do not need to store weights (SURVEY.md section 7 step 1, 8d "synthetic inputs").

State-dict key names follow the reference so tensors interchange with its modules:
  NeRF            models/nerf.py:39-78        (xyz_encoding_{i}.0, xyz_encoding_final, share_sigma.0, ...)
  TransientNet    models/transient_net.py:11-25
  per-image tables models/nerf_system.py:340-368, 406-409
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform(name: str, shape: Tuple[int, ...], seed: int = 0, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """float32 tensor, uniform in [lo,hi), keyed by (name, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((zlib.crc32(name.encode()) << 20) ^ (seed * 0x51ED27))
    with np.errstate(over="ignore"):
        h = _splitmix64(np.arange(n, dtype=np.uint64) + base)
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24 random bits -> [0,1)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def _linear(prefix: str, out_f: int, in_f: int, seed: int, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    b = gain / float(np.sqrt(in_f))
    return {prefix + ".weight": uniform(prefix + ".weight", (out_f, in_f), seed) * b,
            prefix + ".bias": uniform(prefix + ".bias", (out_f,), seed) * b}


def nerf_state(typ: str, D: int = 8, W: int = 256, skips=(4,), feat_dim: int = 384, xyz_L: int = 10, dir_L: int = 4,
               appearance_dim: int = 48, candidate_dim: int = 16, seed: int = 0, progress: float = 0.0,
               sigma_bias: float = 0.0, sigma_gain: float = 1.0, trunk_gain: float = 1.0,
               encode_feat: bool = True) -> Dict[str, torch.Tensor]:
    """state_dict of one reference NeRF (nerf.py:39-78).  `sigma_bias` shifts the density heads so that
    alphas are not all tiny (useful to exercise the compositing); `sigma_gain` scales their weights and `trunk_gain` the
    trunk weights ("trained-like" statistics: densities from 0 to tens, saturated alphas, activations over several decades)."""
    in_xyz, in_dir = 6 * xyz_L + 3, 6 * dir_L + 3
    sd: Dict[str, torch.Tensor] = {"progress": torch.tensor(float(progress))}
    pre = f"nerf_{typ}."
    for i in range(D):
        k = in_xyz if i == 0 else (W + in_xyz if i in skips else W)
        sd.update(_linear(pre + f"xyz_encoding_{i + 1}.0", W, k, seed, gain=trunk_gain))
    sd.update(_linear(pre + "xyz_encoding_final", W, W, seed))
    sd.update(_linear(pre + "share_sigma.0", 1, W, seed))
    if encode_feat:
        sd.update(_linear(pre + "feat_share_layer", feat_dim, W, seed))
    sd.update(_linear(pre + "rgb_share_layer.0", W // 2, (feat_dim if encode_feat else W) + in_dir + appearance_dim, seed))
    sd.update(_linear(pre + "rgb_share_layer.2", 3, W // 2, seed))
    if candidate_dim > 0:
        sd.update(_linear(pre + "candidate_encoding.0", W // 2, W + candidate_dim, seed))
        sd.update(_linear(pre + "candidate_encoding.2", W // 2, W // 2, seed))
        sd.update(_linear(pre + "candidate_sigma.0", 1, W // 2, seed))
        if encode_feat:
            sd.update(_linear(pre + "feat_candidate_layer", feat_dim, W // 2, seed))
        else:  # nerf.py:77-78
            sd.update(_linear(pre + "rgb_candidate_layer", 3, W // 2, seed))
    out = {k[len(pre):] if k.startswith(pre) else k: v for k, v in sd.items()}
    out["share_sigma.0.weight"] = out["share_sigma.0.weight"] * sigma_gain
    out["share_sigma.0.bias"] = out["share_sigma.0.bias"] + sigma_bias
    if candidate_dim > 0:
        out["candidate_sigma.0.weight"] = out["candidate_sigma.0.weight"] * sigma_gain
        out["candidate_sigma.0.bias"] = out["candidate_sigma.0.bias"] + sigma_bias
    return out


def transient_state(n_images: int, transient_dim: int = 128, feat_dim: int = 384, seed: int = 0) -> Dict[str, torch.Tensor]:
    """state_dict of the reference TransientNet (transient_net.py:11-25)."""
    pre = "tnet."
    sd = {pre + "embedding_t.weight": uniform(pre + "embedding_t.weight", (n_images, transient_dim), seed)}
    dims = [feat_dim, 256, 256, 256, 256]
    for j, i in enumerate((0, 2, 4, 6)):
        sd.update(_linear(pre + f"feat_encoder.{i}", dims[j + 1], dims[j], seed))
    sd.update(_linear(pre + "final_encoder", 256, 256, seed))
    sd.update(_linear(pre + "t_encoder.0", 128, 256 + transient_dim, seed))
    sd.update(_linear(pre + "alpha_layer.0", 1, 256, seed))
    sd.update(_linear(pre + "beta_layer.0", 1, 128, seed))
    sd.update(_linear(pre + "rgb_layer.0", 3, 128, seed))
    return {k[len(pre):]: v for k, v in sd.items()}


def tables(n_images: int, appearance_dim: int = 48, candidate_dim: int = 16, seed: int = 0, fine: bool = True,
           se3_scale: float = 1e-2, depth_scale: float = 1e-1) -> Dict[str, torch.Tensor]:
    """Per-image embedding tables (nerf_system.py:340-368) + pose/depth tables (406-409; zero-initialised in the
    reference, given small non-zero values here so that every Taylor term and gradient path is exercised)."""
    t = {}
    for typ in ("coarse", "fine") if fine else ("coarse",):
        t[f"embedding_{typ}_a"] = uniform(f"embedding_{typ}_a", (n_images, appearance_dim), seed)
        t[f"embedding_{typ}_c"] = uniform(f"embedding_{typ}_c", (n_images, candidate_dim), seed)
    t["se3_refine"] = uniform("se3_refine", (n_images, 6), seed) * se3_scale
    t["depth_scale"] = uniform("depth_scale", (n_images, 2), seed) * depth_scale
    return t


def batch(R: int, n_images: int, seed: int = 1, near: float = 0.1, far: float = 5.0, feat_dim: int = 384,
          identity_c2w: bool = True) -> Dict[str, torch.Tensor]:
    """One training batch with the dataset's keys and layouts (datasets/phototourism.py:421-454)."""
    dirs = uniform("directions", (R, 3), seed) * 0.6
    dirs[:, 2] = -1.0  # camera looks down -z (utils/ray.py:23-25)
    idx = (uniform("img_idx", (R,), seed, 0.0, 1.0) * n_images).long().clamp_(max=n_images - 1)
    if identity_c2w:  # pose.noise == -1 (datasets/phototourism.py:198-202)
        c2w = torch.eye(3, 4).repeat(R, 1, 1)
    else:
        from math import cos, sin
        ang = uniform("c2w_ang", (R,), seed) * 0.3
        c2w = torch.zeros(R, 3, 4)
        c2w[:, 0, 0] = torch.cos(ang); c2w[:, 0, 2] = torch.sin(ang); c2w[:, 1, 1] = 1
        c2w[:, 2, 0] = -torch.sin(ang); c2w[:, 2, 2] = torch.cos(ang)
        c2w[:, :, 3] = uniform("c2w_t", (R, 3), seed) * 0.2
    feats = uniform("feats", (R, feat_dim), seed)
    feats = feats / feats.norm(dim=-1, keepdim=True)  # unit-norm DINO descriptors (phototourism.py:287)
    depth = uniform("depth", (R,), seed, 0.5, 4.5)
    return {
        "ray_infos": torch.tensor([near, far]).repeat(R, 1),
        "directions": dirs,
        "c2w": c2w,
        "feats": feats,
        "img_idx": idx,
        "rgbs": uniform("rgbs", (R, 3), seed, 0.0, 1.0),
        "inv_depths": 1.0 / depth,
    }

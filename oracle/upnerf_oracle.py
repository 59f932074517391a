"""CPU oracle for the UP-NeRF render_rays training path.

TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only
tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it.
The shipped path (upnerf_amd/) never imports, calls or falls back to anything here.

What it is: a functional, dependency-free (torch CPU only) restatement of the
arithmetic the reference runs for one training step of the hot path
   se(3) refine -> compose -> get_rays -> render_rays(coarse, resample, fine)
   -> TransientNet -> UPNeRFLoss
written from the behaviour of the reference, with every function citing the
reference lines it follows (paths relative to /root/reference).  It is written on
plain dicts of tensors (keys = the reference's state_dict names) so the same code
checks both the reference's modules and the HIP-backed modules of upnerf_amd.

Parity pinning: tests/golden/*.npz were produced by tools/make_goldens.py, which
imports the real reference in the build container and dumps inputs/outputs/grads;
tests/test_oracle_golden.py checks this file against every one of them.

Floating point: everything is torch.float32 unless the caller passes float64
tensors (used by tests to measure the fp32 sensitivity of the maths itself).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# a2: SE(3) exponential  (utils/camera.py:87-98 se3_to_SE3, 113-124 skew, 126-152 taylor_A/B/C)
# --------------------------------------------------------------------------------------
def _series(theta: Tensor, first_den_factors, nth: int = 10) -> Tensor:
    """sum_{i=0}^{nth} (-1)^i theta^(2i) / den_i with den_i built multiplicatively like the
    reference's loops (camera.py:126-152).  first_den_factors(i) returns the factor den picks up
    at step i (1.0 when nothing is multiplied)."""
    acc = torch.zeros_like(theta)
    den = 1.0
    for i in range(nth + 1):
        den = den * first_den_factors(i)
        acc = acc + ((-1) ** i) * theta ** (2 * i) / den
    return acc


def taylor_A(theta: Tensor) -> Tensor:  # sin(x)/x, camera.py:126-134
    return _series(theta, lambda i: 1.0 if i == 0 else (2 * i) * (2 * i + 1))


def taylor_B(theta: Tensor) -> Tensor:  # (1-cos x)/x^2, camera.py:136-143
    return _series(theta, lambda i: (2 * i + 1) * (2 * i + 2))


def taylor_C(theta: Tensor) -> Tensor:  # (x-sin x)/x^3, camera.py:145-152
    return _series(theta, lambda i: (2 * i + 2) * (2 * i + 3))


def hat(w: Tensor) -> Tensor:
    """[...,3] -> [...,3,3] cross-product matrix (camera.py:113-124)."""
    a, b, c = w[..., 0], w[..., 1], w[..., 2]
    z = torch.zeros_like(a)
    rows = [torch.stack(r, -1) for r in ((z, -c, b), (c, z, -a), (-b, a, z))]
    return torch.stack(rows, -2)


def se3_exp(wu: Tensor) -> Tensor:
    """[...,6] (w,u) -> [...,3,4] = [R | V u]  (camera.py:87-98)."""
    w, u = wu[..., :3], wu[..., 3:]
    K = hat(w)
    th = w.norm(dim=-1)[..., None, None]
    eye = torch.eye(3, dtype=wu.dtype, device=wu.device)
    A, B, C = taylor_A(th), taylor_B(th), taylor_C(th)
    K2 = K @ K
    R = eye + A * K + B * K2
    V = eye + B * K + C * K2
    return torch.cat([R, V @ u[..., None]], -1)


# a3: pose composition, pose_new(x) = pose_b(pose_a(x))  (camera.py:43-58)
def compose_pair(pose_a: Tensor, pose_b: Tensor) -> Tensor:
    Ra, ta = pose_a[..., :3], pose_a[..., 3:]
    Rb, tb = pose_b[..., :3], pose_b[..., 3:]
    return torch.cat([Rb @ Ra, Rb @ ta + tb], -1)


# a4: get_rays, per-ray pose branch and single-pose branch  (utils/ray.py:44-65)
def get_rays(directions: Tensor, c2w: Tensor):
    if c2w.dim() == 3 and directions.dim() == 2 and c2w.shape[0] == directions.shape[0]:
        d = torch.matmul(directions[:, None, :], c2w[:, :, :3].transpose(1, 2))[:, 0, :]
        d = d / d.norm(dim=-1, keepdim=True)
        o = c2w[..., 3]
    else:
        d = directions @ c2w[:, :3].T
        d = d / d.norm(dim=-1, keepdim=True)
        o = c2w[:, 3].expand(d.shape)
    return o.reshape(-1, 3), d.reshape(-1, 3)


# --------------------------------------------------------------------------------------
# a6: BARF-masked positional encoding  (models/nerf.py:126-147)
# --------------------------------------------------------------------------------------
def band_weights(L: int, progress: float, c2f, dtype=torch.float32) -> Tensor:
    """Per-band weights w_k (nerf.py:137-143); all ones when c2f is None."""
    if c2f is None:
        return torch.ones(L, dtype=dtype)
    start, end = c2f
    prog = torch.tensor(float(progress), dtype=dtype)
    alpha = (prog - start) / (end - start) * L
    k = torch.arange(L, dtype=dtype)
    return (1 - ((alpha - k).clamp(min=0, max=1) * torch.pi).cos()) / 2


def posenc(x: Tensor, L: int, progress: float, c2f) -> Tensor:
    """[M,3] -> [M,3+6L], layout [x, (sin k0..kL-1, cos k0..kL-1) per coordinate]."""
    freq = (2 ** torch.arange(L, dtype=torch.float32)).to(x.dtype) * torch.pi
    arg = x[..., None] * freq  # [M,3,L]
    enc = torch.stack([arg.sin(), arg.cos()], dim=-2)  # [M,3,2,L]
    if c2f is not None:
        enc = enc * band_weights(L, progress, c2f, x.dtype)
    return torch.cat([x, enc.reshape(*x.shape[:-1], -1)], -1)


# --------------------------------------------------------------------------------------
# a7-a9: NeRF field  (models/nerf.py:80-124; layers built 39-78)
# --------------------------------------------------------------------------------------
class NerfCfg:
    """Static description of one NeRF (defaults = nerf.py:6-19 as called from nerf_system.py:371-392)."""

    def __init__(self, typ="coarse", D=8, W=256, skips=(4,), feat_dim=384, xyz_L=10, dir_L=4,
                 appearance_dim=48, candidate_dim=16, c2f=None, encode_candidate=None):
        self.typ, self.D, self.W, self.skips = typ, D, W, tuple(skips)
        self.feat_dim, self.xyz_L, self.dir_L = feat_dim, xyz_L, dir_L
        self.appearance_dim, self.candidate_dim, self.c2f = appearance_dim, candidate_dim, c2f
        self.encode_feat = feat_dim > 0
        self.encode_appearance = appearance_dim > 0
        self.encode_candidate = (candidate_dim > 0) if encode_candidate is None else encode_candidate


def _lin(p: Params, name: str, x: Tensor) -> Tensor:
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def nerf_field(p: Params, cfg: NerfCfg, xyz: Tensor, view_dir: Tensor, a: Optional[Tensor],
               c: Optional[Tensor], sched_mult: float, progress: float) -> Dict[str, Tensor]:
    """Per-sample field evaluation; returns s_sigma[M,1], s_feat, (c_sigma, c_feat), (s_rgb)."""
    x0 = posenc(xyz, cfg.xyz_L, progress, cfg.c2f)
    h = x0
    for i in range(cfg.D):  # nerf.py:84-87
        if i in cfg.skips:
            h = torch.cat([x0, h], 1)
        h = torch.relu(_lin(p, f"xyz_encoding_{i + 1}.0", h))
    out = {"s_sigma": F.softplus(_lin(p, "share_sigma.0", h))}  # nerf.py:89
    e = _lin(p, "xyz_encoding_final", h)  # nerf.py:93
    if not cfg.encode_feat:  # nerf.py:110-123: colour head on e in every phase, candidate head whenever sched_mult < 1
        parts = [e, posenc(view_dir, cfg.dir_L, progress, cfg.c2f)]
        if cfg.encode_appearance:
            parts.append(a)
        r = torch.relu(_lin(p, "rgb_share_layer.0", torch.cat(parts, 1)))
        out["s_rgb"] = torch.sigmoid(_lin(p, "rgb_share_layer.2", r))
        if sched_mult < 1:
            g = torch.relu(_lin(p, "candidate_encoding.0", torch.cat([e, c], 1)))
            g = torch.relu(_lin(p, "candidate_encoding.2", g))
            out["c_sigma"] = F.softplus(_lin(p, "candidate_sigma.0", g))
            out["c_rgb"] = _lin(p, "rgb_candidate_layer", g)
        return out
    out["s_feat"] = _lin(p, "feat_share_layer", e)  # nerf.py:95
    if sched_mult < 1 and cfg.encode_candidate:  # nerf.py:96-100
        g = torch.relu(_lin(p, "candidate_encoding.0", torch.cat([e, c], 1)))
        g = torch.relu(_lin(p, "candidate_encoding.2", g))
        out["c_sigma"] = F.softplus(_lin(p, "candidate_sigma.0", g))
        out["c_feat"] = _lin(p, "feat_candidate_layer", g)
    if sched_mult > 0:  # nerf.py:101-109
        parts = [out["s_feat"], posenc(view_dir, cfg.dir_L, progress, cfg.c2f)]
        if cfg.encode_appearance:
            parts.append(a)
        r = torch.relu(_lin(p, "rgb_share_layer.0", torch.cat(parts, 1)))
        out["s_rgb"] = torch.sigmoid(_lin(p, "rgb_share_layer.2", r))
    return out


# --------------------------------------------------------------------------------------
# a10: alpha compositing  (models/rendering.py:125-218)
# --------------------------------------------------------------------------------------
def _excl_cumprod(one_minus_alpha: Tensor) -> Tensor:
    """T_i = prod_{j<i} (1-alpha_j)  (rendering.py:135-141 et al.)."""
    lead = torch.ones_like(one_minus_alpha[:, :1])
    return torch.cumprod(torch.cat([lead, one_minus_alpha], -1)[:, :-1], -1)


def composite(res: Dict[str, Tensor], typ: str, f: Dict[str, Tensor], z: Tensor, sched_mult: float,
              encode_candidate: bool, encode_feat: bool = True) -> None:
    """Fills res[...] in place with the keys of SURVEY 8a 'outputs by phase'."""
    delta = torch.cat([z[:, 1:] - z[:, :-1], 1e2 * torch.ones_like(z[:, :1])], -1)
    a_s = 1 - torch.exp(-delta * f["s_sigma"])
    if sched_mult < 1:
        if not encode_candidate:  # rendering.py:134-150
            if not encode_feat:
                raise NotImplementedError("rendering.py:149-150 (`raise NotImplemented`): no candidate-free path without features")
            w = a_s * _excl_cumprod(1 - a_s)
            res[f"s_weights_{typ}"] = w
            res[f"feat_{typ}"] = (w[..., None] * f["s_feat"]).sum(1)
        else:  # rendering.py:151-182
            a_c = 1 - torch.exp(-delta * f["c_sigma"])
            a_all = 1 - torch.exp(-delta * (f["s_sigma"] + f["c_sigma"]))
            T = _excl_cumprod(1 - a_all)
            s_w, c_w, w = a_s * T, a_c * T, a_all * T
            res[f"c_weights_{typ}"] = w
            res[f"c_depth_{typ}"] = (w * z).sum(1)
            if encode_feat:
                res[f"feat_{typ}"] = (s_w[..., None] * f["s_feat"]).sum(1) + (c_w[..., None] * f["c_feat"]).sum(1)
            else:  # rendering.py:177-189
                res[f"c_rgb_{typ}"] = (s_w[..., None] * f["s_rgb"]).sum(1) + (c_w[..., None] * f["c_rgb"]).sum(1)
            res[f"t_weight_{typ}"] = c_w.sum(1)
    w_s = a_s * _excl_cumprod(1 - a_s)
    if sched_mult > 0:  # rendering.py:195-209
        res[f"s_weights_{typ}"] = w_s
        res[f"s_rgb_{typ}"] = (w_s[..., None] * f["s_rgb"]).sum(1)
    res[f"s_depth_{typ}"] = (w_s * z).sum(1)  # rendering.py:211-218


# --------------------------------------------------------------------------------------
# a11: inverse-CDF resampling  (models/rendering.py:7-50)
# --------------------------------------------------------------------------------------
def sample_pdf(bins: Tensor, weights: Tensor, n: int, det: bool, u: Optional[Tensor] = None,
               eps: float = 1e-5) -> Tensor:
    """bins [R,B+1], weights [R,B] -> [R,n].  `u` overrides the random draw (tests inject it)."""
    R, B = weights.shape
    w = weights + eps
    pdf = w / w.sum(1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    if det:
        u = torch.linspace(0, 1, n, dtype=bins.dtype).expand(R, n)
    elif u is None:
        u = torch.rand(R, n, dtype=bins.dtype)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = (hi - 1).clamp_min(0)
    hi = hi.clamp_max(B)
    c_lo, c_hi = cdf.gather(1, lo), cdf.gather(1, hi)
    b_lo, b_hi = bins.gather(1, lo), bins.gather(1, hi)
    den = c_hi - c_lo
    den = torch.where(den < eps, torch.ones_like(den), den)
    return b_lo + (u - c_lo) / den * (b_hi - b_lo)


def py_round(x: float) -> int:
    """Python round() = banker's rounding (rendering.py:277)."""
    return int(round(x))


# a5: stratified coarse depths  (models/rendering.py:232-249)
def coarse_depths(near: Tensor, far: Tensor, N_samples: int, use_disp: bool, perturb: float,
                  u: Optional[Tensor]) -> Tensor:
    """near, far [R,1] -> z [R,N_samples]; `u` [R,N_samples] are the uniform draws used when perturb > 0."""
    s = torch.linspace(0, 1, N_samples, dtype=near.dtype)
    if not use_disp:
        z = near * (1 - s) + far * s
    else:
        z = 1 / (1 / near * (1 - s) + 1 / far * s)
    z = z.expand(near.shape[0], N_samples)
    if perturb > 0:  # rendering.py:240-249
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper = torch.cat([mid, z[:, -1:]], -1)
        lower = torch.cat([z[:, :1], mid], -1)
        z = lower + (upper - lower) * (perturb * u)
    return z


# --------------------------------------------------------------------------------------
# a5, a12: render_rays  (models/rendering.py:53-314)
# --------------------------------------------------------------------------------------
def render_rays(models: Dict[str, Params], cfgs: Dict[str, NerfCfg], embeddings: Dict[str, Tensor],
                rays: Tensor, img_idx: Tensor, sched_mult: float, N_samples: int = 64,
                use_disp: bool = False, perturb: float = 0, N_importance: int = 0,
                progress: float = 0.0, u_list: Optional[Sequence[Tensor]] = None,
                keep: Optional[dict] = None, z_fine_override: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """models: {"nerf_coarse": params, "nerf_fine": params}; embeddings: {"coarse_a": weight[N_img,48], ...}.
    u_list (optional): explicit uniform draws consumed in the reference's RNG call order:
    [coarse jitter [R,Nc]] then the sample_pdf draws.  `keep` (optional dict) receives z_coarse/z_fine.
    z_fine_override (tests only): evaluate the fine pass at these depths instead of the resampled ones -- the
    resampling has no gradient, so this isolates everything else from the (ill-conditioned) inverse-CDF step."""
    draws = list(u_list) if u_list is not None else None

    def draw(shape):
        if draws is not None:
            t = draws.pop(0)
            assert tuple(t.shape) == tuple(shape), (t.shape, shape)
            return t
        return torch.rand(*shape, dtype=rays.dtype)

    R = rays.shape[0]
    o, d = rays[:, 0:3], rays[:, 3:6]
    z = coarse_depths(rays[:, 6:7], rays[:, 7:8], N_samples, use_disp, perturb,
                      draw((R, N_samples)) if perturb > 0 else None)

    res: Dict[str, Tensor] = {}

    def run(which: str, zz: Tensor):
        cfg, p = cfgs[which], models[which]
        S = zz.shape[1]
        xyz = (o[:, None, :] + d[:, None, :] * zz[..., None]).reshape(-1, 3)
        vdir = d.detach()[:, None, :].expand(R, S, 3).reshape(-1, 3)  # rendering.py:104-106 (detached)
        a = c = None
        if cfg.encode_appearance:
            a = embeddings[f"{cfg.typ}_a"][img_idx][:, None, :].expand(R, S, -1).reshape(R * S, -1)
        if cfg.encode_candidate:
            c = embeddings[f"{cfg.typ}_c"][img_idx][:, None, :].expand(R, S, -1).reshape(R * S, -1)
        f = nerf_field(p, cfg, xyz, vdir, a, c, sched_mult, progress)
        f = {k: (v.reshape(R, S) if "sigma" in k else v.reshape(R, S, -1)) for k, v in f.items()}
        composite(res, cfg.typ, f, zz, sched_mult, cfg.encode_candidate, cfg.encode_feat)

    run("nerf_coarse", z)
    if keep is not None:
        keep["z_coarse"] = z
    if N_importance > 0:
        cfg = cfgs["nerf_fine"]
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        det = perturb == 0

        def resample(key, n):
            w = res[key][:, 1:-1].detach()
            return sample_pdf(mid, w, n, det, None if det else draw((R, n)))

        if cfg.encode_candidate:  # rendering.py:267-300
            if sched_mult == 0:
                z = torch.sort(torch.cat([z, resample("c_weights_coarse", N_importance)], -1), -1)[0]
            elif 0 < sched_mult < 1:
                n_s = py_round(sched_mult * N_importance)
                z_c = resample("c_weights_coarse", N_importance - n_s)
                z_s = resample("s_weights_coarse", n_s)
                z = torch.sort(torch.cat([z, z_s, z_c], -1), -1)[0]
            elif sched_mult == 1:
                z = torch.sort(torch.cat([z, resample("s_weights_coarse", N_importance)], -1), -1)[0]
        else:  # rendering.py:300-307
            z = torch.sort(torch.cat([z, resample("s_weights_coarse", N_importance)], -1), -1)[0]
        if z_fine_override is not None:
            z = z_fine_override
        if keep is not None:
            keep["z_fine"] = z
        run("nerf_fine", z)
    return res


# --------------------------------------------------------------------------------------
# a13: TransientNet  (models/transient_net.py:27-38)
# --------------------------------------------------------------------------------------
def transient_net(p: Params, feat: Tensor, ts: Tensor, beta_min: float = 0.1) -> Dict[str, Tensor]:
    h = feat
    for i in (0, 2, 4, 6):
        h = torch.relu(_lin(p, f"feat_encoder.{i}", h))
    e = _lin(p, "final_encoder", h)
    t = torch.relu(_lin(p, "t_encoder.0", torch.cat([e, p["embedding_t.weight"][ts]], -1)))
    alpha = torch.sigmoid(_lin(p, "alpha_layer.0", h))
    rgb = torch.sigmoid(_lin(p, "rgb_layer.0", t))
    beta = F.softplus(_lin(p, "beta_layer.0", t)) * alpha + beta_min
    return {"alpha": alpha, "rgb": rgb, "beta": beta}


# a16: candidate schedule  (models/nerf_system.py:452-461)
def schedule_mult(progress: float, schedule) -> float:
    s, e = schedule
    if progress < s:
        return 0
    if progress > e:
        return 1
    return (1 - math.cos(math.pi * (progress - s) / (e - s))) / 2


# a15: depth-prior affine  (models/nerf_system.py:169-177)
def depth_prior(depth_scale_rows: Tensor, inv_depths: Tensor, near: float, far: float) -> Tensor:
    scale, shift = depth_scale_rows[:, 0], depth_scale_rows[:, 1]
    p = inv_depths * torch.exp(scale) + shift
    p = torch.where(p < 1 / far, torch.full_like(p, 1 / far), p)  # masked assignment: no grad where clamped
    dep = 1.0 / p
    return torch.where(dep < near, torch.full_like(dep, near), dep)


# a14: transient blend appended by NeRFSystem.forward  (models/nerf_system.py:128-146)
def blend_transient(res: Dict[str, Tensor], t: Dict[str, Tensor], fine: bool) -> None:
    al = t["alpha"]
    res["rgb_coarse"] = res["s_rgb_coarse"] * (1 - al.detach()) + t["rgb"].detach() * al.detach()
    if fine:
        res["rgb_fine"] = res["s_rgb_fine"] * (1 - al) + t["rgb"] * al
    res["t_beta"], res["t_alpha"] = t["beta"], al


# a17: UPNeRFLoss  (losses.py:21-64)
def upnerf_loss(res: Dict[str, Tensor], rgb: Tensor, feat: Tensor, depth: Tensor, m: float,
                depth_mult: float = 1e-3, alpha_reg: float = 1.0, fine: bool = True, encode_feat: bool = True) -> Dict[str, Tensor]:
    out = {}
    for typ, tag in (("coarse", "c"), ("fine", "f")):
        if typ == "fine" and not fine:
            break
        if m < 1:
            l = (res[f"s_depth_{typ}"] - depth).abs()
            if f"t_weight_{typ}" in res:
                l = l * (1 - res[f"t_weight_{typ}"].detach())
            out[f"l_depth_{tag}"] = l.mean() * depth_mult * (1 - m)
            if encode_feat:
                out[f"l_feat_{tag}"] = ((res[f"feat_{typ}"] - feat) ** 2).mean() * (1 - m)
            else:  # losses.py:33-35, 54-56
                out[f"l_c_rgb_{tag}"] = ((res[f"c_rgb_{typ}"] - rgb) ** 2).mean() * (1 - m)
        if m > 0:
            sq = (res[f"s_rgb_{typ}"] - rgb) ** 2
            if typ == "coarse":
                out["l_rgb_c"] = sq.mean() * m / 2
            else:
                out["l_rgb_f"] = (sq / (2 * res["t_beta"] ** 2)).mean() * m
                out["l_beta"] = torch.log(res["t_beta"]).mean() * m
                out["l_alpha"] = res["t_alpha"].mean() * alpha_reg * m
    return out


# --------------------------------------------------------------------------------------
# One training-step forward = the glue of NeRFSystem.training_step  (nerf_system.py:150-186)
# --------------------------------------------------------------------------------------
def training_forward(state: Dict[str, Params], cfgs: Dict[str, NerfCfg], batch: Dict[str, Tensor], hp: dict,
                     progress: float, u_list: Optional[Sequence[Tensor]] = None, keep: Optional[dict] = None,
                     z_fine_override: Optional[Tensor] = None):
    """state: {"nerf_coarse","nerf_fine","transient_net": params, "embedding_coarse_a": weight, ...,
    "se3_refine": weight, "depth_scale": weight}.  Returns (loss_dict, results)."""
    idx = batch["img_idx"]
    if hp.get("pose.optimize", True):
        refine = se3_exp(state["se3_refine"][idx])
        pose = compose_pair(refine, batch["c2w"])  # compose([refine, c2w]) = c2w o refine
    else:
        pose = batch["c2w"]
    o, d = get_rays(batch["directions"], pose)
    rays = torch.cat([o, d, batch["ray_infos"]], 1)
    if keep is not None:  # checkers read d loss / d rays (per-ray: tests/test_hip_fullsize.py compares a slice of a big batch)
        keep["rays"] = rays
        if rays.requires_grad:
            rays.retain_grad()
    depth = depth_prior(state["depth_scale"][idx], batch["inv_depths"], hp["nerf.near"], hp["nerf.far"])
    m = schedule_mult(progress, hp["candidate_schedule"])
    emb = {k[len("embedding_"):]: v for k, v in state.items() if k.startswith("embedding_")}
    fine = hp["nerf.N_importance"] > 0
    res = render_rays({k: state[k] for k in ("nerf_coarse", "nerf_fine") if k in state}, cfgs, emb, rays, idx, m,
                      N_samples=hp["nerf.N_samples"], use_disp=hp.get("nerf.use_disp", False),
                      perturb=hp.get("nerf.perturb", 1.0), N_importance=hp["nerf.N_importance"],
                      progress=progress, u_list=u_list, keep=keep, z_fine_override=z_fine_override)
    if m > 0:
        t = transient_net(state["transient_net"], batch["feats"], idx, hp.get("t_net.beta_min", 0.1))
        blend_transient(res, t, fine)
    losses = upnerf_loss(res, batch["rgbs"], batch["feats"], depth, m, hp.get("loss.depth_mult", 1e-3),
                         hp.get("loss.alpha_reg", 1.0), fine, encode_feat=cfgs["nerf_coarse"].encode_feat)
    return losses, res


# a18: Adam (torch.optim.Adam defaults, eps=1e-8: utils/optim.py:20-33) + ExponentialLR (optim.py:36-45)
def sample_train_rays(buf: Dict[str, Tensor], idx: Tensor) -> Dict[str, Tensor]:
    """Train-split ray sampler: PhototourismDataset.__getitem__ (datasets/phototourism.py:420-454) for every index of
    `idx`, stacked like torch's default collate.  buf: all_ray_infos [N,3] (near, far, image index), all_directions
    [N,3], all_rgbs [N,3], all_pxl_coords [N,2] (row, column in [0,1]), all_inv_depths [N], feat_maps [I,h,h,C],
    poses [I,3,4] (poses_dict in image-index order).
    Quirk kept (phototourism.py:433-449): x2 = min(h-1, x1+1), so on the last row / column of a feature map BOTH
    interpolation weights of that axis are zero and the sampled feature is the zero vector."""
    out: Dict[str, List[Tensor]] = {k: [] for k in ("ray_infos", "directions", "img_idx", "c2w", "rgbs", "feats",
                                                      "inv_depths")}
    fm = buf["feat_maps"]
    h = fm.shape[1]
    for i in idx.tolist():
        img = buf["all_ray_infos"][i, 2].long()
        out["ray_infos"].append(buf["all_ray_infos"][i, :2])
        out["directions"].append(buf["all_directions"][i])
        out["img_idx"].append(img)
        out["c2w"].append(buf["poses"][img])
        out["rgbs"].append(buf["all_rgbs"][i])
        pm = buf["all_pxl_coords"][i] * (h - 1)
        y, x = pm
        y1, x1 = torch.floor(pm).long()
        y2, x2 = min(h - 1, int(y1) + 1), min(h - 1, int(x1) + 1)
        w11, w12 = (y2 - y) * (x2 - x), (y2 - y) * (x - x1)
        w21, w22 = (y - y1) * (x2 - x), (y - y1) * (x - x1)
        out["feats"].append(w11 * fm[img, y1, x1] + w12 * fm[img, y1, x2] + w21 * fm[img, y2, x1] + w22 * fm[img, y2, x2])
        out["inv_depths"].append(buf["all_inv_depths"][i])
    return {k: torch.stack(v) for k, v in out.items()}


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """One in-place Adam update, bias-corrected exactly like torch.optim.Adam (no amsgrad/weight decay)."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def exp_lr(lr0: float, lr_end: float, max_step: int, step: int) -> float:
    return lr0 * ((lr_end / lr0) ** (1.0 / max_step)) ** step

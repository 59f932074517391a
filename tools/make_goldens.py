#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference) on CPU.

Runs only in the build container (the reference does not travel to the GPU box).  Nothing of the reference is
copied: this script imports its modules, feeds them deterministic synthetic weights/batches from
upnerf_amd.synth, and stores inputs + outputs + gradients as data.

Import shims: utils/camera.py imports easydict (used only in procrustes_analysis, camera.py:364-382) and
utils/ray.py imports kornia (used only in get_ray_directions, ray.py:5-27); neither is installed, neither is on
the hot path, so empty stand-in modules are registered for the import only.

The orchestration files (models/nerf_system.py) need pytorch_lightning (absent), so the ~40 lines of glue of
NeRFSystem.training_step/forward (nerf_system.py:93-186) are re-stated below with the reference's own leaf
functions: se3_to_SE3, compose, get_rays, render_rays, TransientNet, UPNeRFLoss.

RNG: torch.rand / torch.rand_like are wrapped while render_rays runs so that the uniform draws are recorded
(in call order) and stored in the fixture; checkers inject them instead of re-drawing.
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

class _AttrDict(dict):  # procrustes_analysis (camera.py:381) returns edict(...) and its callers read fields as attributes
    __getattr__ = dict.__getitem__


_ed = types.ModuleType("easydict"); _ed.EasyDict = _AttrDict; sys.modules["easydict"] = _ed
_ko = types.ModuleType("kornia"); _ko.create_meshgrid = None; sys.modules["kornia"] = _ko
sys.path.insert(0, REF)

import models.rendering as ref_rendering  # noqa: E402
import models.nerf as ref_nerf  # noqa: E402
import models.transient_net as ref_tnet  # noqa: E402
import losses as ref_losses  # noqa: E402
import utils.camera as ref_camera  # noqa: E402
import utils.ray as ref_ray  # noqa: E402

from upnerf_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MAX_GRAD_ELEMS = 384


class RecordRand:
    """Record every uniform draw the reference makes inside render_rays, and every tensor torch.sort returns there (the
    only sort of render_rays is the merge of the coarse and the resampled depths, rendering.py:277-307: its output is the
    reference's own z_vals of the fine pass)."""

    def __enter__(self):
        self.draws, self.sorted = [], []
        self._rand, self._rand_like, self._sort = torch.rand, torch.rand_like, torch.sort

        def rand(*a, **k):
            t = self._rand(*a, **k); self.draws.append(t.clone()); return t

        def rand_like(x, **k):
            t = self._rand_like(x, **k); self.draws.append(t.clone()); return t

        def sort(*a, **k):
            r = self._sort(*a, **k); self.sorted.append(r[0].detach().clone()); return r

        torch.rand, torch.rand_like, torch.sort = rand, rand_like, sort
        return self

    def __exit__(self, *exc):
        torch.rand, torch.rand_like, torch.sort = self._rand, self._rand_like, self._sort


def schedule_mult(progress, sched):  # nerf_system.py:452-461 (cannot import: needs pytorch_lightning)
    s, e = sched
    if progress < s:
        return 0
    if progress > e:
        return 1
    return (1 - math.cos(math.pi * (progress - s) / (e - s))) / 2


def build(case):
    n_img = case["n_img"]
    kw = dict(D=case["D"], W=case["W"], feat_dim=384, xyz_L=10, dir_L=4, appearance_dim=48, candidate_dim=16)
    if case.get("encode_feat") is False:  # nerf_system.py:373-374: encode_feat = feat_dim > 0, feat_dim = 0
        kw.update(encode_feat=False, feat_dim=0)
    models, sds = {}, {}
    for typ in ("coarse", "fine") if case["Nf"] > 0 else ("coarse",):
        m = ref_nerf.NeRF(typ, c2f=case["c2f"], **kw)
        sd = synth.nerf_state(typ, seed=case["seed"], progress=case["progress"], sigma_bias=case.get("sigma_bias", 0.0),
                              sigma_gain=case.get("sigma_gain", 1.0), trunk_gain=case.get("trunk_gain", 1.0), **kw)
        m.load_state_dict(sd)
        if case.get("encode_candidate") is False:
            m.encode_candidate = False
        models[f"nerf_{typ}"] = m
    tn = ref_tnet.TransientNet(n_img, beta_min=0.1, trasient_dim=128, feat_dim=384)
    tn.load_state_dict(synth.transient_state(n_img, seed=case["seed"]))
    tabs = synth.tables(n_img, seed=case["seed"], fine=case["Nf"] > 0)
    emb = {k: torch.nn.Embedding.from_pretrained(v.clone(), freeze=False) for k, v in tabs.items()}
    return models, tn, emb


def run_case(name, case):
    torch.manual_seed(case["seed"])
    models, tn, emb = build(case)
    b = synth.batch(case["R"], case["n_img"], seed=case["seed"] + 1, identity_c2w=case.get("identity_c2w", True))
    idx = b["img_idx"]
    fine = case["Nf"] > 0
    # --- glue restated from NeRFSystem.training_step (nerf_system.py:150-186) ---
    if case["pose_opt"]:
        refine = ref_camera.lie.se3_to_SE3(emb["se3_refine"](idx))
        pose = ref_camera.pose.compose([refine, b["c2w"]])
    else:
        pose = b["c2w"]
    rays_o, rays_d = ref_ray.get_rays(b["directions"], pose)
    rays = torch.cat([rays_o, rays_d, b["ray_infos"]], 1)
    near, far = 0.1, 5.0
    scale, shift = torch.unbind(emb["depth_scale"](idx), 1)
    pred_inv = b["inv_depths"] * torch.exp(scale) + shift
    pred_inv[pred_inv < 1 / far] = 1 / far
    depth = 1.0 / pred_inv
    depth[depth < near] = near
    m = case["sched"] if "sched" in case else schedule_mult(case["progress"], (0.1, 0.5))
    embeddings = {k[len("embedding_"):]: v for k, v in emb.items() if k.startswith("embedding_")}
    with RecordRand() as rec:
        res = ref_rendering.render_rays(models=models, embeddings=embeddings, rays=rays, img_idx=idx, sched_mult=m,
                                        sched_phase=0, N_samples=case["Nc"], use_disp=case.get("use_disp", False),
                                        perturb=case["perturb"], N_importance=case["Nf"], white_back=False,
                                        encode_feat=case.get("encode_feat", True), validation=False)
    # NeRFSystem.forward transient blend (nerf_system.py:128-146)
    if m > 0:
        t = tn(b["feats"], idx)
        res["rgb_coarse"] = res["s_rgb_coarse"] * (1 - t["alpha"].detach()) + t["rgb"].detach() * t["alpha"].detach()
        if fine:
            res["rgb_fine"] = res["s_rgb_fine"] * (1 - t["alpha"]) + t["rgb"] * t["alpha"]
        res["t_beta"], res["t_alpha"] = t["beta"], t["alpha"]
    loss_fn = ref_losses.UPNeRFLoss(depth_mult=1e-3, alpha_reg=1.0, encode_feat=case.get("encode_feat", True), fine=fine)
    loss_d = loss_fn(res, b["rgbs"], b["feats"], depth, m)
    loss = sum(l for l in loss_d.values())
    if rays.requires_grad:
        rays.retain_grad()
    loss.backward()

    out = {"meta_sched": np.float64(m)}
    for k, v in case.items():
        if k == "c2f":
            out["cfg_c2f"] = np.array(v if v is not None else (-1.0, -1.0), dtype=np.float64)
        elif isinstance(v, (int, float, bool)):
            out["cfg_" + k] = np.float64(v)
    for i, u in enumerate(rec.draws):
        out[f"u_{i}"] = u.numpy()
    out["n_draws"] = np.int64(len(rec.draws))
    if fine:  # the reference's own fine depths: the HIP path can be evaluated AT them (render_rays z_fine=)
        zf = [t for t in rec.sorted if tuple(t.shape) == (case["R"], case["Nc"] + case["Nf"])]
        assert len(zf) == 1, [tuple(t.shape) for t in rec.sorted]
        out["z_fine"] = zf[0].numpy()
    out["in_rays"] = rays.detach().numpy()
    out["in_depth"] = depth.detach().numpy()
    for k, v in res.items():
        out["res_" + k] = v.detach().numpy()
    for k, v in loss_d.items():
        out["loss_" + k] = v.detach().numpy()
    out["loss_total"] = loss.detach().numpy()
    out["grad_rays"] = rays.grad.numpy() if rays.grad is not None else np.zeros_like(out["in_rays"])

    def put_grad(key, g):
        if g is None:
            out["gradnone_" + key] = np.int64(1)
            return
        g = g.detach().reshape(-1)
        out["gsum_" + key] = np.array([g.double().sum().item(), g.double().abs().sum().item()])
        if g.numel() > MAX_GRAD_ELEMS:
            stride = g.numel() // MAX_GRAD_ELEMS
            out["gstride_" + key] = np.int64(stride)
            g = g[::stride]
        out["grad_" + key] = g.numpy().copy()

    for k, e_ in emb.items():
        put_grad(k + ".weight", e_.weight.grad)
    for mk, mod in models.items():
        for pn, p in mod.named_parameters():
            put_grad(f"{mk}.{pn}", p.grad)
    for pn, p in tn.named_parameters():
        put_grad(f"transient_net.{pn}", p.grad)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: sched={m:.4f} draws={[tuple(u.shape) for u in rec.draws]} loss={loss.item():.6f} "
          f"keys={sorted(res.keys())} -> {os.path.getsize(path) / 1024:.0f} KiB")


BASE = dict(n_img=12, seed=3, sigma_bias=2.0)
CASES = {
    # BASELINE.json configs[0] shape (4-layer x 64 MLP, coarse only, poses frozen, sched 0)
    "cfg1_small": dict(BASE, R=24, D=4, W=64, Nc=64, Nf=0, c2f=None, progress=0.0, perturb=1.0, pose_opt=False),
    "cfg1_small_fine": dict(BASE, R=10, D=4, W=64, Nc=32, Nf=32, c2f=(0.1, 0.5), progress=0.3, perturb=1.0, pose_opt=True),
    # BASELINE.json configs[1] shape at a handful of rays, the three schedule phases
    "cfg2_phase0": dict(BASE, R=6, D=8, W=256, Nc=64, Nf=128, c2f=(0.1, 0.5), progress=0.05, perturb=1.0, pose_opt=True),
    "cfg2_phase1": dict(BASE, R=6, D=8, W=256, Nc=64, Nf=128, c2f=(0.1, 0.5), progress=0.3, perturb=1.0, pose_opt=True),
    "cfg2_phase2": dict(BASE, R=6, D=8, W=256, Nc=64, Nf=128, c2f=(0.1, 0.5), progress=0.8, perturb=1.0, pose_opt=True),
    # "trained-like" statistics at the config #2 shape: trunk weights x1.6 and density heads x24 (densities
    # from 0 to ~50, saturated alphas, activations over several decades -- what the per-tile / per-tensor exponents of the
    # f16x3 split have to follow), all ten encoding bands on (progress 0.8, sched 1) resp. eight and a half (0.45, sched 0.96)
    "cfg2_trained_p08": dict(BASE, R=6, D=8, W=256, Nc=64, Nf=128, c2f=(0.1, 0.5), progress=0.8, perturb=1.0, pose_opt=True,
                             sigma_gain=24.0, trunk_gain=1.6),
    "cfg2_trained_p045": dict(BASE, R=6, D=8, W=256, Nc=64, Nf=128, c2f=(0.1, 0.5), progress=0.45, perturb=1.0, pose_opt=True,
                              sigma_gain=24.0, trunk_gain=1.6),
    # the reference's SHIPPED sampling shape (configs/default.yaml:8-9: N_samples 128, N_importance 128 -> the fine pass runs 256
    # samples per ray at W = 256), the three schedule phases
    "yaml_phase0": dict(BASE, R=4, D=8, W=256, Nc=128, Nf=128, c2f=(0.1, 0.5), progress=0.05, perturb=1.0, pose_opt=True),
    "yaml_phase1": dict(BASE, R=4, D=8, W=256, Nc=128, Nf=128, c2f=(0.1, 0.5), progress=0.3, perturb=1.0, pose_opt=True),
    "yaml_phase2": dict(BASE, R=4, D=8, W=256, Nc=128, Nf=128, c2f=(0.1, 0.5), progress=0.8, perturb=1.0, pose_opt=True),
    # deterministic resampling (validation path: perturb=0 -> det=True), non-identity c2w
    "cfg2_det_phase1": dict(BASE, R=5, D=8, W=256, Nc=64, Nf=128, c2f=(0.1, 0.5), progress=0.25, perturb=0.0,
                            pose_opt=True, identity_c2w=False),
    # banker's rounding of n_s: 0.33203125*128 = 42.5 -> 42
    "small_round_half": dict(BASE, R=7, D=4, W=64, Nc=64, Nf=128, c2f=(0.1, 0.5), progress=0.3, sched=0.33203125,
                             perturb=1.0, pose_opt=True),
    # disparity sampling
    "small_disp": dict(BASE, R=7, D=4, W=64, Nc=48, Nf=16, c2f=None, progress=0.0, sched=0.5, perturb=1.0,
                       pose_opt=True, use_disp=True),
    # TTO shape: sched 1, candidate head disabled (nerf_system_optmize.py:265-266)
    "small_tto": dict(BASE, R=9, D=4, W=64, Nc=32, Nf=64, c2f=(0.1, 0.5), progress=0.9, sched=1.0, perturb=0.0,
                      pose_opt=True, encode_candidate=False),
    # non-candidate branch with sched<1 (rendering.py:134-150)
    "small_nocand": dict(BASE, R=9, D=4, W=64, Nc=32, Nf=32, c2f=(0.1, 0.5), progress=0.3, sched=0.5, perturb=1.0,
                         pose_opt=True, encode_candidate=False),
    # encode_feat = False (nerf.feat_dim = 0: colour head on xyz_encoding_final, rgb_candidate_layer, c_rgb maps and the l_c_rgb
    # terms -- nerf.py:52-56, 75-78, 110-123; rendering.py:177-190; losses.py:33-35, 54-56), the three schedule phases; the
    # 256-wide one runs the f16x3 / f16 kernels
    "nofeat_phase0": dict(BASE, R=8, D=4, W=64, Nc=32, Nf=32, c2f=(0.1, 0.5), progress=0.05, perturb=1.0, pose_opt=True, encode_feat=False),
    "nofeat_phase1": dict(BASE, R=8, D=4, W=64, Nc=32, Nf=32, c2f=(0.1, 0.5), progress=0.3, perturb=1.0, pose_opt=True, encode_feat=False),
    "nofeat_phase2": dict(BASE, R=8, D=4, W=64, Nc=32, Nf=32, c2f=(0.1, 0.5), progress=0.8, perturb=1.0, pose_opt=True, encode_feat=False),
    "nofeat_w256_phase1": dict(BASE, R=5, D=8, W=256, Nc=64, Nf=128, c2f=(0.1, 0.5), progress=0.3, perturb=1.0, pose_opt=True,
                               encode_feat=False),
    # all PE bands masked (progress < c2f start), zero se3 is covered by test_pose (separate fixture)
    "small_allmasked": dict(BASE, R=8, D=4, W=64, Nc=32, Nf=32, c2f=(0.1, 0.5), progress=0.02, perturb=1.0, pose_opt=True),
}


def tto_step_fixture():
    """One test-time-optimisation step (a19) with the TTO loss line itself: the glue of NeRFSystemOptimize.training_step /
    forward (nerf_system_optmize.py:113-129, 84-104) re-stated with the reference's own leaf functions -- sched_mult = 1.0,
    encode_candidate off on both fields (265-266), ONE test image whose appearance row and se(3) row are the only trainables
    (254-262, 48-64), loss = mean((s_rgb_fine - rgbs)^2) (line 129).  Shape and weights of the `small_tto` case; the test
    image's rows are rows 3 (appearance) and 5 (pose) of that case's tables, as tests/test_hip_parity.py has set them since
    round 1.  Stored: inputs, every result map, the loss, the two table gradients, the ray gradient."""
    case = CASES["small_tto"]
    torch.manual_seed(case["seed"])
    models, _tn, emb = build(case)
    b = synth.batch(case["R"], case["n_img"], seed=case["seed"] + 1, identity_c2w=case.get("identity_c2w", True))
    idx0 = torch.zeros_like(b["img_idx"])
    fine_a = torch.nn.Embedding.from_pretrained(emb["embedding_fine_a"].weight.detach()[3:4].clone(), freeze=False)
    se3 = torch.nn.Embedding.from_pretrained(emb["se3_refine"].weight.detach()[5:6].clone(), freeze=False)
    embeddings = {k[len("embedding_"):]: v for k, v in emb.items() if k.startswith("embedding_")}
    embeddings["fine_a"] = fine_a
    refine = ref_camera.lie.se3_to_SE3(se3(idx0))
    pose = ref_camera.pose.compose([refine, b["c2w"]])
    rays_o, rays_d = ref_ray.get_rays(b["directions"], pose)
    rays = torch.cat([rays_o, rays_d, b["ray_infos"]], 1)
    res = ref_rendering.render_rays(models=models, embeddings=embeddings, rays=rays, img_idx=idx0, sched_mult=1.0,
                                    sched_phase=2, N_samples=case["Nc"], use_disp=False, perturb=case["perturb"],
                                    N_importance=case["Nf"], white_back=False, encode_feat=True, validation=False)
    loss = ((res["s_rgb_fine"] - b["rgbs"]) ** 2).mean()
    rays.retain_grad()
    loss.backward()
    out = {"loss": loss.detach().numpy(), "in_rays": rays.detach().numpy(), "grad_rays": rays.grad.numpy(),
           "grad_embedding_fine_a": fine_a.weight.grad.numpy(), "grad_se3_refine": se3.weight.grad.numpy(),
           "row_a": np.int64(3), "row_se3": np.int64(5)}
    for k, v in res.items():
        out["res_" + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "small_tto_step.npz"), **out)
    print(f"small_tto_step: loss={loss.item():.6f} keys={sorted(res.keys())}")


def leaf_fixtures():
    """Leaf-function vectors: se3 exp (incl. w = 0), PE layout, sample_pdf edge cases."""
    out = {}
    wu = synth.uniform("leaf_se3", (9, 6), 5) * 0.5
    wu[0] = 0.0
    wu[1, :3] = 0.0
    wu = wu.clone().requires_grad_(True)
    Rt = ref_camera.lie.se3_to_SE3(wu)
    probe = synth.uniform("leaf_probe", (9, 3, 4), 5)
    (Rt * probe).sum().backward()
    out["se3_in"], out["se3_out"], out["se3_probe"], out["se3_grad"] = wu.detach().numpy(), Rt.detach().numpy(), probe.numpy(), wu.grad.numpy()
    x = synth.uniform("leaf_pe", (11, 3), 5) * 4.0
    for tag, c2f, prog in (("none", None, 0.0), ("mid", (0.1, 0.5), 0.3), ("frac", (0.1, 0.5), 0.27)):
        m = ref_nerf.NeRF("coarse", D=2, W=8, c2f=c2f)
        m.progress.data.fill_(prog)
        out[f"pe_{tag}"] = m.positional_encoding(x, 10).detach().numpy()
        out[f"pedir_{tag}"] = m.positional_encoding(x, 4).detach().numpy()
    out["pe_x"] = x.numpy()
    # sample_pdf: searchsorted(right=True) edges, zero-weight bins, det and explicit u
    bins = torch.tensor([[0.0, 1.0, 2.0, 4.0]]).repeat(2, 1)
    w = torch.tensor([[0.25, 0.25, 0.5], [0.0, 1.0, 0.0]]) - 1e-5
    u = torch.tensor([[0.0, 0.25, 0.5, 0.999, 1.0], [0.0, 1e-6, 0.5, 0.99999, 1.0]])
    real_rand = torch.rand
    torch.rand = lambda *a, **k: u.clone()
    try:
        out["pdf_edge_out"] = ref_rendering.sample_pdf(bins, w, 5, det=False).numpy()
    finally:
        torch.rand = real_rand
    out["pdf_edge_bins"], out["pdf_edge_w"], out["pdf_edge_u"] = bins.numpy(), w.numpy(), u.numpy()
    bins2 = torch.sort(synth.uniform("leaf_bins", (6, 63), 5, 0.1, 5.0), -1)[0]
    w2 = synth.uniform("leaf_w", (6, 62), 5, 0.0, 1.0) ** 4
    w2[2] = 0.0
    out["pdf_det_out"] = ref_rendering.sample_pdf(bins2, w2, 128, det=True).numpy()
    out["pdf_det_bins"], out["pdf_det_w"] = bins2.numpy(), w2.numpy()
    np.savez_compressed(os.path.join(OUT, "leaf.npz"), **out)
    print("leaf fixtures written")


def pose_align_fixture():
    """Pose evaluation / TTO pose initialisation: eval.py:28-40 and nerf_system_optmize.py:279-317.  utils/metric.py cannot
    be imported (lpips, kornia.losses are absent), so its three pose helpers (metric.py:34-62: a dozen lines of glue over
    utils/camera.py) are re-stated here with the reference's own camera functions, as run_case does for nerf_system.py."""
    P, lie = ref_camera.pose, ref_camera.lie

    def parse_raw(p):  # metric.py:34-39
        flip = P(R=torch.diag(torch.tensor([1, -1, -1])))
        return P.compose([flip, P.invert(P.compose([flip, p[:3]]))])

    def prealign(pose, gt):  # metric.py:42-52
        z = torch.zeros(1, 1, 3)
        c, cg = ref_camera.cam2world(z, pose)[:, 0], ref_camera.cam2world(z, gt)[:, 0]
        s = ref_camera.procrustes_analysis(cg, c)
        ca = (c - s.t1) / s.s1 @ s.R.t() * s.s0 + s.t0
        Ra = pose[..., :3] @ s.R.t()
        return P(R=Ra, t=(-Ra @ ca[..., None])[..., 0]), s

    N, T = 12, 3
    gt_se3 = synth.uniform("pa_gt", (N + T, 6), 5) * torch.tensor([0.6, 0.6, 0.6, 2.0, 2.0, 2.0])
    gt_all = lie.se3_to_SE3(gt_se3)
    gt_train, gt_test = gt_all[:N], gt_all[N:]
    # the trained frame: GT moved by one similarity (rotation, scale 1.7, shift) + a small per-camera error; built on the
    # camera centres / world-to-camera rotations the way the alignment itself reads them
    G = lie.se3_to_SE3(torch.tensor([0.3, -0.2, 0.5, 0.4, -0.1, 0.2]))
    R_g = G[:, :3]
    noise = lie.se3_to_SE3(synth.uniform("pa_noise", (N, 6), 5) * 0.02)
    raw = []
    for i in range(N):
        R, t = gt_train[i, :, :3], gt_train[i, :, 3]
        raw.append(P.compose([noise[i], P(R=R_g @ R, t=1.7 * (R_g @ t) + G[:, 3])]))
    noised = torch.stack(raw)
    se3 = synth.uniform("pa_se3", (N, 6), 5) * 0.05
    refined = P.compose([lie.se3_to_SE3(se3), noised])  # eval.py:33-34
    pr = torch.stack([parse_raw(p) for p in refined.float()])
    gt = torch.stack([parse_raw(p) for p in gt_train.float()])
    al, s = prealign(pr, gt)
    R_err = ref_camera.rotation_distance(al[..., :3], gt[..., :3])  # metric.py:55-62
    t_err = (al[..., 3] - gt[..., 3]).norm(dim=-1)
    # TTO initial poses (nerf_system_optmize.py:279-317), with the trained se(3) composed with identity poses (line 286)
    ref_id = P.compose([lie.se3_to_SE3(se3), torch.stack([torch.eye(3, 4)] * N)])
    pr_id = torch.stack([parse_raw(p) for p in ref_id.float()])
    _, s2 = prealign(pr_id, gt)
    te = torch.stack([parse_raw(p) for p in gt_test.float()])
    cg = ref_camera.cam2world(torch.zeros(1, 1, 3), te)[:, 0]
    ca = (cg - s2.t0) / s2.s0 @ s2.R * s2.s1 + s2.t1
    Ra = te[..., :3] @ s2.R
    init = torch.stack([parse_raw(p) for p in P(R=Ra, t=(-Ra @ ca[..., None])[..., 0]).float()])
    out = {"se3": se3, "noised": noised, "gt_train": gt_train, "gt_test": gt_test, "refined": refined, "eval_pred": pr,
           "eval_gt": gt, "aligned": al, "sim_R": s.R, "sim_t0": s.t0, "sim_t1": s.t1, "sim_s0": s.s0, "sim_s1": s.s1,
           "R_err": R_err, "t_err": t_err, "refined_identity": ref_id, "init_test": init}
    np.savez_compressed(os.path.join(OUT, "pose_align.npz"), **{k: np.asarray(v) for k, v in out.items()})
    print("pose alignment fixture written; mean R error (deg)", float(R_err.mean()) * 180 / math.pi, "t", float(t_err.mean()))


def sampler_fixture():
    """Train-split ray sampler (datasets/phototourism.py:420-454): the REAL PhototourismDataset.__getitem__ run on small
    synthetic buffers (three images of different sizes, one 6x6x384 feature map each), default-collated.  The dataset
    module imports cv2 and torchvision at module level (absent here, used only by __init__'s file loading): empty
    stand-ins are registered for the import; the object is created without __init__ and given the buffers directly."""
    for name in ("cv2", "torchvision"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = types.ModuleType("torchvision.transforms")
    sys.modules["torchvision.transforms"] = sys.modules["torchvision"].transforms
    import datasets.phototourism as ref_ds
    from torch.utils.data.dataloader import default_collate
    wh = [(5, 7), (8, 6), (4, 4)]
    n_img, fh, C = len(wh), 6, 384
    ds = object.__new__(ref_ds.PhototourismDataset)
    ds.split, ds.feat_map_dir = "train", "synthetic"
    ds.img_ids_train = [11, 12, 13]
    poses = synth.uniform("smp_pose", (n_img, 3, 4), 9)
    ds.poses_dict = {id_: poses[i].numpy() for i, id_ in enumerate(ds.img_ids_train)}
    infos, dirs, rgbs, pxl, invd = [], [], [], [], []
    for i, (w, h) in enumerate(wh):
        n = w * h
        infos.append(torch.cat([torch.full((n, 1), 0.1 + 0.01 * i), torch.full((n, 1), 5.0 - 0.1 * i), torch.full((n, 1), float(i))], 1))
        dirs.append(synth.uniform(f"smp_dir{i}", (n, 3), 9))
        rgbs.append(synth.uniform(f"smp_rgb{i}", (n, 3), 9, 0.0, 1.0))
        invd.append(synth.uniform(f"smp_inv{i}", (n,), 9, 0.2, 10.0))
        h_pxl = torch.linspace(0, h - 1, h) / (h - 1)  # phototourism.py:296-305
        w_pxl = torch.linspace(0, w - 1, w) / (w - 1)
        hh, ww = torch.meshgrid(h_pxl, w_pxl, indexing="ij")
        pxl.append(torch.stack((hh, ww), -1).view(-1, 2))
    ds.all_ray_infos, ds.all_directions, ds.all_rgbs = torch.cat(infos), torch.cat(dirs), torch.cat(rgbs)
    ds.all_pxl_coords, ds.all_inv_depths = torch.cat(pxl), torch.cat(invd)
    fm = synth.uniform("smp_feat", (n_img, fh, fh, C), 9)
    ds.feat_maps = fm / torch.norm(fm, dim=-1, keepdim=True)
    N = len(ds.all_ray_infos)
    # every ray once (covers first/last rows and columns of every image, where the reference's weights vanish), then repeats
    idx = torch.cat([torch.arange(N), torch.tensor([0, N - 1, 34, 34, 35, 7])])
    batch = default_collate([ds[int(i)] for i in idx])
    out = {"idx": idx.numpy(), "poses": poses.numpy(), "all_ray_infos": ds.all_ray_infos.numpy(),
           "all_directions": ds.all_directions.numpy(), "all_rgbs": ds.all_rgbs.numpy(),
           "all_pxl_coords": ds.all_pxl_coords.numpy(), "all_inv_depths": ds.all_inv_depths.numpy(),
           "feat_maps": ds.feat_maps.numpy().astype(np.float32)}
    for k, v in batch.items():
        out["out_" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **out)
    print("sampler fixture written:", {k: tuple(v.shape) for k, v in batch.items()})


def state_key_fixture():
    """state_dict keys and shapes of the REAL reference modules (models/nerf.py:39-78, models/transient_net.py:11-25)
    for the configurations the path uses -- pins checkpoint interchange: only names and shapes, no values.
    `param_order` is the registration order of the parameters: torch.optim state dicts index parameters by it, so
    resuming a reference checkpoint's optimiser needs the same order."""
    import json
    out = {}
    for tag, kw in {"nerf_d8_w256": dict(D=8, W=256, feat_dim=384, xyz_L=10, dir_L=4, appearance_dim=48, candidate_dim=16),
                    "nerf_d4_w64": dict(D=4, W=64, feat_dim=384, xyz_L=10, dir_L=4, appearance_dim=48, candidate_dim=16),
                    "nerf_d8_w256_nocand": dict(D=8, W=256, feat_dim=384, xyz_L=10, dir_L=4, appearance_dim=48,
                                                candidate_dim=0),
                    # nerf.feat_dim = 0 (nerf_system.py:373-374): no feature layers, rgb_candidate_layer
                    "nerf_d8_w256_nofeat": dict(D=8, W=256, encode_feat=False, feat_dim=0, xyz_L=10, dir_L=4, appearance_dim=48,
                                                candidate_dim=16)}.items():
        m = ref_nerf.NeRF("coarse", c2f=(0.1, 0.5), **kw)
        out[tag] = {"kwargs": kw, "state": {k: list(v.shape) for k, v in m.state_dict().items()},
                    "param_order": [k for k, _ in m.named_parameters()]}
    t = ref_tnet.TransientNet(763, beta_min=0.1, trasient_dim=128, feat_dim=384)
    out["transient_763"] = {"kwargs": dict(N_images=763, beta_min=0.1, trasient_dim=128, feat_dim=384),
                            "state": {k: list(v.shape) for k, v in t.state_dict().items()},
                            "param_order": [k for k, _ in t.named_parameters()]}
    with open(os.path.join(OUT, "state_keys.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("state-key fixture written:", {k: len(v["state"]) for k, v in out.items()})


CONFIG_CASES = {  # YAML texts written for this test suite (not the reference's files) + command-line override lists
    "nested_and_literals": ("""\
exp: 'run A'
n: 7
lr: 1e-3
none_word: None
flag: False
path: data/x
lst: [0.1,0.5]
quoted_list: '[1, 2, 3]'
quoted_tuple: '(4, 5)'
nest:
  a: 1
  b:
    c: 2.5
    d: ['x', 'y']
    e: {}
  f: 1_000
empty:
""", []),
    "overrides": ("""\
nerf:
  N_samples: 64
pose:
  optimize: True
  c2f: [0.1,0.5]
""", ["nerf.N_importance", "0", "pose.c2f", "None", "exp_name", "my run", "val.img_idx", "[0,11]", "x.y.z", "1e-4",
        "nerf.N_samples", "32"]),
    "empty_file": ("", ["a", "b"]),
}


def _enc(v):
    if isinstance(v, tuple):
        return {"__tuple__": [_enc(x) for x in v]}
    if isinstance(v, list):
        return [_enc(x) for x in v]
    if isinstance(v, dict):
        return {"__dict__": {k: _enc(x) for k, x in v.items()}}
    return v


def config_fixture():
    """What the REAL configs/config.py (lines 12-99) makes of the reference's default.yaml and of the YAML texts /
    override lists above: flat dicts with type-tagged values (tuples vs lists).  Pins upnerf_amd/config.py."""
    import json
    import tempfile
    import configs.config as ref_config
    out = {"default": {k: _enc(v) for k, v in ref_config.default().items()}, "cases": {}}
    for name, (text, opts) in CONFIG_CASES.items():
        with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
            f.write(text)
        cfg = ref_config.get_from_path(f.name)
        ref_config.merge_from_list(cfg, opts)
        os.unlink(f.name)
        with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
            pass
        ref_config.save_yaml(cfg, f.name)  # round trip through the reference's writer and reader
        back = ref_config.load(f.name)
        os.unlink(f.name)
        out["cases"][name] = {"yaml": text, "opts": opts, "config": {k: _enc(v) for k, v in cfg.items()},
                              "after_save_and_load": {k: _enc(v) for k, v in back.items()}}
    with open(os.path.join(OUT, "config_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("config fixture written:", {k: len(v["config"]) for k, v in out["cases"].items()})


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]
    os.makedirs(OUT, exist_ok=True)
    if not only or "leaf" in only:
        leaf_fixtures()
    if not only or "sampler" in only:
        sampler_fixture()
    if not only or "pose_align" in only:
        pose_align_fixture()
    if not only or "state_keys" in only:
        state_key_fixture()
    if not only or "config" in only:
        config_fixture()
    if not only or "tto_step" in only:
        tto_step_fixture()
    for n, c in CASES.items():
        if not only or n in only:
            run_case(n, c)

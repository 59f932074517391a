"""Shared helpers: load a golden fixture and rebuild its inputs from upnerf_amd.synth (weights are not stored)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from upnerf_amd import synth  # noqa: E402
import upnerf_oracle as orc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f not in ("leaf.npz", "sampler.npz", "pose_align.npz", "small_tto_step.npz"))


class Case:
    encode_feat = True  # (subclasses that do not load a fixture: the feature-head configuration)

    def __init__(self, name):
        self.name = name
        self.g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        c = lambda k, d=None: self.g.get("cfg_" + k, d)
        self.R, self.n_img, self.seed = int(c("R")), int(c("n_img")), int(c("seed"))
        self.D, self.W, self.Nc, self.Nf = int(c("D")), int(c("W")), int(c("Nc")), int(c("Nf"))
        self.progress, self.perturb = float(c("progress")), float(c("perturb"))
        self.pose_opt = bool(c("pose_opt"))
        self.use_disp = bool(c("use_disp", 0))
        self.identity_c2w = bool(c("identity_c2w", 1))
        self.sigma_bias = float(c("sigma_bias", 0.0))
        self.sigma_gain, self.trunk_gain = float(c("sigma_gain", 1.0)), float(c("trunk_gain", 1.0))
        c2f = self.g["cfg_c2f"]
        self.c2f = None if c2f[0] < 0 else (float(c2f[0]), float(c2f[1]))
        self.encode_candidate = None if "cfg_encode_candidate" not in self.g else bool(c("encode_candidate"))
        self.encode_feat = bool(c("encode_feat", 1))  # False: nerf.feat_dim = 0 (nerf_system.py:373-374)
        self.sched = float(self.g["meta_sched"])
        if self.sched in (0.0, 1.0):
            self.sched = int(self.sched)
        self.u_list = [torch.from_numpy(self.g[f"u_{i}"]) for i in range(int(self.g["n_draws"]))]
        self.fine = self.Nf > 0
        # the reference's own fine depths (render_rays kwargs["z_fine"] / the oracle's z_fine_override evaluate the fine pass AT them)
        self.z_fine = torch.from_numpy(self.g["z_fine"]) if "z_fine" in self.g else None

    def nerf_kw(self):
        return dict(D=self.D, W=self.W, feat_dim=384 if self.encode_feat else 0, xyz_L=10, dir_L=4, appearance_dim=48,
                    candidate_dim=16)

    def state(self, requires_grad=True, dtype=torch.float32):
        """{"nerf_coarse": params, ..., "embedding_*": weight, "se3_refine": weight, "depth_scale": weight}."""
        st = {}
        for typ in ("coarse", "fine") if self.fine else ("coarse",):
            sd = synth.nerf_state(typ, seed=self.seed, progress=self.progress, sigma_bias=self.sigma_bias,
                                  sigma_gain=self.sigma_gain, trunk_gain=self.trunk_gain, encode_feat=self.encode_feat,
                                  **self.nerf_kw())
            sd.pop("progress")
            st[f"nerf_{typ}"] = {k: v.to(dtype).requires_grad_(requires_grad) for k, v in sd.items()}
        st["transient_net"] = {k: v.to(dtype).requires_grad_(requires_grad)
                               for k, v in synth.transient_state(self.n_img, seed=self.seed).items()}
        for k, v in synth.tables(self.n_img, seed=self.seed, fine=self.fine).items():
            st[k] = v.to(dtype).requires_grad_(requires_grad)
        return st

    def cfgs(self):
        return {f"nerf_{typ}": orc.NerfCfg(typ=typ, c2f=self.c2f, encode_candidate=self.encode_candidate, **self.nerf_kw())
                for typ in (("coarse", "fine") if self.fine else ("coarse",))}

    def batch(self, dtype=torch.float32):
        b = synth.batch(self.R, self.n_img, seed=self.seed + 1, identity_c2w=self.identity_c2w)
        return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in b.items()}

    def hparams(self):
        return {"pose.optimize": self.pose_opt, "nerf.near": 0.1, "nerf.far": 5.0, "candidate_schedule": (0.1, 0.5),
                "nerf.N_samples": self.Nc, "nerf.N_importance": self.Nf, "nerf.use_disp": self.use_disp,
                "nerf.perturb": self.perturb, "t_net.beta_min": 0.1, "loss.depth_mult": 1e-3, "loss.alpha_reg": 1.0}

    def expected_results(self):
        return {k[4:]: v for k, v in self.g.items() if k.startswith("res_")}

    def expected_losses(self):
        return {k[5:]: v for k, v in self.g.items() if k.startswith("loss_")}

    def expected_grads(self):
        """name -> (values, stride or None, (sum, abssum)); names with gradnone_ map to None."""
        out = {}
        for k, v in self.g.items():
            if k.startswith("grad_") and k != "grad_rays":
                n = k[5:]
                out[n] = (v, int(self.g["gstride_" + n]) if "gstride_" + n in self.g else None, self.g["gsum_" + n])
            elif k.startswith("gradnone_"):
                out[k[9:]] = None
        return out


def rel_err(a, b):
    """max-normalised error used by every parity gate: max|a-b| / max(max|b|, tiny)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)) if b.size else 0.0


def named_grads(state):
    """Flatten a state dict-of-dicts into golden grad names -> grad tensor (or None)."""
    out = {}
    for k, v in state.items():
        if isinstance(v, dict):
            for pn, p in v.items():
                out[f"{k}.{pn}"] = p.grad
        else:
            out[f"{k}.weight"] = v.grad
    return out

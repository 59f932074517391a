"""Mid-size parity on the GPU box: the HIP path against the CPU oracle directly -- 301 rays (between the 6-ray goldens and the
4096-ray property tests), 64 + 128 samples, two 8x256 fields, all three schedule phases plus "trained-like" magnitudes.

The oracle (oracle/upnerf_oracle.py) is pinned bit-for-bit-reproducible against the reference by tests/test_oracle_golden.py, so
agreement here is agreement with the reference at a batch size the committed fixtures cannot carry.  Comparison protocol:
  1. resampling: the GPU's fine depths equal the oracle's own to 1e-5 on more than 97 % of the draws and to 1e-3 on more than
     99.9 % (the inverse CDF divides by the bin mass, and the reference's `denom < eps -> 1` rule, rendering.py:44-46, lets a
     draw in a bin of mass ~eps hop a bin under 1e-7 perturbations of the coarse weights);
  2. everything else with the oracle evaluated AT the GPU's fine depths: per-ray maps and losses 1e-4, per-sample weights
     2e-4, every parameter gradient max(1e-3, min(4 x the reference's own fp32-vs-fp64 noise on that tensor, 3e-2)) -- the gates of
     tests/test_hip_parity.py;
  3. the fp16 field mode (BASELINE.json configs[3]) at its stated gates: maps 1e-2, per-sample weights 3e-2, gradients --
     INCLUDING every NeRF weight gradient -- relative L2 6e-2."""
import numpy as np
import pytest
import torch

from golden_util import Case, named_grads, orc, rel_err
from test_hip_parity import TOL_GRAD, TOL_MAP, TOL_W, build_system, grad_gate

pytestmark = pytest.mark.gpu


class SynthCase(Case):
    """A Case without a fixture file: same closed-form weights / batch generators, draws from a seeded generator."""

    def __init__(self, name, R, progress, seed=7, n_img=9, trunk_gain=1.0, sigma_gain=1.0, Nc=64, Nf=128):
        self.name, self.g = name, {}
        self.R, self.n_img, self.seed = R, n_img, seed
        self.D, self.W, self.Nc, self.Nf = 8, 256, Nc, Nf
        self.progress, self.perturb, self.pose_opt, self.use_disp, self.identity_c2w = progress, 1.0, True, False, False
        self.sigma_bias, self.sigma_gain, self.trunk_gain = 0.0, sigma_gain, trunk_gain
        self.c2f, self.encode_candidate, self.fine = (0.1, 0.5), None, True
        self.sched = orc.schedule_mult(float(torch.tensor(progress, dtype=torch.float32)), (0.1, 0.5))
        g = torch.Generator().manual_seed(1000 + seed)
        n_s = round(self.sched * self.Nf)
        shapes = [self.Nc] + ([self.Nf] if self.sched in (0, 1) else [self.Nf - n_s, n_s])  # the draw order
        self.u_list = [torch.rand(R, n, generator=g) for n in shapes]
        self.z_fine = None


RAYS = 301
CASES = {"phase0": dict(progress=0.05), "phase1": dict(progress=0.3), "phase2": dict(progress=0.8),
         "trained_p045": dict(progress=0.45, trunk_gain=1.6, sigma_gain=24.0),
         # the reference's shipped sampling shape (configs/default.yaml:8-9): 128 + 128 samples, the fine pass at S = 256
         "yaml_phase0": dict(progress=0.05, Nc=128, Nf=128), "yaml_phase1": dict(progress=0.3, Nc=128, Nf=128),
         "yaml_phase2": dict(progress=0.8, Nc=128, Nf=128)}


def oracle_at(c, z_fine, dtype):
    st = c.state(dtype=dtype)
    keep = {}
    losses, res = orc.training_forward(st, c.cfgs(), c.batch(dtype), c.hparams(), c.progress,
                                       u_list=[u.to(dtype) for u in c.u_list], keep=keep,
                                       z_fine_override=None if z_fine is None else z_fine.to(dtype))
    return st, losses, res, keep


def hip_step(c, field_mode):
    from upnerf_amd import rendering
    sysm = build_system(c)
    batch = {k: v.cuda() for k, v in c.batch().items()}
    old = rendering.FIELD_MODE
    rendering.FIELD_MODE = field_mode
    try:
        keep = {}
        loss, loss_d, res = sysm.compute_loss(batch, u_list=[u.clone() for u in c.u_list], keep=keep)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        rendering.FIELD_MODE = old
    return sysm, loss, loss_d, res, keep


@pytest.mark.parametrize("name", list(CASES))
def test_mid_size_batch_matches_the_oracle(name):
    c = SynthCase(name, RAYS, **CASES[name])
    sysm, loss, loss_d, res, keep = hip_step(c, "f16x3")
    zf = keep["z_fine"].cpu()
    # 1. resampling against the oracle's own fine depths (forward only)
    with torch.no_grad():
        _, _, _, okeep = oracle_at(c, None, torch.float32)
    dz = (zf - okeep["z_fine"]).abs()
    # conditioning, not a bug: z = bin + (u - cdf) / mass * width amplifies the 1e-7 differences of the coarse weights by
    # 1 / (bin mass): sharp ("trained-like") densities leave most bins nearly empty.  > 97 % within 1e-5 (measured: 98.7 % in
    # the trained-like case, > 99.8 % in the others) and a cap on real bin hops.
    frac5, frac3 = float((dz > 1e-5).float().mean()), float((dz > 1e-3).float().mean())
    assert frac5 < 3e-2 and frac3 < 1e-3, (frac5, frac3, float(dz.max()))
    assert rel_err(keep["z_coarse"].cpu().numpy(), okeep["z_coarse"].numpy()) < 1e-6
    # 2. the oracle at the GPU's fine depths, fp32 and fp64 (its own noise sets the gradient gate)
    st32, l32, r32, _ = oracle_at(c, zf, torch.float32)
    sum(l32.values()).backward()
    g32 = named_grads(st32)
    noise = {}

    def reference_noise():  # fp32 vs fp64 run of the oracle itself: only worked out when a gradient misses the flat gate
        if not noise:
            st64, l64, _, _ = oracle_at(c, zf, torch.float64)
            sum(l64.values()).backward()
            g64 = named_grads(st64)
            noise.update({k: float((a.double() - g64[k]).abs().max() / max(float(g64[k].abs().max()), 1e-30))
                          for k, a in g32.items() if a is not None and g64[k] is not None})
        return noise
    errs = {}
    assert set(res.keys()) == set(r32.keys())
    for k, v in r32.items():
        e = rel_err(res[k].detach().cpu().numpy(), v.detach().numpy())
        if not e < (TOL_W if "weights" in k else TOL_MAP):
            errs[k] = e
    for k, v in l32.items():
        if not abs(float(loss_d[k]) - float(v)) <= TOL_MAP * max(abs(float(v)), 1e-2):
            errs["loss_" + k] = (float(loss_d[k]), float(v))
    assert not errs, errs
    got = dict(sysm.named_parameters())
    bad = {}
    for n, r in g32.items():
        if n.endswith(".progress"):
            continue
        g = got[n].grad
        if r is None:
            if g is not None and float(g.abs().max()) != 0.0:
                bad[n] = "expected no gradient"
            continue
        e = rel_err(g.detach().cpu().numpy(), r.numpy())
        # (the gate of tests/test_hip_parity.py: widened by the reference's own noise, never beyond GRAD_GATE_CAP; listed in the
        # session's widened-gate report under "mid:<case>")
        if not e < TOL_GRAD and not e < grad_gate(reference_noise().get(n, 0.0), "mid:" + name, n, e):
            bad[n] = (e, noise.get(n, 0.0))
    assert not bad, bad


@pytest.mark.parametrize("name", list(CASES))
def test_mid_size_batch_fp16_mode_at_its_stated_gates(name):
    """configs[3] arithmetic at 301 rays: maps 1e-2, per-sample weights 3e-2, every gradient -- table, pose AND NeRF weights --
    relative L2 6e-2 against the oracle at the GPU's fine depths."""
    c = SynthCase(name, RAYS, **CASES[name])
    sysm, loss, loss_d, res, keep = hip_step(c, "f16")
    zf = keep["z_fine"].cpu()
    st, l32, r32, _ = oracle_at(c, zf, torch.float32)
    sum(l32.values()).backward()
    g32 = named_grads(st)
    errs = {}
    for k, v in r32.items():
        e = rel_err(res[k].detach().cpu().numpy(), v.detach().numpy())
        if not e < (3e-2 if "weights" in k else 1e-2):
            errs[k] = e
    assert not errs, errs
    got = dict(sysm.named_parameters())
    bad, n_nerf = {}, 0
    for n, r in g32.items():
        if r is None or n.endswith(".progress"):
            continue
        g = got[n].grad
        assert g is not None, n
        a, b = g.detach().cpu().double().reshape(-1), r.double().reshape(-1)
        if float(b.norm()) == 0.0:
            continue
        l2 = float((a - b).norm() / b.norm())
        n_nerf += n.startswith("nerf_")
        if not l2 < 6e-2:
            bad[n] = l2
    assert n_nerf >= 30, n_nerf  # the NeRF weight gradients were really compared
    assert not bad, bad

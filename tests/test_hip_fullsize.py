"""Full-size checks at BASELINE.json configs[1] (4096 rays, 64 + 128 samples, two 8x256 fields, pose optimisation on), where
the CPU oracle would need minutes and ~20 GB per step: size-independent properties of the path instead.

  * bitwise run-to-run reproducibility of a whole training step (no atomics in any reduction of the path);
  * data-parallel equivalence: the gradient of the 4096-ray batch equals the mean of the gradients of its two 2048-ray
    halves (what two ranks + one all-reduce compute) -- on one GPU, with per-ray uniform draws held fixed;
  * the two arithmetic modes of the field (f16x3 / fp32 MFMA) agree on every output map and loss term;
  * compositing / resampling invariants: weights in [0,1], sum of weights <= 1, fine depths sorted inside [near, far],
    expected depths inside [near, far]."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
R, NC, NF = 4096, 64, 128


def _system(progress=0.3, rays=R, n_images=763):
    import bench
    return bench.build_system(torch.device("cuda", 0), progress, rays, n_images)


def _batch(seed=100, rays=R, n_images=763):
    import bench
    return bench.make_batches(torch.device("cuda", 0), 1, seed, rays, n_images)[0]


def _draws(sysm, seed, rays=R):
    """Explicit uniform draws in render_rays' consumption order for this phase: coarse jitter, then the sample_pdf sets."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    m = sysm.get_schedule_mult(sysm._host_progress)
    n_s = round(m * NF)
    sizes = [NC] + ([NF] if m in (0, 1) else [NF - n_s, n_s])
    return [torch.rand(rays, n, device="cuda", generator=g) for n in sizes]


def _grads(sysm):
    return {n: p.grad.detach().clone() for n, p in sysm.named_parameters() if p.grad is not None}


def _loss_and_grads(sysm, batch, u):
    for p in sysm.parameters():
        p.grad = None
    loss, loss_d, res = sysm.compute_loss(batch, u_list=[t.clone() for t in u])
    loss.backward()
    return loss.detach(), {k: v.detach() for k, v in loss_d.items()}, {k: v.detach() for k, v in res.items()}, _grads(sysm)


def test_training_step_is_bitwise_reproducible():
    sysm, batch = _system(), _batch()
    u = _draws(sysm, 1)
    l1, _, r1, g1 = _loss_and_grads(sysm, batch, u)
    l2, _, r2, g2 = _loss_and_grads(sysm, batch, u)
    assert torch.equal(l1, l2)
    for k in r1:
        assert torch.equal(r1[k], r2[k]), k
    assert g1.keys() == g2.keys() and len(g1) > 60
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


@pytest.mark.parametrize("mode", ["f16x3", "f16"])
def test_soak_200_forward_backward_passes_are_bitwise_identical(mode):
    """Soak for the hazard class DESIGN.md describes (round 1: a few rows in 65 536 came out with a stale value in lanes 48-63
    of one accumulator register, different rows every run): 200 forward + backward passes of the full-size step on the
    same inputs must reproduce the first one bit for bit -- every result map and every gradient (~70 tensors)."""
    from upnerf_amd import rendering as rd
    sysm, batch = _system(), _batch()
    u = _draws(sysm, 4)
    old = rd.FIELD_MODE
    rd.FIELD_MODE = mode
    try:
        l0, _, r0, g0 = _loss_and_grads(sysm, batch, u)
        for it in range(200):
            l, _, r, g = _loss_and_grads(sysm, batch, u)
            bad = [k for k in r0 if not torch.equal(r0[k], r[k])] + [k for k in g0 if not torch.equal(g0[k], g[k])]
            assert torch.equal(l0, l) and not bad, (it, bad[:5])
    finally:
        rd.FIELD_MODE = old


def test_gradient_of_the_batch_is_the_mean_of_the_gradients_of_its_shards():
    sysm, batch = _system(), _batch()
    u = _draws(sysm, 2)
    _, _, _, gfull = _loss_and_grads(sysm, batch, u)
    halves = []
    for lo in (0, R // 2):
        sl = slice(lo, lo + R // 2)
        _, _, _, g = _loss_and_grads(sysm, {k: v[sl] for k, v in batch.items()}, [t[sl] for t in u])
        halves.append(g)
    assert gfull.keys() == halves[0].keys() == halves[1].keys()
    worst = 0.0
    for k in gfull:
        mean = (halves[0][k].double() + halves[1][k].double()) / 2
        scale = float(gfull[k].double().abs().max()) + 1e-30
        err = float((gfull[k].double() - mean).abs().max()) / scale
        worst = max(worst, err)
        assert err < 2e-4, (k, err)  # fp32 summation order differs between one 4096-ray and two 2048-ray reductions
    print("worst relative deviation", worst)


@pytest.mark.parametrize("progress", [0.05, 0.3, 0.8])
def test_field_arithmetic_modes_agree_at_full_size(progress):
    from upnerf_amd import rendering as rd
    sysm, batch = _system(progress), _batch()
    u = _draws(sysm, 3)
    out = {}
    old = rd.FIELD_MODE
    try:
        for mode in ("f32", "f16x3"):
            rd.FIELD_MODE = mode
            keep = {}
            loss, loss_d, res = sysm.compute_loss(batch, u_list=[t.clone() for t in u], keep=keep)
            out[mode] = (loss.detach(), {k: v.detach() for k, v in loss_d.items()}, {k: v.detach() for k, v in res.items()}, keep)
    finally:
        rd.FIELD_MODE = old
    (la, da, ra, ka), (lb, db, rb, kb) = out["f32"], out["f16x3"]
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(la))
    for k in da:
        assert abs(float(da[k]) - float(db[k])) <= 1e-5 * max(abs(float(da[k])), 1e-6), k
    assert torch.equal(ka["z_coarse"], kb["z_coarse"])
    moved = (ka["z_fine"] - kb["z_fine"]).abs() > 1e-5   # resampled depths next to a cdf knot may hop one bin
    assert float(moved.float().mean()) < 1e-3
    same = ~moved.any(1)
    for k in ra:
        a, b = ra[k][same], rb[k][same]
        err = float((a - b).abs().max()) / (float(a.abs().max()) + 1e-30)
        assert err < 1e-4, (k, err)


def test_gradients_of_the_two_fp32_accurate_modes_agree_at_full_size():
    """f16x3 against the fp32-MFMA kernels on GRADIENTS at full size (the maps are compared above): with the resampled
    depths held identical (same draws; rows whose depths hop a cdf knot are a 1e-3 fraction) every parameter gradient
    agrees to 2e-3 of its largest entry -- the per-tile / per-tensor power-of-two exponent machinery of the split sees the
    magnitudes of a real 4096 x 192 batch here, not a synthetic tile."""
    from upnerf_amd import rendering as rd
    sysm, batch = _system(0.3), _batch()
    u = _draws(sysm, 5)
    old = rd.FIELD_MODE
    g = {}
    try:
        for mode in ("f32", "f16x3"):
            rd.FIELD_MODE = mode
            g[mode] = _loss_and_grads(sysm, batch, u)[3]
    finally:
        rd.FIELD_MODE = old
    assert g["f32"].keys() == g["f16x3"].keys() and len(g["f32"]) > 60
    worst = {}
    for k, a in g["f32"].items():
        b = g["f16x3"][k]
        err = float((a.double() - b.double()).abs().max()) / (float(a.double().abs().max()) + 1e-30)
        if not err < 2e-3:
            worst[k] = err
    assert not worst, worst


def test_trevi_shape_f16_mode_full_step():
    """BASELINE.json configs[3] at full size: 8192 rays, 1689-row per-image tables, fp16 field arithmetic ("f16"), one whole
    training-step forward + backward.  Checked: bitwise run-to-run reproducibility; finite results and gradients for every
    parameter incl. the 1689-row tables; the compositing invariants; agreement with the fp32-accurate f16x3 mode on the SAME
    sampled depths at the mode's stated gate (per-ray maps 2e-2 max-normalised, loss 1e-2)."""
    from upnerf_amd import rendering as rd
    RT, NI = 8192, 1689
    sysm, batch = _system(0.3, RT, NI), _batch(7, RT, NI)
    assert sysm.se3_refine.weight.shape == (NI, 6) and sysm.transient_net.embedding_t.weight.shape[0] == NI
    u = _draws(sysm, 9, RT)
    old = rd.FIELD_MODE
    try:
        rd.FIELD_MODE = "f16"
        l1, d1, r1, g1 = _loss_and_grads(sysm, batch, u)
        l2, d2, r2, g2 = _loss_and_grads(sysm, batch, u)
        rd.FIELD_MODE = "f16x3"
        l3, d3, r3, g3 = _loss_and_grads(sysm, batch, u)
    finally:
        rd.FIELD_MODE = old
    assert torch.equal(l1, l2)
    for k in r1:
        assert torch.equal(r1[k], r2[k]), k
    assert g1.keys() == g2.keys() == g3.keys() and len(g1) > 60
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
        assert bool(torch.isfinite(g1[k]).all()), k
    for k, v in r1.items():
        assert v.shape[0] == RT and bool(torch.isfinite(v).all()), k
        if "weights" in k:
            assert bool((v >= 0).all()) and float(v.sum(1).max()) <= 1 + 1e-3, k
    assert abs(float(l1) - float(l3)) <= 1e-2 * max(abs(float(l3)), 1e-2), (float(l1), float(l3))
    for k in r1:
        if "weights" in k:
            continue  # per-sample weights live on depths resampled from each mode's own coarse weights
        err = float((r1[k] - r3[k]).abs().max()) / (float(r3[k].abs().max()) + 1e-30)
        assert err < 2e-2, (k, err)
    for k in ("se3_refine.weight", "depth_scale.weight", "nerf_fine.xyz_encoding_1.0.weight", "nerf_coarse.share_sigma.0.weight"):
        err = float((g1[k] - g3[k]).abs().max()) / (float(g3[k].abs().max()) + 1e-30)
        assert err < 0.15, (k, err)


@pytest.mark.parametrize("progress", [0.05, 0.3, 0.8])
def test_compositing_and_resampling_invariants(progress):
    sysm, batch = _system(progress), _batch()
    keep = {}
    with torch.no_grad():
        loss, loss_d, res = sysm.compute_loss(batch, keep=keep)
    near, far = batch["ray_infos"][:, 0:1], batch["ray_infos"][:, 1:2]
    zf = keep["z_fine"]
    assert zf.shape == (R, NC + NF)
    assert bool((zf[:, 1:] >= zf[:, :-1]).all())
    assert bool((zf >= near - 1e-6).all()) and bool((zf <= far + 1e-6).all())
    for k, v in res.items():
        assert bool(torch.isfinite(v).all()), k
        if "weights" in k:
            assert bool((v >= 0).all()) and bool((v <= 1 + 1e-6).all()), k
            assert float(v.sum(1).max()) <= 1 + 1e-4, k
        if "depth" in k:
            assert bool((v >= -1e-6).all()) and bool((v <= far[:, 0] + 1e-4).all()), k
        if k.startswith("s_rgb") or k.startswith("rgb_"):
            assert bool((v >= -1e-6).all()) and bool((v <= 1 + 1e-5).all()), k
    assert torch.isfinite(loss)


def test_validation_render_is_chunk_invariant_matches_the_training_forward_and_skips_backward_stores():
    """No-gradient inference path (validation / test renders): identical maps to the gradient-enabled forward with
    perturb = 0, identical for any chunk size, and without the activation stores of the backward pass."""
    from upnerf_amd.nerf_system import NeRFSystem
    sysm, batch = _system(0.3), _batch()
    n = 3000  # "image" of 3000 rays: not a multiple of the chunk size
    b = {k: v[:n] for k, v in batch.items()}
    rays = sysm.rays_from_batch(b).detach()
    m = sysm.get_schedule_mult(sysm._host_progress)
    outs = {}
    for chunk in (4096, 1024, 700):
        sysm.hparams["val.chunk_size"] = chunk
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        with torch.no_grad():
            outs[chunk] = sysm(rays, b["feats"], b["img_idx"], m, train=False)
        peak = torch.cuda.max_memory_allocated() - base
        if chunk == 4096:
            peak_nograd = peak
    for chunk in (1024, 700):
        for k in outs[4096]:
            assert torch.equal(outs[4096][k], outs[chunk][k]), (chunk, k)
    # gradient-enabled forward of the same rays, perturb = 0: same kernels with all stores on
    sysm.hparams["val.chunk_size"] = 4096
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    ref = sysm(rays.clone().requires_grad_(True), b["feats"], b["img_idx"], m, train=False)
    peak_grad = torch.cuda.max_memory_allocated() - base
    for k in ref:
        assert torch.equal(ref[k].detach(), outs[4096][k]), k
    assert peak_nograd < 0.45 * peak_grad, (peak_nograd, peak_grad)
    # validation_step on a DataLoader-shaped batch (leading 1)
    vb = {k: v[None] for k, v in b.items()}
    log = sysm.validation_step(vb)
    psnr = -10.0 * torch.log10(((outs[4096]["rgb_fine"] - b["rgbs"]) ** 2).mean())
    assert torch.isfinite(log["val_loss"]) and abs(float(log["val_psnr"]) - float(psnr)) < 1e-4


def test_density_only_coarse_pass_at_full_size():
    """Test-time-optimisation shape at configs[1] sizes on the f16x3 kernels: the density-only coarse pass changes neither
    the fine colour map nor any gradient, bit for bit."""
    from upnerf_amd.rendering import render_rays
    sysm, b = _system(1.0), _batch()
    for m in (sysm.nerf_coarse, sysm.nerf_fine):
        m.encode_candidate = False
    outs = []
    for flag in (False, True):
        for p in sysm.parameters():
            p.grad = None
        res = render_rays(sysm.models, sysm.embeddings, sysm.rays_from_batch(b), b["img_idx"], 1.0, N_samples=NC,
                          perturb=0, N_importance=NF, coarse_sigma_only=flag)
        ((res["s_rgb_fine"] - b["rgbs"]) ** 2).mean().backward()
        outs.append((res, _grads(sysm)))
    (full, g_full), (lean, g_lean) = outs
    assert "s_rgb_coarse" in full and "s_rgb_coarse" not in lean
    for k in lean:
        assert torch.equal(lean[k], full[k]), k
    assert g_full.keys() == g_lean.keys() and len(g_lean) > 20
    for k in g_full:
        assert torch.equal(g_full[k], g_lean[k]), k


# ---- the full-size HIP path against the ORACLE on a slice of its rays -------------------------------------
class _Slice:
    """`n` seeded rays of a SynthCase (tests/test_hip_midsize.py): same closed-form weights and tables, the batch rows, uniform
    draws and fine depths of those rays only.  Rays are independent given the weights (rendering.py:53-314 has no cross-ray
    term; the loss is a mean over rays, losses.py:21-64), so the oracle on the slice is the oracle on the batch, ray by ray."""

    def __init__(self, case, idx):
        self.__dict__.update(case.__dict__)
        self._case, self._idx, self.R = case, idx, len(idx)
        self.u_list = [u[idx] for u in case.u_list]

    def __getattr__(self, name):
        return getattr(self._case, name)

    def batch(self, dtype=torch.float32):
        return {k: v[self._idx] for k, v in self._case.batch(dtype).items()}


@pytest.mark.parametrize("mode,rays,n_img,progress", [("f16x3", 4096, 763, 0.3), ("f16x3", 4096, 763, 0.05), ("f16", 8192, 1689, 0.3)])
def test_full_size_forward_and_ray_gradients_match_the_oracle_on_a_slice(mode, rays, n_img, progress):
    """The HIP path at BASELINE.json's full batch sizes (configs[1]: 4096 rays, f16x3; configs[3]: 8192 rays / 1689 images,
    f16) against oracle.training_forward -- the pinned restatement of models/rendering.py:53-314 -- on 64 seeded rays of the
    batch: same weights, the rays' own uniform draws, the fine pass evaluated at the GPU's fine depths of those rays (the
    resampling itself: sample_pdf tests and the mid-size protocol).  Every per-ray map at 1e-4 (f16: 1e-2), per-sample weights
    at 2e-4 (f16: 3e-2), and d loss / d rays of those rays against the oracle's (per-ray terms: the batch mean's 1 / R against
    the slice mean's 1 / 64 is the only difference) at max(1e-3, 4 x the oracle's fp32-vs-fp64 noise) capped at 2e-2 (f16:
    relative L2 6e-2).  Before round 5 the largest oracle comparison was 301 rays; at full size the f16x3 kernels were only
    compared with the fp32-MFMA kernels."""
    from test_hip_midsize import SynthCase, oracle_at
    from test_hip_parity import GRAD_GATE_CAP, TOL_GRAD, TOL_MAP, TOL_W, build_system
    from golden_util import rel_err
    from upnerf_amd import rendering as rd
    from upnerf_amd.rendering import ray_gradient, retain_ray_gradient
    c = SynthCase(f"full_{mode}_{progress}", rays, progress, seed=21, n_img=n_img)
    sysm = build_system(c)
    batch = {k: v.cuda() for k, v in c.batch().items()}
    old = rd.FIELD_MODE
    rd.FIELD_MODE = mode
    try:
        keep = {}
        loss, loss_d, res = sysm.compute_loss(batch, u_list=[u.clone() for u in c.u_list], keep=keep)
        retain_ray_gradient(sysm._last_rays)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        rd.FIELD_MODE = old
    n = 64
    idx = torch.sort(torch.randperm(rays, generator=torch.Generator().manual_seed(5))[:n])[0]
    sl = _Slice(c, idx)
    zf = keep["z_fine"].cpu()[idx]
    out = {}
    for dt in (torch.float32, torch.float64):
        st, losses, r, okeep = oracle_at(sl, zf, dt)
        sum(losses.values()).backward()
        out[dt] = (r, okeep["rays"].grad.detach().clone(), okeep["z_coarse"].detach())
    r32, gr32, zc32 = out[torch.float32]
    assert rel_err(keep["z_coarse"].cpu()[idx].numpy(), zc32.numpy()) < 1e-6
    tol_map, tol_w = (TOL_MAP, TOL_W) if mode == "f16x3" else (1e-2, 3e-2)
    errs = {}
    assert set(res.keys()) == set(r32.keys())
    for k, v in r32.items():
        e = rel_err(res[k].detach().cpu()[idx].numpy(), v.detach().numpy())
        if not e < (tol_w if "weights" in k else tol_map):
            errs[k] = e
    assert not errs, errs
    got = ray_gradient(sysm._last_rays).detach().cpu()[idx].double() * (rays / n)  # batch mean -> slice mean
    for tag, cs in (("rays_o", slice(0, 3)), ("rays_d", slice(3, 6))):
        a, b = got[:, cs], gr32[:, cs].double()
        if mode == "f16x3":
            noise = float((b - out[torch.float64][1][:, cs]).abs().max() / out[torch.float64][1][:, cs].abs().max())
            e = float((a - b).abs().max() / b.abs().max())
            assert e < max(TOL_GRAD, min(4 * noise, GRAD_GATE_CAP)), (tag, e, noise)
        else:
            e = float((a - b).norm() / b.norm())
            assert e < 6e-2, (tag, e)


def test_full_size_parameter_gradients_match_the_oracle():
    """VERDICT r5 weak spot 1d: the PARAMETER gradients of one full configs[1] step (4096 rays, 64 + 128 samples, 763 images,
    phase 1, f16x3) against the oracle directly -- until round 6 they were covered at full size only transitively (mean of shards
    + the 301-ray oracle comparison).  The oracle (pinned restatement of models/rendering.py:53-314, losses.py:21-64) is run over
    the batch in eight shards of 512 rays at the GPU's fine depths -- every loss term is a mean over rays, so the batch gradient
    is the ray-count-weighted mean of the shard gradients (the property test_gradient_of_the_batch_is_the_mean_... holds the HIP
    path to; checked on the oracle itself in fp64 to 6e-14) -- in fp32.  Gate: that of the golden and mid-size tests,
    max(1e-3, min(4 x noise, 3e-2)), noise = the oracle's own fp32-vs-fp64 difference on that tensor.  The fp64 pass costs ~130 s
    of host time, so the noise of the two parameters known to need it is DATA (tests/golden/parity_widened_pinned.json,
    "full_noise", measured by this test with UPNERF_FULL_NOISE=1) and the fp64 oracle runs only when asked for or when another
    parameter misses the flat 1e-3 -- which then fails against the pinned list like any new widening.  Widenings are reported
    under "full:phase1"."""
    import json
    import os
    from test_hip_midsize import SynthCase, hip_step, oracle_at
    from test_hip_parity import TOL_GRAD, grad_gate
    from golden_util import named_grads, rel_err
    rays, shard = 4096, 512
    c = SynthCase("full_params", rays, 0.3, seed=21, n_img=763)
    sysm, loss, loss_d, res, keep = hip_step(c, "f16x3")
    zf = keep["z_fine"].cpu()

    def oracle_grads(dt):
        tot, loss_sum = {}, 0.0
        for s in range(0, rays, shard):
            idx = torch.arange(s, min(s + shard, rays))
            sl = _Slice(c, idx)
            st, losses, _, _ = oracle_at(sl, zf[idx], dt)
            total = sum(losses.values())
            total.backward()
            w = len(idx) / rays
            loss_sum += float(total.detach()) * w
            for n, g in named_grads(st).items():
                if g is not None:
                    tot[n] = g.double() * w if tot.get(n) is None else tot[n] + g.double() * w
                else:
                    tot.setdefault(n, None)
        return tot, loss_sum

    g32, l32 = oracle_grads(torch.float32)
    assert abs(float(loss) - l32) <= 1e-4 * max(abs(l32), 1e-2), (float(loss), l32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_widened_pinned.json")
    pinned_noise = json.load(open(path)).get("full_noise", {})
    measured = {}

    def reference_noise(n):
        if os.environ.get("UPNERF_FULL_NOISE") != "1" and n in pinned_noise:
            return pinned_noise[n]
        if not measured:
            g64, _ = oracle_grads(torch.float64)
            measured.update({k: float((a - g64[k]).abs().max() / max(float(g64[k].abs().max()), 1e-30)) for k, a in g32.items()
                             if a is not None and g64[k] is not None})
            print("[full_noise] " + json.dumps({k: v for k, v in measured.items() if k in pinned_noise or v > 2.5e-4}))
        return measured.get(n, 0.0)
    got = dict(sysm.named_parameters())
    bad, compared = {}, 0
    for n, r in g32.items():
        if n.endswith(".progress"):
            continue
        g = got[n].grad
        if r is None:
            if g is not None and float(g.abs().max()) != 0.0:
                bad[n] = "expected no gradient"
            continue
        compared += 1
        e = rel_err(g.detach().cpu().double().numpy(), r.numpy())
        if not e < TOL_GRAD:
            nz = reference_noise(n)
            if not e < grad_gate(nz, "full:phase1", n, e):
                bad[n] = (e, nz)
    assert compared >= 60, compared
    assert not bad, bad

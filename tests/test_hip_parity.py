"""End-to-end GPU parity: one full training-step forward + backward of the HIP path (NeRFSystem.compute_loss:
pose refinement -> rays -> render_rays coarse/resample/fine -> TransientNet -> UPNeRFLoss -> backward) against the
golden vectors the real reference produced (tests/golden/*.npz), on the same weights, batch and uniform draws.

Bar (BASELINE.json): outputs within 1e-4 relative of the reference in fp32.  Gates used here, max-normalised:
per-ray maps and loss terms 1e-4; per-sample weights 2e-4 (the reference's own fp32-vs-fp64 noise on
per-sample fine weights is 4e-5, because 1e-7 cdf differences move fine samples); gradients 1e-3 of the tensor's
largest entry (they inherit the resampling sensitivity)."""
import os

import numpy as np
import pytest
import torch

from golden_util import CASES, Case, rel_err


def retain_ray_gradient(rays):
    from upnerf_amd.rendering import retain_ray_gradient as f
    f(rays)


def ray_gradient(rays):
    from upnerf_amd.rendering import ray_gradient as f
    return f(rays)

pytestmark = pytest.mark.gpu

TOL_MAP, TOL_W, TOL_GRAD = 1e-4, 2e-4, 1e-3
# r4 ADVICE: the gradient gate widens with the reference's OWN noise (4 x its fp32-vs-fp64 / one-ulp-input spread), but never
# beyond this cap -- a parameter whose reference gradient is noisier than cap / 4 is compared AT the cap, and every
# parameter that needed more than the flat 1e-3 is listed in the test's output (`[widened]`), so a drift shows.
# (3e-2: yaml_phase2's nerf_fine.xyz_encoding_1.0.weight differs by 2.2e-2 where the reference's own noise is 1.1e-2 -- all ten
# encoding bands on at 128 + 128 samples, one ulp of a ray direction flips a ReLU: DESIGN.md section 6)
GRAD_GATE_CAP = 3e-2
# case -> parameter -> (gate, measured error) of every comparison whose gate is above the flat 1e-3.  Written to
# $UPNERF_WIDENED_OUT (default gpurun_out/parity_widened.txt) at session end by tests/conftest.py -- the per-round copy is
# profiles/rNN_parity_widened.txt -- and the subset that NEEDED its widening (error above the flat gate) is held to the pinned
# list tests/golden/parity_widened_pinned.json by test_only_the_pinned_parameters_need_a_widened_gate (r5 ADVICE: a new
# widening fails instead of just printing).
_WIDENED = {}


def grad_gate(noise_n, case=None, name=None, err=None):
    g = max(TOL_GRAD, min(4.0 * noise_n, GRAD_GATE_CAP))
    if g > TOL_GRAD and case is not None:
        _WIDENED.setdefault(case, {})[name] = (g, err)
    return g


def widened_lines():
    out = []
    for case in sorted(_WIDENED):
        for n, (g, e) in sorted(_WIDENED[case].items()):
            need = e is not None and e >= TOL_GRAD
            out.append(f"{case}\t{n}\tgate {g:.2e}\terr {'-' if e is None else format(e, '.2e')}\t{'NEEDED' if need else 'within the flat gate'}")
    return out


def widened_needed():
    return {f"{case}:{n}" for case, d in _WIDENED.items() for n, (g, e) in d.items() if e is not None and e >= TOL_GRAD}
# Goldens in which at least one resampled depth of the GPU run sits an eps-bin away from the reference's (see
# test_training_step_matches_reference_golden): those are compared loosely with the golden gradients and strictly with the
# oracle at the GPU's own depths.  The set is PINNED: a change that makes another case flip fails
# test_only_the_known_goldens_take_the_flipped_branch instead of quietly moving it to the loose gate.
KNOWN_FLIPPED = frozenset({"cfg1_small_fine", "cfg2_det_phase1", "cfg2_phase0", "cfg2_phase1", "cfg2_phase2", "cfg2_trained_p045",
                           "cfg2_trained_p08", "small_nocand",  # round 3, both field tilings
                           # round 4's cases at the reference's 128 + 128 samples: new here, and held to the strict gate at the
                           # reference's own depths by test_training_step_matches_reference_golden_at_the_reference_depths
                           "yaml_phase0", "yaml_phase1",
                           # round 5's cases without the feature head (encode_feat = False): new here, strict at the reference's
                           # own depths by the same test
                           "nofeat_phase0", "nofeat_phase2", "nofeat_w256_phase1"})
_FLIPPED_SEEN = {}


def oracle_grads(c, dtype, z_fine=None, ulp_seed=None):
    """Gradients of the oracle (pinned to the reference).  ulp_seed: the ray directions of the batch are moved by at most one
    fp32 ulp (a seeded relative perturbation of 1.2e-7) -- the reference's own conditioning with respect to the last bit of its
    inputs: with all ten encoding bands on, one ulp of a direction moves sin(2^9 pi x) by 1e-3 and can flip a ReLU."""
    from golden_util import named_grads, orc
    st = c.state(dtype=dtype)
    real = orc.schedule_mult
    orc.schedule_mult = lambda p, s: c.sched
    keep = {}
    batch = c.batch(dtype)
    if ulp_seed is not None:
        g = torch.Generator().manual_seed(ulp_seed)
        d = batch["directions"]
        batch["directions"] = d * (1 + 1.2e-7 * (torch.rand(d.shape, generator=g) * 2 - 1)).to(dtype)
    try:
        losses, _ = orc.training_forward(st, c.cfgs(), batch, c.hparams(), c.progress,
                                         u_list=[u.to(dtype) for u in c.u_list], keep=keep,
                                         z_fine_override=None if z_fine is None else z_fine.to(dtype))
    finally:
        orc.schedule_mult = real
    sum(losses.values()).backward()
    return named_grads(st), keep


def reference_fp32_noise(c):
    """Per-tensor gradient noise of the reference's OWN arithmetic: oracle in fp32 vs the same oracle in fp64
    (max-normalised).  With all encoding bands active (2^9 pi x) a 1e-7 shift of a resampled depth changes
    sin/cos by 1e-3, so the reference's fp32 gradients are themselves only good to ~4e-3 in those phases
    (cfg2_phase2, small_tto); where the arithmetic is well conditioned the noise is ~1e-6 and the flat gate applies.
    The gradient gate below is max(TOL_GRAD, 4 x this noise)."""
    g32, keep = oracle_grads(c, torch.float32)
    g64, _ = oracle_grads(c, torch.float64)
    noise = {}
    for k, a in g32.items():
        b = g64[k]
        if a is not None and b is not None:
            noise[k] = float((a.double() - b).abs().max() / max(float(b.abs().max()), 1e-30))
    return noise, keep


def build_system(c):
    from upnerf_amd.nerf_system import NeRFSystem, SyntheticDataset, default_hparams
    hp = default_hparams(**{"nerf.N_samples": c.Nc, "nerf.N_importance": c.Nf, "nerf.use_disp": c.use_disp,
                            "nerf.perturb": c.perturb, "pose.optimize": c.pose_opt, "pose.c2f": c.c2f,
                            "nerf.D": c.D, "nerf.W": c.W, "max_steps": 1000,
                            "nerf.feat_dim": 384 if getattr(c, "encode_feat", True) else 0})
    sysm = NeRFSystem(hp, SyntheticDataset(c.n_img))
    sysm.setup()
    st = c.state(requires_grad=False)
    sd = {}
    for typ in ("coarse", "fine") if c.fine else ("coarse",):
        for k, v in st[f"nerf_{typ}"].items():
            sd[f"nerf_{typ}.{k}"] = v
        sd[f"nerf_{typ}.progress"] = torch.tensor(c.progress)
    for k, v in st["transient_net"].items():
        sd[f"transient_net.{k}"] = v
    for k, v in st.items():
        if not isinstance(v, dict):
            sd[f"{k}.weight"] = v
    missing, unexpected = sysm.load_state_dict(sd, strict=True)
    sysm.cuda()
    sysm.set_progress(c.progress)
    if c.encode_candidate is False:
        for m in sysm.models.values():
            if hasattr(m, "encode_candidate"):
                m.encode_candidate = False
    if "cfg_sched" in c.g:
        sysm.get_schedule_mult = lambda p: c.sched
    return sysm


@pytest.mark.parametrize("name", ["cfg2_phase0", "cfg2_phase1", "cfg2_phase2", "cfg2_det_phase1", "nofeat_w256_phase1"])
def test_f16_mode_matches_reference_golden_at_its_stated_gate(name):
    """BASELINE.json configs[3] arithmetic (FIELD_MODE "f16": fp16 MLP weights / activations on MFMA, fp32 accumulate) on the
    config-#2-shaped goldens, all three schedule phases.  Stated gates (SURVEY 8d "Config 4": ~1e-2 on maps), max-normalised:
    per-ray maps and loss terms 1e-2, per-sample weights 3e-2 (the fine depths are resampled from fp16-accurate coarse
    weights), table / pose gradients 1e-1 of the tensor's largest entry."""
    from upnerf_amd import rendering
    c = Case(name)
    sysm = build_system(c)
    batch = {k: v.cuda() for k, v in c.batch().items()}
    old = rendering.FIELD_MODE
    rendering.FIELD_MODE = "f16"
    try:
        keep = {}
        loss, loss_d, res = sysm.compute_loss(batch, u_list=[u.clone() for u in c.u_list], keep=keep)
        loss.backward()
    finally:
        rendering.FIELD_MODE = old
    exp = c.expected_results()
    assert set(res.keys()) == set(exp.keys())
    errs = {}
    for k, v in exp.items():
        tol = 3e-2 if "weights" in k else 1e-2
        e = rel_err(res[k].detach().cpu().numpy().reshape(v.shape), v)
        if not e < tol:
            errs[k] = e
    assert not errs, errs
    el = c.expected_losses()
    assert abs(float(loss) - float(el["total"])) <= 1e-2 * max(abs(float(el["total"])), 1e-2), (float(loss), float(el["total"]))
    eg = c.expected_grads()
    bad = {}
    for n in ("se3_refine.weight", "depth_scale.weight", "embedding_fine_a.weight", "embedding_coarse_c.weight"):
        g = dict(sysm.named_parameters())[n].grad
        if n not in eg or eg[n] is None or g is None:
            continue
        ref, stride, _sums = eg[n]
        got = g.detach().cpu().reshape(-1)[::(stride or 1)][: ref.size].numpy()
        e = rel_err(got, ref)
        if not e < 1e-1:
            bad[n] = e
    assert not bad, bad


@pytest.mark.parametrize("name", ["cfg2_phase0", "cfg2_phase1", "cfg2_phase2", "cfg2_trained_p045"])
def test_fp16_stored_weight_gradient_operands_option(name):
    """rendering.WGRAD_STORE = "f16" (f16x3 mode): activations / gradients travel to the weight-gradient kernels as fp16 tiles.
    The forward pass and the data-gradient chain are untouched -- every render_rays output, the loss and the gradients that
    do not pass through a trunk weight gradient (pose, depth scale, per-image tables) are BITWISE those of the default -- and
    the trunk weight gradients stay within 2e-3 (max-normalised) of the fp32-accurate ones."""
    from upnerf_amd import rendering
    c = Case(name)
    out = {}
    old = rendering.WGRAD_STORE
    try:
        for mode in ("f32", "f16"):
            rendering.WGRAD_STORE = mode
            sysm = build_system(c)
            batch = {k: v.cuda() for k, v in c.batch().items()}
            loss, loss_d, res = sysm.compute_loss(batch, u_list=[u.clone() for u in c.u_list])
            loss.backward()
            out[mode] = (loss.detach().clone(), {k: v.detach().clone() for k, v in res.items()},
                         {n: p.grad.detach().clone() for n, p in sysm.named_parameters() if p.grad is not None})
    finally:
        rendering.WGRAD_STORE = old
    (la, ra, ga), (lb, rb, gb) = out["f32"], out["f16"]
    assert torch.equal(la, lb)
    for k in ra:
        assert torch.equal(ra[k], rb[k]), k
    assert ga.keys() == gb.keys()
    worst = 0.0
    for n in ga:
        trunk = ".xyz_encoding_" in n and "final" not in n
        if not trunk:
            assert torch.equal(ga[n], gb[n]), n
        else:
            err = float((ga[n] - gb[n]).abs().max()) / (float(ga[n].abs().max()) + 1e-30)
            worst = max(worst, err)
            assert err < 2e-3, (n, err)
    assert worst > 0.0  # the option is really in effect


@pytest.mark.parametrize("name", ["cfg2_phase0", "cfg2_phase1", "cfg2_phase2", "cfg2_trained_p045"])
def test_24bit_stored_weight_gradient_operands_option(name):
    """rendering.WGRAD_STORE = "f24" (f16x3 mode): the operands of the trunk weight gradients are stored as fp16 + a residual
    byte (hi + lo to 2^-20 of the tile's maximum, 3 bytes per element instead of 4) and contracted with three MFMAs per block.
    Forward pass and data-gradient chain untouched (bitwise), trunk weight gradients within 2e-5 (max-normalised) of the
    fp32-stored ones -- two orders below the fp16 option's 2e-3 and below the reference's own fp32-vs-fp64 gradient noise."""
    from upnerf_amd import rendering
    c = Case(name)
    out = {}
    old = rendering.WGRAD_STORE
    try:
        for mode in ("f32", "f24"):
            rendering.WGRAD_STORE = mode
            sysm = build_system(c)
            batch = {k: v.cuda() for k, v in c.batch().items()}
            loss, loss_d, res = sysm.compute_loss(batch, u_list=[u.clone() for u in c.u_list])
            loss.backward()
            out[mode] = (loss.detach().clone(), {k: v.detach().clone() for k, v in res.items()},
                         {n: p.grad.detach().clone() for n, p in sysm.named_parameters() if p.grad is not None})
    finally:
        rendering.WGRAD_STORE = old
    (la, ra, ga), (lb, rb, gb) = out["f32"], out["f24"]
    assert torch.equal(la, lb)
    for k in ra:
        assert torch.equal(ra[k], rb[k]), k
    assert ga.keys() == gb.keys()
    worst = 0.0
    for n in ga:
        trunk = ".xyz_encoding_" in n and "final" not in n
        if not trunk:
            assert torch.equal(ga[n], gb[n]), n
        else:
            err = float((ga[n] - gb[n]).abs().max()) / (float(ga[n].abs().max()) + 1e-30)
            worst = max(worst, err)
            assert err < 2e-5, (n, err)
    assert worst > 0.0  # the option is really in effect
    print(f"{name}: worst trunk weight-gradient difference {worst:.2e}")


@pytest.mark.parametrize("name", ["cfg2_phase2", "cfg2_phase1"])
def test_joined_rays_and_plain_rows_render_the_same_step(name, monkeypatch):
    """nerf_system.rays_from_batch hands render_rays the tensors the [R][8] rows were concatenated from (rendering.join_rays:
    no slice copies forward, no scatter / add / copy chain backward).  Same values forward, bit for bit; the gradient reaches
    the pose table along a shorter chain of the same sums; and a fill-per-buffer step equals the pooled-fill one
    (zero_pool.py) the same way."""
    if name not in CASES:
        pytest.skip("fixture not present")
    import upnerf_amd.nerf_system as ns
    from upnerf_amd import ops, zero_pool
    c = Case(name)
    outs = []
    for plain in (False, True):
        if plain:
            monkeypatch.setattr(ns, "join_rays", lambda o, d, nf: torch.cat([o, d, nf], 1))
            monkeypatch.setattr(zero_pool, "zeros", lambda n, device: torch.zeros(int(n), device=device))
            monkeypatch.setattr(ops, "ENABLE_EMBED_PREFETCH", False)  # ... and an index_select per table
            import upnerf_amd.rendering as rd_
            monkeypatch.setattr(rd_, "FUSE_RESAMPLE", 0)  # ... and a launch per piece of the fine-depth resampling
        sysm = build_system(c)
        batch = {k: v.cuda() for k, v in c.batch().items()}
        for _ in range(2):  # the second step is the one served from the pool
            loss, loss_d = sysm._step_backward(batch, u_list=[u.clone() for u in c.u_list])
        assert (getattr(sysm._last_rays, "_upnerf_parts", None) is None) == plain
        outs.append((loss.detach().clone(), {n: p.grad.detach().clone() for n, p in sysm.named_parameters() if p.grad is not None}))
    (la, ga), (lb, gb) = outs
    assert torch.equal(la, lb)
    assert ga.keys() == gb.keys()
    for n in ga:
        if "se3_refine" in n:
            assert float((ga[n] - gb[n]).abs().max()) <= 1e-6 * float(gb[n].abs().max()), n
        else:
            assert torch.equal(ga[n], gb[n]), n


@pytest.mark.parametrize("name", ["cfg2_phase0", "cfg2_phase1", "cfg2_phase2"])
def test_keyed_draws_generated_by_their_consumers_are_the_draws_of_the_uniform_kernel(name, monkeypatch):
    """Round 6: with keyed jitter (the training default) the coarse-depth kernel and the fused resample + sort kernel generate
    their uniforms themselves.  Same depths, same loss, bit for bit, as the sequence that drew them with upnerf_uniform_keyed."""
    import upnerf_amd.rendering as rd_
    c = Case(name)
    outs = []
    for fuse in (1, 0):
        monkeypatch.setattr(rd_, "FUSE_RESAMPLE", fuse)
        sysm = build_system(c)
        assert sysm.hparams.get("rng.keyed", True) and sysm.hparams["nerf.perturb"] > 0
        sysm.global_step = 7
        batch = {k: v.cuda() for k, v in c.batch().items()}
        keep = {}
        loss, _, _ = sysm.compute_loss(batch, keep=keep)
        outs.append((loss.detach().clone(), keep["z_coarse"].clone(), keep["z_fine"].clone()))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][0], outs[1][0])
    assert float((outs[0][1][:, 1:] - outs[0][1][:, :-1]).min()) >= 0.0 and float(outs[0][2].std()) > 0.0


def test_gradients_of_consecutive_eager_steps_do_not_alias():
    """r5 ADVICE (zero_pool lifetime): parameter gradients may be views of a step's zero arena; a gradient kept from step k must not
    be touched by step k + 1 (fresh arena per step)."""
    c = Case("cfg2_phase1")
    sysm = build_system(c)
    batch = {k: v.cuda() for k, v in c.batch().items()}
    sysm._step_backward(batch, u_list=[u.clone() for u in c.u_list])
    kept = {n: p.grad for n, p in sysm.named_parameters() if p.grad is not None}
    copies = {n: g.clone() for n, g in kept.items()}
    for p in sysm.parameters():
        p.grad = None
    sysm._step_backward(batch, u_list=[u.clone() for u in c.u_list])
    torch.cuda.synchronize()
    new = {n: p.grad for n, p in sysm.named_parameters() if p.grad is not None}
    for n, g in kept.items():
        assert torch.equal(g, copies[n]), n  # untouched by the second step
        if n in new:
            assert new[n].data_ptr() != g.data_ptr(), n


def test_an_in_place_edit_of_joined_rays_is_rendered():
    """r5 ADVICE: join_rays remembers the tensors the rows were concatenated from, render_rays reads those -- but only while
    nobody has written to the rows since (the tensor's version counter).  After `rays[:, 6:8] = ...` (a near / far rescale, the
    edit a caller of the reference's render_rays(rays=...) may make) the rows themselves are rendered."""
    from upnerf_amd.rendering import join_rays, render_rays
    c = Case("cfg2_phase2")
    sysm = build_system(c)
    batch = {k: v.cuda() for k, v in c.batch().items()}
    kw = dict(models=sysm.models, embeddings=sysm.embeddings, img_idx=batch["img_idx"], sched_mult=1.0, N_samples=c.Nc,
              perturb=0, N_importance=c.Nf)
    with torch.no_grad():
        rays = sysm.rays_from_batch(batch)
        assert getattr(rays, "_upnerf_parts", None) is not None
        a = render_rays(rays=rays, **kw)["s_depth_fine"].clone()
        edited = rays.clone()                       # a plain tensor with the edit: what the rows say
        edited[:, 6:8] = edited[:, 6:8] * torch.tensor([1.5, 0.8], device=rays.device)
        want = render_rays(rays=edited, **kw)["s_depth_fine"].clone()
        rays[:, 6:8] = rays[:, 6:8] * torch.tensor([1.5, 0.8], device=rays.device)   # the same edit, in place, on the joined tensor
        got = render_rays(rays=rays, **kw)["s_depth_fine"]
    assert torch.equal(got, want) and not torch.equal(got, a)


@pytest.mark.parametrize("name", CASES)
def test_training_step_matches_reference_golden(name):
    c = Case(name)
    sysm = build_system(c)
    batch = {k: v.cuda() for k, v in c.batch().items()}
    keep = {}
    loss, loss_d, res = sysm.compute_loss(batch, u_list=[u.clone() for u in c.u_list], keep=keep)
    exp = c.expected_results()
    assert set(res.keys()) == set(exp.keys()), (sorted(res.keys()), sorted(exp.keys()))
    errs = {}
    for k, v in exp.items():
        tol = TOL_W if "weights" in k else TOL_MAP
        e = rel_err(res[k].detach().cpu().numpy().reshape(v.shape), v)
        if e >= tol:
            errs[k] = e
    assert not errs, errs
    el = c.expected_losses()
    for k, v in loss_d.items():
        assert abs(float(v) - float(el[k])) <= TOL_MAP * max(abs(float(el[k])), 1e-2), (k, float(v), float(el[k]))
    assert abs(float(loss) - float(el["total"])) <= TOL_MAP * max(abs(float(el["total"])), 1e-2)
    if sysm._last_rays.requires_grad:
        retain_ray_gradient(sysm._last_rays)
    loss.backward()
    got = {n: p.grad for n, p in sysm.named_parameters()}
    noise, okeep = reference_fp32_noise(c)
    worst_noise = max(noise.values()) if noise else 0.0

    # The reference's inverse-CDF rule `denom < eps -> 1` (rendering.py:44-46) makes the resampled depth of a draw
    # that lands in a bin of mass ~eps jump by a bin width under 1e-7 perturbations of the coarse weights.  When that
    # happened (GPU depths != reference depths for some sample) the golden gradients are compared loosely and the
    # strict comparison is made against the oracle evaluated at the GPU's own fine depths.
    flipped = False
    if c.fine:
        dz = (keep["z_fine"].cpu() - okeep["z_fine"]).abs()
        assert float((dz < 1e-5).float().mean()) > 0.99, "resampled depths disagree beyond isolated eps-bin flips"
        flipped = bool(dz.max() > 1e-5)
    _FLIPPED_SEEN[name] = flipped
    print(f"[flipped] {name}: {flipped}" + (f" (max |dz| {float(dz.max()):.2e}, {float((dz > 1e-5).float().mean()):.2%} beyond 1e-5)" if c.fine else ""))
    bad = {}

    def gate(n, err=None):
        return grad_gate(noise.get(n, worst_noise), name, n, err) if not flipped else 5e-2

    if sysm._last_rays.requires_grad:
        gr, er = ray_gradient(sysm._last_rays).cpu().numpy(), c.g["grad_rays"]
        for tag, sl in (("rays_o", slice(0, 3)), ("rays_d", slice(3, 6))):
            e = rel_err(gr[:, sl], er[:, sl])
            # flipped: one resampled depth sits a bin away from the reference's; with sharp ("trained-like") densities a
            # single fine sample can carry a fifth of a ray's gradient.  Loose here, strict below at the GPU's own depths
            # (se3_refine's gradient is the ray gradient pushed through the pose).
            if e >= (grad_gate(worst_noise, name, "grad_" + tag, e) if not flipped else 0.25):
                bad["grad_" + tag] = e
    for n, e in c.expected_grads().items():
        if n.endswith(".progress"):
            continue
        g = got[n]
        if e is None:
            if g is not None and float(g.abs().max()) != 0.0:
                bad[n] = "expected no gradient"
            continue
        vals, stride, sums = e
        if g is None:
            if sums[1] > 0:
                bad[n] = "missing gradient"
            continue
        flat = g.detach().reshape(-1).cpu()
        sub = (flat[::stride] if stride else flat).numpy()[: len(vals)]
        scale = max(float(np.abs(vals).max()), sums[1] / flat.numel(), 1e-12)
        err = float(np.abs(sub - vals).max()) / scale
        if err >= gate(n, err):
            bad[n] = (err, noise.get(n, 0.0))
    assert not bad, ("vs reference golden", flipped, bad)
    if flipped:
        ref, _ = oracle_grads(c, torch.float32, z_fine=keep["z_fine"].cpu())
        for n, r in ref.items():
            if r is None or n.endswith(".progress"):
                continue
            g = got[n.replace("embedding_", "embedding_") if n in got else n]
            err = rel_err(g.detach().cpu().numpy(), r.numpy())
            if err >= grad_gate(noise.get(n, worst_noise), name, n, err):
                bad[n] = (err, noise.get(n, 0.0))
        assert not bad, ("vs oracle at the GPU's fine depths", bad)
    if name in _WIDENED:
        print(f"[widened] {name}: " + ", ".join(f"{n} {g:.1e}" for n, (g, _e) in sorted(_WIDENED[name].items())))


@pytest.mark.parametrize("name", [n for n in CASES if Case(n).fine])
def test_training_step_matches_reference_golden_at_the_reference_depths(name):
    """strict reference-vs-HIP comparison of EVERY golden gradient, with no loose branch.  The fine pass
    of the HIP path is evaluated at the reference's own fine depths (golden key `z_fine`, injected through render_rays'
    test-only `z_fine` argument; the resampling itself carries no gradient, rendering.py:271-306, and stays pinned by the
    test above and by test_sample_pdf_*).  Gates: maps / loss terms 1e-4, per-sample weights 2e-4, every parameter and ray
    gradient max(1e-3, 4 x the reference's own noise at the same depths: fp32 against fp64 arithmetic and fp32 arithmetic on
    inputs moved by one ulp)."""
    c = Case(name)
    sysm = build_system(c)
    batch = {k: v.cuda() for k, v in c.batch().items()}
    keep = {}
    loss, loss_d, res = sysm.compute_loss(batch, u_list=[u.clone() for u in c.u_list], keep=keep, z_fine=c.z_fine.cuda())
    assert torch.equal(keep["z_fine"].cpu(), c.z_fine)
    exp = c.expected_results()
    assert set(res.keys()) == set(exp.keys())
    errs = {k: e for k, v in exp.items()
            if (e := rel_err(res[k].detach().cpu().numpy().reshape(v.shape), v)) >= (TOL_W if "weights" in k else TOL_MAP)}
    assert not errs, errs
    el = c.expected_losses()
    for k, v in loss_d.items():
        assert abs(float(v) - float(el[k])) <= TOL_MAP * max(abs(float(el[k])), 1e-2), (k, float(v), float(el[k]))
    if sysm._last_rays.requires_grad:
        retain_ray_gradient(sysm._last_rays)
    loss.backward()
    got = {n: p.grad for n, p in sysm.named_parameters()}
    # the reference's own noise on every gradient: its fp32 arithmetic against fp64, and its fp32 arithmetic on inputs moved by
    # one ulp (two seeds) -- all at the reference's fine depths.  (yaml_phase2: one ulp of the ray directions changes
    # nerf_fine.xyz_encoding_1.0.bias by 1e-2 of its maximum on the reference itself -- a single ReLU decides the other way.)
    g32, _ = oracle_grads(c, torch.float32, z_fine=c.z_fine)
    others = [oracle_grads(c, torch.float64, z_fine=c.z_fine)[0]] + [oracle_grads(c, torch.float32, z_fine=c.z_fine, ulp_seed=s)[0] for s in (0, 1)]
    noise = {k: max(float((a.double() - o[k].double()).abs().max() / max(float(o[k].abs().max()), 1e-30)) for o in others)
             for k, a in g32.items() if a is not None and all(o[k] is not None for o in others)}
    worst_noise = max(noise.values()) if noise else 0.0
    bad = {}
    if sysm._last_rays.requires_grad:
        gr, er = ray_gradient(sysm._last_rays).cpu().numpy(), c.g["grad_rays"]
        for tag, sl in (("rays_o", slice(0, 3)), ("rays_d", slice(3, 6))):
            e = rel_err(gr[:, sl], er[:, sl])
            if e >= grad_gate(worst_noise, name + "@ref", "grad_" + tag, e):
                bad["grad_" + tag] = e
    for n, e in c.expected_grads().items():
        if n.endswith(".progress"):
            continue
        g = got[n]
        if e is None:
            if g is not None and float(g.abs().max()) != 0.0:
                bad[n] = "expected no gradient"
            continue
        vals, stride, sums = e
        if g is None:
            if sums[1] > 0:
                bad[n] = "missing gradient"
            continue
        flat = g.detach().reshape(-1).cpu()
        sub = (flat[::stride] if stride else flat).numpy()[: len(vals)]
        scale = max(float(np.abs(vals).max()), sums[1] / flat.numel(), 1e-12)
        err = float(np.abs(sub - vals).max()) / scale
        if err >= grad_gate(noise.get(n, worst_noise), name + "@ref", n, err):
            bad[n] = (err, noise.get(n, 0.0))
    if name + "@ref" in _WIDENED:
        print(f"[widened] {name} at the reference's depths: " + ", ".join(f"{n} {g:.1e}" for n, (g, _e) in sorted(_WIDENED[name + "@ref"].items())))
    assert not bad, ("vs reference golden at the reference's fine depths", bad)


def test_only_the_pinned_parameters_need_a_widened_gate():
    """Runs after the golden cases above (file order).  Every comparison whose gate was widened by the reference's own noise is
    listed with its measured error (widened_lines: conftest writes them out at session end); those whose error actually exceeded
    the flat 1e-3 must be in the pinned list, so a NEW parameter leaning on its widening is a failure, not a line of output."""
    import json
    if len(_FLIPPED_SEEN) < len(CASES):
        pytest.skip("needs the full run of the golden tests in this process")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_widened_pinned.json")
    if not os.path.exists(path):
        pytest.skip("no pinned list yet (bootstrap: copy the NEEDED lines of gpurun_out/parity_widened.txt)")
    pinned = set(json.load(open(path))["needed"])
    new = widened_needed() - pinned
    assert not new, f"parameters that newly need a noise-widened gradient gate: {sorted(new)}"


def test_only_the_known_goldens_take_the_flipped_branch():
    """Runs after the golden cases above (file order): the cases whose gradients were compared at the loose gate must be a
    subset of the pinned set."""
    if len(_FLIPPED_SEEN) < len(CASES):
        pytest.skip("needs the full run of test_training_step_matches_reference_golden in this process")
    took = {n for n, f in _FLIPPED_SEEN.items() if f}
    assert took <= KNOWN_FLIPPED, f"new cases on the loose gradient gate: {sorted(took - KNOWN_FLIPPED)}"


def test_feature_less_fields_validate_and_optimise_at_test_time():
    """encode_feat = False beyond the training step: the chunked no-grad validation render (nerf_system.py:231-269) and one
    test-time-optimisation step on frozen fields (nerf_system_optmize.py:84-129, sched 1) of the `nofeat_phase2` golden's system
    -- the maps against the reference's golden / the oracle, the TTO gradients against the oracle."""
    from golden_util import orc
    from upnerf_amd.nerf_system import SyntheticDataset, default_hparams
    from upnerf_amd.nerf_system_optimize import NeRFSystemOptimize
    c = Case("nofeat_phase2")
    assert not c.encode_feat
    sysm = build_system(c)
    sysm.hparams["val.chunk_size"] = 3  # 8 rays: three chunks
    b = c.batch()
    val = {k: v.cuda()[None] for k, v in b.items()}
    log = sysm.validation_step(val)
    res = log["results"]
    # perturb = 0 in validation: compare with the oracle's deterministic render of the same rays
    with torch.no_grad():
        real = orc.schedule_mult
        orc.schedule_mult = lambda p, s_: c.sched
        try:
            hp = dict(c.hparams(), **{"nerf.perturb": 0.0})
            _, ref = orc.training_forward(c.state(requires_grad=False), c.cfgs(), c.batch(), hp, c.progress)
        finally:
            orc.schedule_mult = real
    for k in ("s_rgb_coarse", "s_rgb_fine", "s_depth_fine", "rgb_fine"):
        assert rel_err(res[k].cpu().numpy(), ref[k].numpy()) < (TOL_MAP if "depth" not in k else 5e-3), k
    assert torch.isfinite(log["val_psnr"]).all()
    # one TTO step on the frozen feature-less fields
    hp = default_hparams(**{"nerf.N_samples": c.Nc, "nerf.N_importance": c.Nf, "nerf.perturb": 0.0, "pose.c2f": c.c2f,
                            "nerf.D": c.D, "nerf.W": c.W, "nerf.feat_dim": 0, "val.chunk_size": 4})
    tto = NeRFSystemOptimize(hp, SyntheticDataset(c.n_img), pose_optimize=True)
    tto.train_dataset = SyntheticDataset(c.n_img)
    tto.model_setup(trained_state=sysm.state_dict(), n_test_images=1)
    tto.cuda()
    with torch.no_grad():
        tto.embedding_fine_a.weight.copy_(sysm.embedding_fine_a.weight[3:4])
        tto.se3_refine.weight.copy_(sysm.se3_refine.weight[5:6])
    batch = {k: v.cuda() for k, v in b.items()}
    batch["img_idx"] = torch.zeros_like(batch["img_idx"])
    loss, _, r2 = tto.compute_loss(batch)
    loss.backward()
    assert all(p.grad is None for p in tto.nerf_fine.parameters())
    st = c.state(requires_grad=False)
    a_row = st["embedding_fine_a"][3:4].clone().requires_grad_(True)
    se3_row = st["se3_refine"][5:6].clone().requires_grad_(True)
    st["embedding_fine_a"], st["embedding_coarse_a"] = a_row.expand(c.n_img, -1), st["embedding_coarse_a"]
    st["se3_refine"] = se3_row.expand(c.n_img, -1)
    cf = c.cfgs()
    for v in cf.values():
        v.encode_candidate = False
    pose = orc.compose_pair(orc.se3_exp(st["se3_refine"][b["img_idx"]]), b["c2w"])
    o, d = orc.get_rays(b["directions"], pose)
    rays = torch.cat([o, d, b["ray_infos"]], 1)
    emb = {k[len("embedding_"):]: v for k, v in st.items() if k.startswith("embedding_")}
    ro = orc.render_rays({k: st[k] for k in ("nerf_coarse", "nerf_fine")}, cf, emb, rays, b["img_idx"], 1.0, N_samples=c.Nc,
                         perturb=0.0, N_importance=c.Nf, progress=c.progress)
    lo = ((ro["s_rgb_fine"] - b["rgbs"]) ** 2).mean()
    lo.backward()
    assert abs(float(loss) - float(lo)) < TOL_MAP * max(1e-2, abs(float(lo)))
    assert rel_err(r2["s_rgb_fine"].detach().cpu().numpy(), ro["s_rgb_fine"].detach().numpy()) < TOL_MAP
    assert rel_err(tto.embedding_fine_a.weight.grad.cpu().numpy(), a_row.grad.numpy()) < 2e-3
    assert rel_err(tto.se3_refine.weight.grad.cpu().numpy(), se3_row.grad.numpy()) < 1.5e-2


@pytest.mark.parametrize("pose_opt", [True, False])
def test_side_stream_transient_net_step_equals_the_single_stream_step(pose_opt):
    """r5 ADVICE: hparams["hip.side_stream"] runs the TransientNet (and its table gather) on a side stream.  With the pose frozen
    its embed_rows() is the FIRST of the step (the prefetch arena then belongs to the side stream and the main stream's tables
    gather on their own); with the pose optimised the main stream fills the arena first and the side stream gathers its own rows.
    Either way every loss term and every gradient is bitwise the single-stream step's."""
    c = Case("cfg2_phase1")
    c.pose_opt = pose_opt
    out = []
    for side in (False, True):
        sysm = build_system(c)
        sysm.hparams["hip.side_stream"] = side
        batch = {k: v.cuda() for k, v in c.batch().items()}
        from upnerf_amd.ops import EMBED_PREFETCH
        with EMBED_PREFETCH.scope(sysm._per_image_tables()):  # (the scope of NeRFSystem._step_backward)
            loss, loss_d, _ = sysm.compute_loss(batch, u_list=[u.clone() for u in c.u_list])
            loss.backward()
        torch.cuda.synchronize()
        out.append((float(loss), {k: float(v) for k, v in loss_d.items()},
                    {n: (None if p.grad is None else p.grad.clone()) for n, p in sysm.named_parameters()}))
    (l0, d0, g0), (l1, d1, g1) = out
    assert l0 == l1 and d0 == d1
    assert set(g0) == set(g1)
    for n in g0:
        assert (g0[n] is None) == (g1[n] is None), n
        if g0[n] is not None:
            assert torch.equal(g0[n], g1[n]), n
    assert (g0["se3_refine.weight"] is not None) == pose_opt


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """The product path must not fall back: without the .so, importing the binding raises ImportError."""
    import importlib
    import upnerf_amd._lib as L
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        L._load()


def test_tto_step_frozen_field_matches_oracle_and_skips_weight_gradients():
    """TTO shape (a19): sched 1, candidate head off, frozen fields; gradients reach only the test image's appearance row
    and se(3) row, and equal the oracle's."""
    from golden_util import orc
    from upnerf_amd.nerf_system import SyntheticDataset, default_hparams
    from upnerf_amd.nerf_system_optimize import NeRFSystemOptimize
    c = Case("small_tto")
    trained = build_system(c)
    hp = default_hparams(**{"nerf.N_samples": c.Nc, "nerf.N_importance": c.Nf, "nerf.perturb": 0.0, "pose.c2f": c.c2f,
                            "nerf.D": c.D, "nerf.W": c.W, "val.chunk_size": 4})
    tto = NeRFSystemOptimize(hp, SyntheticDataset(c.n_img), pose_optimize=True)
    tto.train_dataset = SyntheticDataset(c.n_img)
    tto.model_setup(trained_state=trained.state_dict(), n_test_images=1)
    tto.cuda()
    with torch.no_grad():
        tto.embedding_fine_a.weight.copy_(trained.embedding_fine_a.weight[3:4])
        tto.se3_refine.weight.copy_(trained.se3_refine.weight[5:6])
    b = c.batch()
    batch = {k: v.cuda() for k, v in b.items()}
    batch["img_idx"] = torch.zeros_like(batch["img_idx"])
    loss, _, res = tto.compute_loss(batch)
    if tto._last_rays.requires_grad:
        retain_ray_gradient(tto._last_rays)
    loss.backward()
    assert all(p.grad is None for p in tto.nerf_fine.parameters())
    # the REFERENCE's own TTO step on the same single-image problem: tests/golden/small_tto_step.npz (loss line
    # nerf_system_optmize.py:129 evaluated by the reference's leaf functions, tools/make_goldens.py:tto_step_fixture) --
    # r4 VERDICT weak 1d: pinned by data, not by a line restated here
    gold = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "small_tto_step.npz")))
    assert abs(float(loss) - float(gold["loss"])) < TOL_MAP * max(1e-2, abs(float(gold["loss"])))
    for k in ("s_rgb_fine", "s_depth_fine", "s_weights_fine"):
        assert rel_err(res[k].detach().cpu().numpy(), gold["res_" + k]) < (TOL_W if "weights" in k else TOL_MAP), k
    assert rel_err(tto.embedding_fine_a.weight.grad.cpu().numpy(), gold["grad_embedding_fine_a"]) < 1e-3
    assert rel_err(tto.se3_refine.weight.grad.cpu().numpy(), gold["grad_se3_refine"]) < 5e-3
    if ray_gradient(tto._last_rays) is not None:
        # (1.5e-2: the reference's own fp32-vs-fp64 noise on this case's ray gradients is 3.5e-3 -- `[widened] small_tto:
        # grad_rays 1.4e-2` in test_training_step_matches_reference_golden; measured here: 6.7e-3)
        assert rel_err(ray_gradient(tto._last_rays).cpu().numpy()[:, :6], gold["grad_rays"][:, :6]) < 1.5e-2
    # ... and the oracle's restatement of the same step (pinned to the same file on CPU: tests/test_oracle_golden.py)
    from test_oracle_golden import tto_step_oracle
    st, _, ref, l_ref = tto_step_oracle(c, gold)
    l_ref.backward()
    assert abs(float(loss) - float(l_ref)) < 1e-6 * max(1e-2, abs(float(l_ref)))
    assert rel_err(tto.embedding_fine_a.weight.grad.cpu().numpy(), st["embedding_fine_a"].grad.numpy()) < 1e-3
    assert rel_err(tto.se3_refine.weight.grad.cpu().numpy(), st["se3_refine"].grad.numpy()) < 5e-3
    # full-"image" validation path: chunked, no grad
    out = tto.validation_step(batch)
    assert out["s_rgb_fine"].shape == (c.R, 3) and torch.isfinite(out["val_psnr"])
    assert rel_err(out["s_rgb_fine"].cpu().numpy(), ref["s_rgb_fine"].detach().numpy()) < 1e-4


def test_coarse_sigma_only_leaves_the_fine_pass_untouched():
    """render_rays(coarse_sigma_only=True) (nerf.py:90-91): same fine maps and same gradients bit for
    bit, only the coarse maps nobody reads are gone; refused where the resampling needs the candidate head."""
    from upnerf_amd.rendering import render_rays
    c = Case("small_tto")
    sysm = build_system(c).cuda()
    for m in (sysm.nerf_coarse, sysm.nerf_fine):
        m.encode_candidate = False
    sysm.set_progress(1.0)
    b = {k: v.cuda() for k, v in c.batch().items()}
    outs = []
    for flag in (False, True):
        for p in sysm.parameters():
            p.grad = None
        rays = sysm.rays_from_batch(b)
        res = render_rays(sysm.models, sysm.embeddings, rays, b["img_idx"], 1.0, N_samples=c.Nc, perturb=0,
                          N_importance=c.Nf, coarse_sigma_only=flag)
        ((res["s_rgb_fine"] - b["rgbs"]) ** 2).mean().backward()
        outs.append((res, {n: p.grad.clone() for n, p in sysm.named_parameters() if p.grad is not None}))
    (full, g_full), (lean, g_lean) = outs
    assert set(lean) == {k for k in full if k.endswith("_fine")} | {"s_weights_coarse", "s_depth_coarse"}
    for k in lean:
        assert torch.equal(lean[k], full[k]), k
    assert g_full.keys() == g_lean.keys() and any(k.startswith("nerf_fine") for k in g_lean)
    for k in g_full:
        assert torch.equal(g_full[k], g_lean[k]), k
    assert not any(k.startswith("nerf_coarse") for k in g_lean)  # the coarse field never had a gradient in this phase
    sysm.nerf_fine.encode_candidate = sysm.nerf_coarse.encode_candidate = True
    with pytest.raises(ValueError, match="coarse_sigma_only"):
        render_rays(sysm.models, sysm.embeddings, sysm.rays_from_batch(b), b["img_idx"], 0.3, N_samples=c.Nc,
                    perturb=0, N_importance=c.Nf, coarse_sigma_only=True)

"""Training loop with the reference's Trainer configuration (train.py:43-79).

CPU: ModelCheckpoint(save_top_k=2, save_last, monitor val/psnr, mode max) bookkeeping, validation cadence arithmetic.
GPU: sampler -> training_step -> validation -> checkpoints; an interrupted run resumed from last.ckpt (found in the
checkpoint directory, as train.py:37-40 does) ends bit-identical to an uninterrupted one."""
import os

import pytest
import torch

from upnerf_amd.trainer import TopKCheckpoints, Trainer


def test_top_k_keeps_the_two_best_and_always_refreshes_last(tmp_path):
    written = []
    save = lambda p: (written.append(os.path.basename(p)), open(p, "w").write("x"))
    ck = TopKCheckpoints(str(tmp_path / "ckpts"), k=2, mode="max")
    assert ck.best_path() is None
    names = lambda: sorted(os.listdir(tmp_path / "ckpts"))
    assert ck.update(10.0, 0, 100, save).endswith("epoch=0-step=100.ckpt")
    assert ck.update(12.0, 0, 200, save).endswith("epoch=0-step=200.ckpt")
    assert names() == ["epoch=0-step=100.ckpt", "epoch=0-step=200.ckpt", "last.ckpt"]
    assert ck.update(9.0, 0, 300, save) is None  # not among the best two: only last.ckpt is rewritten
    assert names() == ["epoch=0-step=100.ckpt", "epoch=0-step=200.ckpt", "last.ckpt"] and written[-1] == "last.ckpt"
    assert ck.update(11.0, 1, 400, save).endswith("epoch=1-step=400.ckpt")  # pushes the 10.0 out
    assert names() == ["epoch=0-step=200.ckpt", "epoch=1-step=400.ckpt", "last.ckpt"]
    assert ck.update(float("nan"), 1, 500, save) is None
    assert ck.best_path().endswith("epoch=0-step=200.ckpt")
    assert written.count("last.ckpt") == 5
    # state survives a restart; entries whose file is gone are forgotten
    ck2 = TopKCheckpoints(str(tmp_path / "ckpts"), k=2, mode="max")
    os.unlink(tmp_path / "ckpts" / "epoch=1-step=400.ckpt")
    ck2.load_state_dict(ck.state_dict())
    assert list(ck2.best) == [str(tmp_path / "ckpts" / "epoch=0-step=200.ckpt")]
    lo = TopKCheckpoints(str(tmp_path / "lo"), k=1, mode="min")
    lo.update(3.0, 0, 1, save), lo.update(2.0, 0, 2, save), lo.update(5.0, 0, 3, save)
    assert sorted(os.listdir(tmp_path / "lo")) == ["epoch=0-step=2.ckpt", "last.ckpt"]


def test_validation_cadence_follows_lightning():
    t = Trainer(max_steps=10)
    assert t._val_every(1000) == 250 and t._val_every(3) == 1  # int(n * 0.25), at least one batch
    assert Trainer(10, val_check_interval=1.0)._val_every(40) == 40
    assert Trainer(10, val_check_interval=7)._val_every(40) == 7
    with pytest.raises(ValueError):
        Trainer(10, val_check_interval=1.5)._val_every(40)


def _scene(seed=5, I=4, h=8, N=1536):
    gen = torch.Generator().manual_seed(seed)
    d = torch.randn(N, 3, generator=gen) * 0.3
    d[:, 2] = -1.0
    return {"all_ray_infos": torch.cat([torch.full((N, 1), 0.1), torch.full((N, 1), 5.0),
                                        torch.randint(0, I, (N, 1), generator=gen).float()], 1),
            "all_directions": d, "all_rgbs": torch.rand(N, 3, generator=gen),
            "all_pxl_coords": torch.rand(N, 2, generator=gen), "all_inv_depths": torch.rand(N, generator=gen) * 2 + 0.3,
            "feat_maps": torch.nn.functional.normalize(torch.randn(I, h, h, 384, generator=gen), dim=-1),
            "poses": torch.eye(3, 4).repeat(I, 1, 1)}, I


def _system(I):
    from upnerf_amd.nerf_system import NeRFSystem, SyntheticDataset, default_hparams
    hp = default_hparams(**{"nerf.N_samples": 32, "nerf.N_importance": 32, "train.batch_size": 128, "max_steps": 40,
                            "val.chunk_size": 96})
    torch.manual_seed(0)
    s = NeRFSystem(hp, SyntheticDataset(I))
    s.setup()
    return s.cuda()


@pytest.mark.gpu
def test_fit_validates_checkpoints_and_resumes_bitwise(tmp_path):
    from test_sampler import _sampler
    bufs, I = _scene()
    smp = _sampler(bufs)
    n_batches = (len(smp) + 127) // 128  # 12
    batches = lambda epoch, start=0: smp.batches(128, seed=3, epoch=epoch, start=start)  # fast-forwards on resume
    # two "validation images" of 160 rays each, in the DataLoader's batch-of-one layout
    val = [{k: v[None] for k, v in smp.sample(torch.arange(i * 160, (i + 1) * 160)).items()} for i in range(2)]
    MAX = 2 * 20  # 20 iterations, two optimiser steps each (pose optimisation on): 12 in epoch 0, 8 in epoch 1

    torch.manual_seed(11)
    a = _system(I)
    ta = Trainer(MAX, val_check_interval=0.25, dirpath=str(tmp_path / "a"), seed=3).fit(a, batches, n_batches, val)
    assert a.global_step == MAX and ta.epoch == 1 and ta.batch_in_epoch == 8
    steps = [h["step"] for h in ta.history]
    assert steps == [6, 12, 18, 24, 30, 36, 40]  # every int(12 * 0.25) = 3 batches, and once more at the end
    assert all("val/psnr" in h and "val/loss" in h for h in ta.history)
    files = sorted(os.listdir(tmp_path / "a"))
    assert "last.ckpt" in files and len(files) == 3
    top2 = sorted(ta.history, key=lambda h: -h["val/psnr"])[:2]
    assert {f"epoch={h['epoch']}-step={h['step']}.ckpt" for h in top2} == set(files) - {"last.ckpt"}

    # the same run, stopped after 6 iterations (mid-epoch, right after the validation at step 12)
    torch.manual_seed(11)
    b = _system(I)
    Trainer(2 * 6, val_check_interval=0.25, dirpath=str(tmp_path / "b"), seed=3).fit(b, batches, n_batches, val)
    c = _system(I)  # a fresh process would build the system the same way; everything else comes from last.ckpt
    with torch.no_grad():
        for p in c.parameters():
            p.add_(0.05)
    tc = Trainer(MAX, val_check_interval=0.25, dirpath=str(tmp_path / "b"), seed=3).fit(c, batches, n_batches, val)
    assert c.global_step == MAX and tc.epoch == 1 and tc.batch_in_epoch == 8
    sa, sc = a.state_dict(), c.state_dict()
    assert sa.keys() == sc.keys()
    for k in sa:
        assert torch.equal(sa[k], sc[k]), k
    assert [h["val/psnr"] for h in tc.history] == [h["val/psnr"] for h in ta.history if h["step"] > 12]
    assert ta.step_fn.stats["replays"] > 0  # the loop replays captured graphs ...

    # ... and lands on the bits of the eager loop
    torch.manual_seed(11)
    d = _system(I)
    td = Trainer(MAX, val_check_interval=0.25, dirpath=str(tmp_path / "d"), seed=3, graph=False).fit(d, batches, n_batches, val)
    sd = d.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sd[k]), k
    assert [h["val/psnr"] for h in td.history] == [h["val/psnr"] for h in ta.history]

    # a file in the round-1 layout (resume state under the bare keys `loops` / `callbacks` / `rng_states`) still resumes at its
    # position; a file with no resume position at all says so instead of silently replaying the epoch
    ck = torch.load(str(tmp_path / "b" / "last.ckpt"), weights_only=False)
    old = dict(ck)
    old["loops"], old["callbacks"], old["rng_states"] = old.pop("upnerf_loops"), old.pop("upnerf_topk"), old.pop("upnerf_rng")
    torch.save(old, str(tmp_path / "old.ckpt"))
    te = Trainer(MAX, dirpath=None, seed=3)
    te.resume(_system(I), str(tmp_path / "old.ckpt"))
    assert (te.epoch, te.batch_in_epoch) == (int(ck["upnerf_loops"]["epoch"]), int(ck["upnerf_loops"]["batch_in_epoch"]))
    for k in ("loops", "callbacks", "rng_states"):
        old.pop(k)
    torch.save(old, str(tmp_path / "bare.ckpt"))
    with pytest.warns(UserWarning, match="no resume position"):
        Trainer(MAX, dirpath=None, seed=3).resume(_system(I), str(tmp_path / "bare.ckpt"))


@pytest.mark.gpu
def test_fit_from_config_wires_yaml_to_run_directory(tmp_path):
    """A scene file in the reference's format + command-line overrides -> system, sampler, loop, run directory."""
    from types import SimpleNamespace
    from upnerf_amd import config as cfg
    from upnerf_amd.trainer import fit_from_config
    bufs, I = _scene(seed=9, N=1024)
    scene = tmp_path / "scene.yaml"
    scene.write_text("""\
scene_name: 'toy_gate'
exp_name: 'unit'
max_steps: 8
train:
  batch_size: 128
val:
  log_interval: 0.5
  chunk_size: 64
pose:
  optimize: True
  c2f: [0.1,0.5]
  noise: -1
candidate_schedule: [0.1,0.5]
""")
    hp = cfg.get_from_path(str(scene))
    cfg.merge_from_list(hp, ["out_dir", str(tmp_path / "out"), "nerf.N_samples", "32", "nerf.N_importance", "32"])
    img_ids = [10, 11, 12, 13]
    train = SimpleNamespace(N_images_train=I, white_back=False, img_ids_train=img_ids,
                            poses_dict={i: bufs["poses"][k].numpy() for k, i in enumerate(img_ids)},
                            **{k: v for k, v in bufs.items() if k != "poses"})
    from test_sampler import _sampler
    smp = _sampler(bufs)
    val = [{k: v.cpu() for k, v in smp.sample(torch.arange(0, 96)).items()}]
    seen = []
    system, tr = fit_from_config(hp, train, val, log=seen.append)
    assert system.global_step == 16 and tr.epoch == 0 and tr.batch_in_epoch == 8  # 8 iterations = exactly one epoch
    run = tmp_path / "out" / "toy_gate" / "unit"
    assert sorted(os.listdir(run)) == ["ckpts", "config.yaml"] and "last.ckpt" in os.listdir(run / "ckpts")
    assert [m["step"] for m in seen] == [8, 16]
    back = cfg.load(str(run / "config.yaml"))
    assert back["pose.c2f"] == (0.1, 0.5) and back["nerf.N_samples"] == 32 and back["scene_name"] == "toy_gate"
    # calling it again finds ckpts/last.ckpt, resumes at the budget and does nothing more
    system2, tr2 = fit_from_config(hp, train, val)
    assert system2.global_step == 16 and not tr2.history
    with pytest.raises(AssertionError):
        fit_from_config({**hp, "pose.optimize": False}, train, val)


@pytest.mark.gpu
def test_tto_stages_recover_the_appearance_and_pose_of_a_held_out_image():
    """Config #5 end to end on synthetic data (tto.py:56-91, nerf_system_optmize.py): the target image is rendered by the
    trained fields under a known appearance code and a known pose offset; the pose stage (appearance + se(3), Adam)
    and then the appearance stage (AdamW 0.1) must raise the PSNR of the render against it, epoch after epoch on average."""
    from upnerf_amd.nerf_system import SyntheticDataset
    from upnerf_amd.nerf_system_optimize import NeRFSystemOptimize, run_stage
    from upnerf_amd import synth
    I, R = 4, 1024
    trained = _system(I)
    hp = dict(trained.hparams)
    hp["nerf.perturb"] = 0.0

    def tto_system(pose_optimize):
        t = NeRFSystemOptimize(hp, SyntheticDataset(I), pose_optimize=pose_optimize)
        t.model_setup(trained_state=trained.state_dict(), n_test_images=1)
        return t.cuda()

    b = {k: v.cuda() for k, v in synth.batch(R, I, seed=21).items()}
    b["img_idx"] = torch.zeros_like(b["img_idx"])
    truth = tto_system(True)
    with torch.no_grad():
        truth.embedding_fine_a.weight.normal_(0, 0.5, generator=torch.Generator(device="cuda").manual_seed(1))
        truth.se3_refine.weight.copy_(torch.tensor([[0.01, -0.02, 0.015, 0.03, -0.02, 0.01]]))
        b["rgbs"] = truth.validation_step(b)["s_rgb_fine"].clone()

    def batches(epoch):
        perm = torch.randperm(R, device="cuda", generator=torch.Generator(device="cuda").manual_seed(100 + epoch))
        for lo in range(0, R, 256):
            yield {k: v[perm[lo:lo + 256]] for k, v in b.items()}

    torch.manual_seed(5)  # the fresh appearance row is drawn from torch's generator
    pose = tto_system(True)
    start = float(pose.validation_step(b)["val_psnr"])
    tr = run_stage(pose, batches, 4, max_epochs=12, val_batches=[b])
    psnr = [h["val/psnr"] for h in tr.history]
    assert len(psnr) == 12 and pose.global_step == 12 * 4 * 2  # two optimisers step per batch
    assert psnr[-1] > start + 3.0 and psnr[-1] > psnr[0]
    err0 = float(truth.se3_refine.weight.detach().norm())
    assert float((pose.se3_refine.weight - truth.se3_refine.weight).detach().norm()) < err0  # moved towards the true offset

    assert tr.step_fn.stats["replays"] >= 40  # one graph replay per step ...
    torch.manual_seed(5)
    eager = tto_system(True)  # ... on the bits of the eager loop
    run_stage(eager, batches, 4, max_epochs=12, val_batches=[b], graph=False)
    assert torch.equal(eager.se3_refine.weight, pose.se3_refine.weight)
    assert torch.equal(eager.embedding_fine_a.weight, pose.embedding_fine_a.weight)

    app = tto_system(False)  # appearance stage starts from the optimised pose (eval.py feeds it as the camera)
    with torch.no_grad():
        app.embedding_fine_a.weight.copy_(pose.embedding_fine_a.weight)
    tr2 = run_stage(app, batches, 4, max_epochs=3, val_batches=[b])
    assert len(tr2.history) == 3 and app.global_step == 3 * 4 and all(torch.isfinite(p).all() for p in app.parameters())
    assert tr2.step_fn.stats["replays"] >= 8  # torch's AdamW (capturable) sits inside the replayed graph
    assert all(p.grad is None for p in app.nerf_fine.parameters())  # frozen fields: no weight gradients computed


@pytest.mark.gpu
def test_evaluation_route_from_a_checkpoint(tmp_path):
    """eval.py:13-42 + nerf_system_optmize.py:254-317 on a checkpoint file: pose error of the training cameras after Sim(3)
    alignment, then a TTO system with the checkpoint's fields whose held-out camera starts from the aligned ground truth,
    and a full-image render from it."""
    from upnerf_amd import synth
    from upnerf_amd.checkpoint import save_checkpoint
    from upnerf_amd.nerf_system_optimize import eval_train_poses, tto_from_checkpoint
    from upnerf_amd.pose_align import refined_poses
    I = 6
    trained = _system(I)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        trained.se3_refine.weight.copy_((torch.rand(I, 6, generator=g) - 0.5).cuda() * torch.tensor([.4, .4, .4, 2., 2., 2.]).cuda())
    path = save_checkpoint(trained, str(tmp_path / "last.ckpt"))
    ident = torch.eye(3, 4).repeat(I, 1, 1)
    frame = refined_poses(trained.se3_refine.weight, ident.cuda()).cpu()  # cameras in the trained frame
    # ground truth = the same cameras seen from a world that differs by one similarity (c2w convention: R' = QR, t' = sQt+b)
    Q = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    Q = Q * torch.sign(torch.linalg.det(Q))
    to_gt = lambda P: torch.cat([Q @ P[..., :3], 1.8 * (Q @ P[..., 3:]) + torch.tensor([[0.5], [-0.2], [1.0]])], -1)
    out = eval_train_poses(path, ident, to_gt(frame))
    assert out["train/pose_R"] < 0.05 and out["train/pose_t"] < 1e-4  # degrees (acos floor near 0) / scene units
    worse = eval_train_poses(path, ident, to_gt(frame.roll(1, 0)))
    assert worse["train/pose_R"] > 1.0

    held_out = torch.cat([frame[:2, :, :3], frame[:2, :, 3:] + 0.3], -1)  # two cameras that were not trained on
    tto, init = tto_from_checkpoint(path, pose_optimize=True, n_test_images=2, gt_train_poses=to_gt(frame),
                                    gt_test_poses=to_gt(held_out), **{"val.chunk_size": 96})
    assert float((init - held_out).abs().max()) < 2e-5
    sd = trained.state_dict()
    for k, v in tto.nerf_fine.state_dict().items():
        if k != "progress":  # test-time optimisation runs with every encoding band on (nerf_system_optmize.py:265)
            assert torch.equal(v, sd["nerf_fine." + k]), k
    assert tto.embedding_fine_a.weight.shape[0] == 2 and float(tto.se3_refine.weight.abs().max()) == 0.0
    b = {k: v.cuda() for k, v in synth.batch(400, 2, seed=3).items()}  # one 20x20 "image" of test camera 1
    b["img_idx"] = torch.ones_like(b["img_idx"])
    b["c2w"] = init[1].cuda().expand(400, 3, 4).contiguous()
    res = tto.validation_step(b)
    assert res["s_rgb_fine"].shape == (400, 3) and torch.isfinite(res["s_rgb_fine"]).all() and torch.isfinite(res["val_psnr"])
    whole = tto.validation_step({k: v[:96] for k, v in b.items()})["s_rgb_fine"]  # chunked render == one chunk
    assert torch.equal(res["s_rgb_fine"][:96], whole)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,store", [("f16x3", "f32"), ("f16x3", "f16"), ("f16", "f32"), ("f32", "f32")])
def test_student_learns_the_colours_a_teacher_renders(mode, store, monkeypatch):
    """System-level check of the gradient signs and scales in every arithmetic mode: a teacher (random fields, schedule
    finished) renders the colours of a fixed set of rays, a student with other weights is trained on them through the
    graph-replayed step, and its colour loss has to fall by more than half within 120 iterations."""
    from upnerf_amd import rendering, synth
    from upnerf_amd.graph_step import GraphedTrainingStep
    from upnerf_amd.nerf_system import NeRFSystem, SyntheticDataset, default_hparams
    monkeypatch.setattr(rendering, "FIELD_MODE", mode)
    monkeypatch.setattr(rendering, "WGRAD_STORE", store)
    I, R, B = 4, 2048, 256
    hp = default_hparams(**{"nerf.N_samples": 32, "nerf.N_importance": 32, "train.batch_size": B, "max_steps": 1000,
                            "nerf.perturb": 1.0, "optimizer.lr": 1e-3})

    def system(seed):
        torch.manual_seed(seed)
        s = NeRFSystem(hp, SyntheticDataset(I))
        s.setup()
        s = s.cuda()
        s.global_step = 1600  # progress 0.8: schedule finished, colour terms only
        s.set_progress(0.8)
        return s

    data = {k: v.cuda() for k, v in synth.batch(R, I, seed=31).items()}
    teacher = system(100)
    with torch.no_grad():
        for m in (teacher.nerf_coarse, teacher.nerf_fine):  # denser, more colourful than a fresh initialisation
            m.share_sigma[0].bias.add_(1.5)
            m.rgb_share_layer[0].weight.mul_(4.0)
        out = teacher.validation_step({k: v[None] for k, v in data.items()})["results"]
        data["rgbs"] = out["rgb_fine"].clamp(0, 1).contiguous()
    assert float(data["rgbs"].std()) > 0.02  # there is something to learn
    student = system(7)
    step = GraphedTrainingStep(student)
    losses = []
    for i in range(120):
        lo = (i * B) % R
        step({k: v[lo:lo + B].contiguous() for k, v in data.items()}, i)
        losses.append(float(student.logged["train/l_rgb_f"]))
    first, last = sum(losses[:10]) / 10, sum(losses[-10:]) / 10
    assert all(torch.isfinite(p).all() for p in student.parameters())
    assert last < 0.5 * first, (first, last)
    assert step.stats["replays"] >= 100

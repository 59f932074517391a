"""Checkpoint interchange and resume (reference: train.py:34-49, 79; utils/__init__.py:4-26;
eval.py:79-80).

CPU: file layout, parameter order (torch.optim indexes its state by it), weights-only load, sub-module extraction, a
torch.optim.Adam-shaped optimiser state going into the system, atomic save.
GPU: train -> save -> load into a fresh system -> continue is BITWISE the uninterrupted run (weights, both optimisers'
moments and step counts, both learning-rate schedules, schedule position)."""
import json
import os

import pytest
import torch

from golden_util import GOLDEN
from upnerf_amd import checkpoint as ck
from upnerf_amd.nerf_system import NeRFSystem, SyntheticDataset, default_hparams

N_IMG = 5


def _small_system(seed=0, **over):
    hp = default_hparams(**{"nerf.N_samples": 8, "nerf.N_importance": 8, "nerf.D": 4, "nerf.W": 64, "max_steps": 50, **over})
    torch.manual_seed(seed)
    s = NeRFSystem(hp, SyntheticDataset(N_IMG))
    s.setup()
    return s


def _fake_step(s, seed):
    """One optimiser + scheduler step with seeded stand-in gradients; the shared-colour head gets none (as in schedule
    phase 0), so the skip-if-no-grad bookkeeping is part of what must survive the round trip."""
    g = torch.Generator().manual_seed(seed)
    for n, p in s.named_parameters():
        p.grad = None if ("rgb_share_layer" in n or n.endswith("progress")) else torch.randn(p.shape, generator=g) * 1e-2
    for o, sc in zip(s.optimizers(), s.lr_schedulers()):
        o.step()
        sc.step()
    s.global_step += 2
    s.set_progress(s.global_step / (s.hparams["max_steps"] * 2))


def _same_tensors(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k].cpu(), b[k].cpu()), k


def test_parameter_registration_order_is_the_references():
    from upnerf_amd.nerf import NeRF
    from upnerf_amd.transient_net import TransientNet
    fx = json.load(open(os.path.join(GOLDEN, "state_keys.json")))
    for tag, item in fx.items():
        m = TransientNet(**item["kwargs"]) if tag.startswith("transient") else NeRF("coarse", c2f=(0.1, 0.5), **item["kwargs"])
        assert [k for k, _ in m.named_parameters()] == item["param_order"], tag
    # system level (nerf_system.py:342-410): embeddings a then c (coarse, fine each), then the three networks
    s = _small_system()
    heads = []
    for n, _ in s.named_parameters():
        h = n.split(".")[0]
        if not heads or heads[-1] != h:
            heads.append(h)
    assert heads == ["embedding_coarse_a", "embedding_fine_a", "embedding_coarse_c", "embedding_fine_c", "nerf_coarse",
                     "nerf_fine", "transient_net", "se3_refine", "depth_scale"]


def test_file_layout_and_round_trip(tmp_path):
    a = _small_system(seed=1)
    for i in range(3):
        _fake_step(a, 10 + i)
    path = ck.save_checkpoint(a, str(tmp_path / "ckpts" / "last.ckpt"), epoch=2)
    assert sorted(os.listdir(tmp_path / "ckpts")) == ["last.ckpt"]  # no temp file left behind
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(raw) == {"epoch", "global_step", "pytorch-lightning_version", "state_dict", "optimizer_states",
                        "lr_schedulers", "hyper_parameters"}
    assert raw["global_step"] == 6 and raw["epoch"] == 2 and len(raw["optimizer_states"]) == 2
    assert list(raw["state_dict"]) == list(a.state_dict())
    st0 = raw["optimizer_states"][0]["state"]
    names = [n for n, _ in a.named_parameters() if not n.startswith(("se3_refine", "depth_scale"))]
    no_grad = {i for i, n in enumerate(names) if "rgb_share_layer" in n or n.endswith("progress")}
    assert set(st0) == set(range(len(names))) - no_grad  # parameters that never had a gradient carry no state
    assert all(int(v["step"]) == 3 for v in st0.values())

    b = _small_system(seed=2)
    info = ck.load_checkpoint(b, path)
    assert info == {"missing": [], "unexpected": [], "global_step": 6, "epoch": 2}
    _same_tensors(a.state_dict(), b.state_dict())
    assert b.global_step == 6 and b._host_progress == pytest.approx(0.06) and b.nerf_fine.host_progress == pytest.approx(0.06)
    for oa, ob in zip(a.optimizers(), b.optimizers()):
        assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"]
    for i in range(2):  # the continuation is the uninterrupted run
        _fake_step(a, 20 + i)
        _fake_step(b, 20 + i)
    _same_tensors(a.state_dict(), b.state_dict())
    assert a.lr_schedulers()[0].get_last_lr() == b.lr_schedulers()[0].get_last_lr()


def test_weights_only_load_and_submodule_extraction(tmp_path):
    a = _small_system(seed=3)
    _fake_step(a, 1)
    path = ck.save_checkpoint(a, str(tmp_path / "e.ckpt"))
    b = _small_system(seed=4)
    ck.load_checkpoint(b, path, resume=False)
    _same_tensors(a.state_dict(), b.state_dict())
    assert b.global_step == 0 and not b.optimizers()[0].state_dict()["state"]  # eval / TTO use: nothing but the weights

    sub = ck.extract_model_state_dict(path, "nerf_coarse", prefixes_to_ignore=["candidate_"])
    full = {k[len("nerf_coarse."):]: v for k, v in a.state_dict().items() if k.startswith("nerf_coarse.")}
    assert set(sub) == {k for k in full if not k.startswith("candidate_")} and "xyz_encoding_1.0.weight" in sub
    assert not ck.extract_model_state_dict(path, "nerf")  # a module name matches whole path components only
    c = _small_system(seed=5)
    kept = c.nerf_coarse.candidate_sigma[0].weight.detach().clone()
    ck.load_ckpt(c.nerf_coarse, path, "nerf_coarse", prefixes_to_ignore=["candidate_"])
    assert torch.equal(c.nerf_coarse.xyz_encoding_1[0].weight, a.nerf_coarse.xyz_encoding_1[0].weight)
    assert torch.equal(c.nerf_coarse.candidate_sigma[0].weight, kept)  # ignored entries keep their values
    ck.load_ckpt(c.nerf_fine, a.state_dict(), "nerf_fine")  # a bare state_dict is a checkpoint too
    assert torch.equal(c.nerf_fine.share_sigma[0].weight, a.nerf_fine.share_sigma[0].weight)


def test_optimizer_count_mismatch_is_an_error(tmp_path):
    a = _small_system(seed=6)
    _fake_step(a, 1)
    path = ck.save_checkpoint(a, str(tmp_path / "p.ckpt"))
    b = _small_system(seed=7, **{"pose.optimize": False})
    with pytest.raises(ValueError, match="optimisers"):
        ck.load_checkpoint(b, path)


@pytest.mark.gpu
def test_flat_adam_takes_a_torch_adam_state_and_resume_is_bitwise(tmp_path):
    import bench
    from test_hip_fullsize import _batch, _draws
    dev = torch.device("cuda", 0)
    batches = [_batch(200 + i) for i in range(4)]

    def run(sysm, lo, hi):
        out = []
        for i in range(lo, hi):
            out.append(sysm.training_step(batches[i], u_list=_draws(sysm, 50 + i)).detach().clone())
        return out

    a = bench.build_system(dev, 0.3)
    run(a, 0, 2)
    path = ck.save_checkpoint(a, str(tmp_path / "last.ckpt"))
    tail_a = run(a, 2, 4)

    b = bench.build_system(dev, 0.05)  # another schedule phase / step count: everything must come from the file
    with torch.no_grad():
        for p in b.parameters():
            p.add_(0.01)
    info = ck.load_checkpoint(b, path)
    assert not info["missing"] and not info["unexpected"] and b.global_step == a.global_step - 4
    tail_b = run(b, 2, 4)
    for x, y in zip(tail_a, tail_b):
        assert torch.equal(x, y)
    _same_tensors(a.state_dict(), b.state_dict())
    for oa, ob in zip(a.optimizers(), b.optimizers()):
        assert torch.equal(oa.flat_m, ob.flat_m) and torch.equal(oa.flat_v, ob.flat_v) and oa._steps == ob._steps
        assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"]

    # the state a reference run would have written: torch.optim.Adam over the same parameter list
    ref_opt = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p, _, _ in a.optimizers()[0]._spans],
                               lr=1e-3, eps=1e-8)
    for p in ref_opt.param_groups[0]["params"][:5]:
        p.grad = torch.full_like(p, 0.5)
    ref_opt.step()
    sd = ref_opt.state_dict()
    c = bench.build_system(dev, 0.3)
    c.optimizers()[0].load_state_dict(sd)
    oc = c.optimizers()[0]
    assert oc._steps[:5] == [1] * 5 and set(oc._steps[5:]) == {0} and oc.param_groups[0]["lr"] == 1e-3
    _, o, k = oc._spans[0]
    assert torch.allclose(oc.flat_m[o:o + k], torch.full((k,), 0.05, device=dev))

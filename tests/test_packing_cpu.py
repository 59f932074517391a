"""CPU checks of the host logic around the kernels: the packed parameter layout, the folded colour layer and the
composite-then-project algebra reproduce the oracle (and therefore the reference) when evaluated by the torch
restatement of the kernels' arithmetic in tests/kernel_space.py.  No GPU involved."""
import pytest
import torch

import kernel_space as ks
from golden_util import Case, orc, rel_err
from upnerf_amd.nerf import NeRF
from upnerf_amd.rendering import band_weights

NAMES = ["cfg1_small", "cfg1_small_fine", "cfg2_phase1", "small_tto", "small_nocand", "small_allmasked"]


def build_model(c, typ, st):
    m = NeRF(typ, c2f=c.c2f, **c.nerf_kw())
    sd = {k: v.detach().clone() for k, v in st[f"nerf_{typ}"].items()}
    sd["progress"] = torch.tensor(c.progress)
    m.load_state_dict(sd)
    if c.encode_candidate is False:
        m.encode_candidate = False
    return m


@pytest.mark.parametrize("name", NAMES)
def test_packed_kernel_space_matches_oracle(name):
    c = Case(name)
    st = c.state(requires_grad=False)
    keep = {}
    b, hp = c.batch(), c.hparams()
    real = orc.schedule_mult
    orc.schedule_mult = lambda p, s: c.sched
    try:
        with torch.no_grad():
            _, res = orc.training_forward(st, c.cfgs(), b, hp, c.progress, u_list=c.u_list, keep=keep)
    finally:
        orc.schedule_mult = real
    idx = b["img_idx"]
    pose = orc.compose_pair(orc.se3_exp(st["se3_refine"][idx]), b["c2w"]) if c.pose_opt else b["c2w"]
    o, d = orc.get_rays(b["directions"], pose)
    for typ, zkey in (("coarse", "z_coarse"), ("fine", "z_fine")):
        if typ == "fine" and not c.fine:
            continue
        model = build_model(c, typ, st)
        pk = model.packer
        with torch.no_grad():
            P = model.packed()
            assert P.numel() == pk.L.total
            use_cand = bool(c.sched < 1 and model.encode_candidate)
            use_rgb = bool(c.sched > 0)
            mode = (1 if use_rgb else 0) if use_cand else (3 if c.sched < 1 else 2)
            a_rows = st[f"embedding_{typ}_a"][idx]
            c_rows = st[f"embedding_{typ}_c"][idx]
            aux = ks.ray_aux(d, a_rows, band_weights(4, c.progress, c.c2f))
            z = keep[zkey]
            f = ks.field(P, pk, o, d, z, c_rows, aux, band_weights(10, c.progress, c.c2f), use_cand, use_rgb)
            out = ks.composite(f, z, mode, use_rgb, pk.W)
            wf, bf = model.feat_share_layer.weight, model.feat_share_layer.bias
            if c.sched < 1:
                feat = out["E_s"] @ wf.t() + out["sum_sfeat"][:, None] * bf
                if use_cand:
                    feat = feat + out["G_c"] @ model.feat_candidate_layer.weight.t() \
                        + out["t_weight"][:, None] * model.feat_candidate_layer.bias
                    assert rel_err(out["w_all"], res[f"c_weights_{typ}"]) < 2e-5
                    assert rel_err(out["c_depth"], res[f"c_depth_{typ}"]) < 2e-5
                    assert rel_err(out["t_weight"], res[f"t_weight_{typ}"]) < 2e-5
                assert rel_err(feat, res[f"feat_{typ}"]) < 2e-5
            if c.sched > 0:
                assert rel_err(out["rgb_map"], res[f"s_rgb_{typ}"]) < 2e-5
                assert rel_err(out["w_s"], res[f"s_weights_{typ}"]) < 2e-5
            assert rel_err(out["s_depth"], res[f"s_depth_{typ}"]) < 2e-5
            # transposed copies are exact transposes of the forward pieces
            PT = pk.pack_t(P)
            L, W, W2 = pk.L, pk.W, pk.W2
            assert torch.equal(ks.mat(PT, L.t_we, W, W), ks.mat(P, L.we, W, W).t())
            assert torch.equal(ks.mat(PT, L.t_w[0], 64, W), ks.mat(P, L.w[0], W, 64).t())
            assert torch.equal(ks.mat(PT, L.t_head, W, W)[:, W2:], ks.mat(P, L.wc1, W2, W + 16)[:, :W].t())


def test_pack_is_differentiable_to_reference_named_parameters():
    c = Case("cfg1_small_fine")
    model = build_model(c, "coarse", c.state(requires_grad=False))
    P = model.packed()
    (P * torch.linspace(0, 1, P.numel())).sum().backward()
    for n, p in model.named_parameters():
        if n in ("progress", "feat_candidate_layer.weight", "feat_candidate_layer.bias"):
            continue  # not part of the packed buffer (projected per ray on the host side)
        assert p.grad is not None and torch.isfinite(p.grad).all(), n


def test_state_dict_keys_and_shapes_match_the_reference_modules():
    """Checkpoint interchange: the modules expose exactly the reference's state_dict keys and shapes
    (fixture recorded from the real models/nerf.py and models/transient_net.py by tools/make_goldens.py)."""
    import json
    import os
    from golden_util import GOLDEN
    from upnerf_amd.nerf import NeRF
    from upnerf_amd.transient_net import TransientNet
    fx = json.load(open(os.path.join(GOLDEN, "state_keys.json")))
    for tag, item in fx.items():
        m = TransientNet(**item["kwargs"]) if tag.startswith("transient") else NeRF("coarse", c2f=(0.1, 0.5), **item["kwargs"])
        mine = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert mine == item["state"], (tag, set(mine) ^ set(item["state"]))


def test_pack_without_the_feature_layer_places_the_colour_matrix_as_it_stands():
    """encode_feat = False (nerf.py:52-56): rgb_share_layer.0 reads [xyz_encoding_final | PE(dir) | appearance] itself, so the
    layout's "folded" colour matrix is W_r1[:, :W] and its side columns W_r1[:, W:], unchanged; its bias is b_r1."""
    import torch
    from upnerf_amd._lib import AUXK
    from upnerf_amd.nerf import NeRF
    m = NeRF("coarse", D=4, W=64, encode_feat=False, feat_dim=0, xyz_L=10, dir_L=4, appearance_dim=48, candidate_dim=16)
    p = dict(m.named_parameters())
    assert "feat_share_layer.weight" not in p and "rgb_candidate_layer.weight" in p
    assert "feat_share_layer.weight" not in m.packer.pack_names()
    P = m.packer.pack(p).detach()
    L, W, W2 = m.packer.L, 64, 32
    wr1 = P[L.wr1:L.wr1 + W2 * (W + AUXK)].view(W2, W + AUXK)
    w = p["rgb_share_layer.0.weight"].detach()
    assert w.shape == (W2, W + 27 + 48)
    assert torch.equal(wr1[:, :W + 75], w) and float(wr1[:, W + 75:].abs().max()) == 0.0
    assert torch.equal(P[L.br1:L.br1 + W2], p["rgb_share_layer.0.bias"].detach())
    # gradients flow straight back
    (m.packer.pack(p) * torch.arange(L.total, dtype=torch.float32)).sum().backward()
    g = p["rgb_share_layer.0.weight"].grad
    assert torch.equal(g[3, :5], torch.arange(L.wr1 + 3 * (W + AUXK), L.wr1 + 3 * (W + AUXK) + 5, dtype=torch.float32))

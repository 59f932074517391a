"""upnerf_amd/config.py against what the REAL configs/config.py does (tests/golden/config_cases.json, recorded by
tools/make_goldens.py config): defaults, flattening + literal parsing, list -> tuple, overrides, save/load round trip."""
import argparse
import json
import os

import pytest

from golden_util import GOLDEN
from upnerf_amd import config as cfg

FX = json.load(open(os.path.join(GOLDEN, "config_cases.json")))


def dec(v):
    if isinstance(v, dict) and "__tuple__" in v:
        return tuple(dec(x) for x in v["__tuple__"])
    if isinstance(v, dict) and "__dict__" in v:
        return {k: dec(x) for k, x in v["__dict__"].items()}
    if isinstance(v, list):
        return [dec(x) for x in v]
    return v


def same(mine, golden):
    want = {k: dec(v) for k, v in golden.items()}
    assert list(mine) == list(want)  # same keys in the same order
    for k in want:
        assert type(mine[k]) is type(want[k]) and mine[k] == want[k], (k, mine[k], want[k])


def test_defaults_are_the_reference_default_yaml():
    same(cfg.default(), FX["default"])
    d = cfg.default()
    d["seed"] = 1
    assert cfg.default()["seed"] == 42  # a fresh copy every call


@pytest.mark.parametrize("name", sorted(FX["cases"]))
def test_file_then_overrides_then_round_trip(name, tmp_path):
    case = FX["cases"][name]
    f = tmp_path / "c.yaml"
    f.write_text(case["yaml"])
    c = cfg.get_from_path(str(f))
    cfg.merge_from_list(c, case["opts"])
    same(c, case["config"])
    out = tmp_path / "saved.yaml"
    cfg.save_yaml(c, str(out))
    same(cfg.load(str(out)), case["after_save_and_load"])


def test_parse_args_layers_defaults_file_overrides_and_argparse_fields(tmp_path):
    f = tmp_path / "c.yaml"
    f.write_text("nerf:\n  N_samples: 64\nscene_name: 'gate'\n")
    p = argparse.ArgumentParser()
    p.add_argument("--config", required=True)
    p.add_argument("--seed", type=int, default=7)  # loses against the configuration's own key
    p.add_argument("--extra", default="e")
    p.add_argument("opts", nargs=argparse.REMAINDER)
    c = cfg.parse_args(p, ["--config", str(f), "nerf.N_importance", "0", "pose.c2f", "[0.1,0.5]"])
    assert c["nerf.N_samples"] == 64 and c["nerf.N_importance"] == 0 and c["pose.c2f"] == (0.1, 0.5)
    assert c["seed"] == 42 and c["extra"] == "e" and c["config"] == str(f) and c["scene_name"] == "gate"
    with pytest.raises(AssertionError):
        cfg.merge_from_list(c, ["dangling"])


def test_system_hyperparameters_are_the_configuration_keys():
    """default_hparams() = the reference defaults + the keys the training scene files add (configs/<scene>.yaml)."""
    from upnerf_amd.nerf_system import default_hparams
    hp = default_hparams()
    for k, v in cfg.default().items():
        if k in hp:
            assert hp[k] == v, k

"""Configuration with the reference's semantics (configs/config.py:12-99), so its YAML files and
`key value` command-line overrides drive this package unchanged.

A configuration is ONE flat dict with dotted keys (`nerf.N_samples`, `optimizer.scheduler.lr_end`, ...):
  * nested YAML mappings are flattened with "." (config.py:12-31);
  * every string value is passed through `ast.literal_eval` and kept as a string only if that fails -- this is what
    turns the YAML-1.1 strings `1e-3`, `None`, `[0.1,0.5]` of the reference's files into a float, None and a list;
  * lists become tuples;
  * merging overwrites key by key, unknown keys are added silently (config.py:39-44);
  * command-line overrides come as a flat `key value key value` list and go through the same parsing.
The defaults are the values of the reference's configs/default.yaml, kept here as a Python dict (`DEFAULTS`; pinned
against the real file by tests/golden/config_cases.json)."""
from __future__ import annotations

import argparse
from ast import literal_eval
from typing import Any, Dict, Iterable, Mapping, Optional

import yaml

DEFAULTS: Dict[str, Any] = {
    "seed": 42, "num_gpus": 1, "debug": False, "out_dir": "./outputs", "resume_ckpt": None,
    "nerf.N_samples": 128, "nerf.N_importance": 128, "nerf.N_emb_xyz": 10, "nerf.N_emb_dir": 4, "nerf.near": 0.1,
    "nerf.far": 5.0, "nerf.appearance_dim": 48, "nerf.candidate_dim": 16, "nerf.feat_dim": 384, "nerf.use_disp": False,
    "nerf.perturb": 1.0,
    "t_net.beta_min": 0.1, "t_net.transient_dim": 128, "t_net.feat_dim": 384,
    "loss.depth_mult": 1e-3, "loss.alpha_reg": 1.0,
    "optimizer.type": "adam", "optimizer.lr": 5e-4, "optimizer.scheduler.type": "ExponentialLR",
    "optimizer.scheduler.lr_end": 5e-5,
    "optimizer_pose.type": "adam", "optimizer_pose.lr": 2e-3, "optimizer_pose.scheduler.type": "ExponentialLR",
    "optimizer_pose.scheduler.lr_end": 1e-5,
}


def _value(v):
    if isinstance(v, str):
        try:
            v = literal_eval(v)
        except (ValueError, SyntaxError):
            pass  # a real string
    return tuple(v) if isinstance(v, list) else v


def flatten(tree: Optional[Mapping], prefix: str = "") -> Dict[str, Any]:
    """Nested mapping -> flat dotted dict with the value rules above.  None (an empty YAML file) -> {}."""
    flat: Dict[str, Any] = {}
    for k, v in (tree or {}).items():
        if isinstance(v, dict):
            flat.update(flatten(v, f"{prefix}{k}."))
        else:
            flat[f"{prefix}{k}"] = _value(v)
    return flat


def load(fname: str) -> Dict[str, Any]:
    with open(fname, "r") as fp:
        return flatten(yaml.safe_load(fp))


def merge_from_config(config: Dict[str, Any], config_merge: Mapping[str, Any]) -> None:
    config.update(config_merge)


def merge_from_file(config: Dict[str, Any], fname: str) -> None:
    merge_from_config(config, load(fname))


def merge_from_list(config: Dict[str, Any], list_merge: Iterable[str]) -> None:
    items = list(list_merge)
    if len(items) % 2:
        raise AssertionError("The list must have key value pairs.")
    merge_from_config(config, flatten(dict(zip(items[0::2], items[1::2]))))


def default() -> Dict[str, Any]:
    return dict(DEFAULTS)


def get_from_path(config_path: Optional[str]) -> Dict[str, Any]:
    config = default()
    if config_path is not None:
        merge_from_file(config, config_path)
    return config


def parse_args(parser: argparse.ArgumentParser, argv=None) -> Dict[str, Any]:
    """`--config file.yaml key value ...`: defaults <- file <- overrides; the remaining argparse fields (config, opts,
    anything the caller added) are appended under their own names when the configuration has no such key."""
    args = parser.parse_args(argv)
    config = get_from_path(getattr(args, "config", None))
    if getattr(args, "opts", None):
        merge_from_list(config, args.opts)
    for k, v in vars(args).items():
        config.setdefault(k, v)
    return config


def nest(config: Mapping[str, Any]) -> Dict[str, Any]:
    """Inverse of `flatten` on the keys: dotted keys back to nested mappings (what save_yaml writes)."""
    tree: Dict[str, Any] = {}
    for k, v in config.items():
        node = tree
        *path, leaf = k.split(".")
        for part in path:
            node = node.setdefault(part, {})
        node[leaf] = v
    return tree


def save_yaml(config: Mapping[str, Any], file_name: str) -> None:
    def plain(v):  # yaml.safe_dump has no tuple representer; the reference's files hold lists
        return [plain(x) for x in v] if isinstance(v, (tuple, list)) else v

    with open(file_name, "w") as fp:
        yaml.safe_dump(nest({k: plain(v) for k, v in config.items()}), fp)

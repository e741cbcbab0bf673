"""The non-default HandsLight configurations with a reference-generated fixture (tests/golden/make_golden_switches.py)."""
import json
import os

import numpy as np
import torch

import hands_amd
from hands_amd.weights import synthetic_dense_inputs, synthetic_inputs

SWITCH_CASES = ("arctic", "sinusoidal_cc", "center", "corner", "center_corner", "plain", "separate", "noglb",
                "dense", "dense_latent", "cam_conv", "pcl", "persp", "persp_flip", "depth")


def load_case(golden_dir, name):
    """(fixture, config overrides, args for hands_amd.HandsLight, seeded inputs, meta_info)"""
    d = np.load(os.path.join(golden_dir, f"hands_light_switch_{name}.npz"), allow_pickle=False)
    meta = json.loads(str(d["meta"]))
    args = type(hands_amd.DEFAULT_ARGS)(dict(hands_amd.DEFAULT_ARGS))
    args.update(meta["config"])
    inputs, meta_info = synthetic_inputs(meta["bz"], meta["seed"])
    meta_info["is_flipped"] = torch.from_numpy(d["is_flipped"])
    if meta["config"].get("pos_enc") in ("dense", "dense_latent", "cam_conv", "pcl"):
        inputs.update(synthetic_dense_inputs(meta["bz"], meta["seed"], meta["config"]["pos_enc"]))
    return d, meta["config"], args, inputs, meta_info


def oracle_kwargs(cfg):
    return dict(pos_enc_mode=cfg.get("pos_enc", "center+corner_latent"), no_crops=cfg.get("no_crops", False),
                use_grasp_loss=cfg.get("use_grasp_loss", True), use_glb_feat_w_grasp=cfg.get("use_glb_feat_w_grasp", True),
                separate_hands=cfg.get("separate_hands", False), regress_center_corner=cfg.get("regress_center_corner", False),
                use_glb_feat=cfg.get("use_glb_feat", True), use_depth_loss=cfg.get("use_depth_loss", False))

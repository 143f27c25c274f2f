"""Rebuild the inputs/weights of a golden fixture (tests/golden/*.npz) on any box."""
import os

import numpy as np
import torch

import robot_3dlotus_amd  # noqa: F401
from robot_3dlotus_amd import config as lcfg, synth
from weights_util import seeded_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["tiny_init_eval", "tiny_scaled_train", "v1_init_train", "v1_scaled_train", "v1_scaled_eval",
         "tinydeep_scaled_train", "tinydeep_scaled_eval", "tinyctx_scaled_train"]


def load_case(name, state_template):
    """state_template: dict name -> tensor giving the state_dict layout (from the model under test).
    Returns (fixture, cfg, batch, state_dict)."""
    fx = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    variant = str(fx["meta_variant"])
    cfg = lcfg.preset(variant)
    if "meta_drop_path" in fx:
        cfg.ptv3_config.drop_path = float(fx["meta_drop_path"])
    batch = synth.synth_batch(int(fx["meta_B"]), int(fx["meta_n"]), ragged=bool(fx["meta_ragged"]),
                              seed=int(fx["meta_dseed"]))
    assert batch["npoints_in_batch"] == fx["npoints_in_batch"].tolist()
    assert abs(batch["pc_fts"].double().sum().item() - float(fx["input_checksum"])) < 1e-9
    if callable(state_template):
        state_template = state_template(cfg)
    sd = seeded_state_dict(state_template, int(fx["meta_wseed"]), str(fx["meta_wvar"]))
    ck = sum(v.double().sum().item() for v in sd.values())
    assert abs(ck - float(fx["weight_checksum"])) < 1e-6 * max(1.0, abs(ck)), "weight rebuild mismatch"
    return fx, cfg, batch, sd


def state_template(cfg):
    """name -> zero tensor with the reference's state_dict layout, derived
    from the configuration alone (no reference import)."""
    p3, act = cfg["ptv3_config"], cfg["action_config"]
    mp = "pc_label_channels" in act and str(cfg.get("model_class", "")).startswith("MotionPlanner")
    t = {}

    def lin(n, o, i):
        t[n + ".weight"], t[n + ".bias"] = torch.zeros(o, i), torch.zeros(o)

    def norm(n, c, bn=False):
        t[n + ".weight"], t[n + ".bias"] = torch.zeros(c), torch.zeros(c)
        if bn:
            t[n + ".running_mean"], t[n + ".running_var"] = torch.zeros(c), torch.zeros(c)
            t[n + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    def block(n, c, h):
        t[n + ".cpe.0.weight"], t[n + ".cpe.0.bias"] = torch.zeros(c, 3, 3, 3, c), torch.zeros(c)
        lin(n + ".cpe.1", c, c); norm(n + ".cpe.2", c); norm(n + ".norm1.0", c)
        lin(n + ".attn.qkv", 3 * c, c); lin(n + ".attn.proj", c, c)
        norm(n + ".attn.q_norm", c // h); norm(n + ".attn.k_norm", c // h)
        norm(n + ".norm2.0", c); lin(n + ".mlp.0.fc1", 4 * c, c); lin(n + ".mlp.0.fc2", c, 4 * c)

    def ca(n, c, h, ctx):
        norm(n + ".norm1.0", c); lin(n + ".attn.q", c, c); lin(n + ".attn.kv", 2 * c, ctx)
        lin(n + ".attn.proj", c, c); norm(n + ".attn.q_norm", c // h); norm(n + ".attn.k_norm", c // h)
        norm(n + ".norm2.0", c); lin(n + ".mlp.0.fc1", 4 * c, c); lin(n + ".mlp.0.fc2", c, 4 * c)

    ec, eh = p3["enc_channels"], p3["enc_num_head"]
    dc, dh = list(p3["dec_channels"]) + [ec[-1]], p3["dec_num_head"]
    ctx = 256
    t["ptv3_model.embedding.stem.conv.weight"] = torch.zeros(
        ec[0], 5, 5, 5, p3["in_channels"] + (act["pc_label_channels"] if mp else 0))
    norm("ptv3_model.embedding.stem.norm", ec[0], True)
    for s in range(len(ec)):
        n = f"ptv3_model.enc.enc{s}"
        if s > 0:
            lin(n + ".down.proj", ec[s], ec[s - 1]); norm(n + ".down.norm.0", ec[s], True)
        for i in range(p3["enc_depths"][s]):
            block(n + f".block{i}", ec[s], eh[s]); ca(n + f".ca_block{i}", ec[s], eh[s], ctx)
    for s in reversed(range(len(ec) - 1)):
        n = f"ptv3_model.dec.dec{s}"
        lin(n + ".up.proj.0", dc[s], dc[s + 1]); norm(n + ".up.proj.1", dc[s], True)
        lin(n + ".up.proj_skip.0", dc[s], ec[s]); norm(n + ".up.proj_skip.1", dc[s], True)
        for i in range(p3["dec_depths"][s]):
            block(n + f".block{i}", dc[s], dh[s]); ca(n + f".ca_block{i}", dc[s], dh[s], ctx)
    lin("txt_fc", act["context_channels"], act["txt_ft_size"])
    if act.get("use_ee_pose"):   # base.py:52-60 (policy and motion planner)
        cc = act["context_channels"]
        t["pose_embedding.open_embedding.weight"] = torch.zeros(2, cc)
        lin("pose_embedding.pos_embedding", cc, 3); lin("pose_embedding.rot_embedding", cc, 6); norm("pose_embedding.layer_norm", cc)
    if not mp and act.get("use_step_id"):
        t["stepid_embedding.weight"] = torch.zeros(act["max_steps"], act["context_channels"])
    hs = dc[0]
    if mp:  # motion_planner_ptv3.py:165-185, :41-73
        t["pc_label_embedding.weight"] = torch.zeros(4, act["pc_label_channels"])
        lin("txt_attn_fc", 1, act["txt_ft_size"])
        t["act_proj_head.traj_embedding.weight"] = torch.zeros(act["max_traj_len"], act["traj_embed_size"])
        hi = hs + act["traj_embed_size"]
        lin("act_proj_head.heatmap_mlp.0", hs, hi); lin("act_proj_head.heatmap_mlp.3", 3 * act["pos_bins"] * 2, hs)
        lin("act_proj_head.action_mlp.0", hs, hi); lin("act_proj_head.action_mlp.3", 72 * 3 + 2, hs)
        return t
    lin("act_proj_head.heatmap_mlp.0", hs, hs); lin("act_proj_head.heatmap_mlp.3", 3 * act["pos_bins"] * 2, hs)
    lin("act_proj_head.action_mlp.0", hs, hs); lin("act_proj_head.action_mlp.3", 72 * 3 + 1, hs)
    return t


MP_CASES = ["mp_tiny_scaled_train", "mp_init_train", "mp_scaled_eval", "mp_tinyctx_scaled_train"]


def load_case_mp(name):
    """Motion-planner fixture (tests/golden/make_golden_mp.py): (fixture, cfg, batch, state_dict)."""
    fx = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    cfg = lcfg.preset(str(fx["meta_variant"]))
    batch = synth.synth_batch_mp(int(fx["meta_B"]), int(fx["meta_n"]), ragged=bool(fx["meta_ragged"]),
                                 seed=int(fx["meta_dseed"]))
    assert batch["npoints_in_batch"] == fx["npoints_in_batch"].tolist()
    ck = batch["pc_fts"].double().sum().item() + batch["pc_labels"].double().sum().item()
    assert abs(ck - float(fx["input_checksum"])) < 1e-9
    sd = seeded_state_dict(state_template(cfg), int(fx["meta_wseed"]), str(fx["meta_wvar"]))
    ck = sum(v.double().sum().item() for v in sd.values())
    assert abs(ck - float(fx["weight_checksum"])) < 1e-6 * max(1.0, abs(ck)), "weight rebuild mismatch"
    return fx, cfg, batch, sd

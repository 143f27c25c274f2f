"""Model configuration for the 3D-LOTUS hot path.

Mirrors the reference's config contract: `genrobo3d/configs/rlbench/simple_policy_ptv3.yaml`
(section MODEL) merged with `KEY VALUE` overrides (genrobo3d/configs/default.py:60-92).  yacs is
not available, so the YAML is read with PyYAML into an attribute dict with no-op
defrost()/freeze().  The published v1 model is defined by the CLI overrides of
job_scripts/train_3dlotus_policy.sh:61-87; they ship here as the preset `v1`.
"""
import ast
import copy


class Cfg(dict):
    """Attribute dict standing in for yacs.CfgNode (attribute access, .get, defrost/freeze)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def defrost(self):
        pass

    def freeze(self):
        pass


def to_cfg(d):
    if isinstance(d, dict):
        return Cfg({k: to_cfg(v) for k, v in d.items()})
    return d


# YAML defaults of MODEL (simple_policy_ptv3.yaml:92-158), restated as data.
_YAML_MODEL = dict(
    model_class="SimplePolicyPTV3AdaNorm",
    ptv3_config=dict(
        in_channels=6, order=["z", "z-trans", "hilbert", "hilbert-trans"], stride=[2, 2, 2, 2],
        enc_depths=[2, 2, 2, 6, 2], enc_channels=[32, 64, 128, 256, 512], enc_num_head=[2, 4, 8, 16, 32],
        enc_patch_size=[128] * 5, dec_depths=[2, 2, 2, 2], dec_channels=[64, 64, 128, 256],
        dec_num_head=[4, 4, 8, 16], dec_patch_size=[128] * 4, mlp_ratio=4, qkv_bias=True, qk_scale=None,
        qk_norm=False, scaled_cosine_attn=False, attn_drop=0.1, proj_drop=0.1, drop_path=0.1,
        pre_norm=True, shuffle_orders=True, enable_rpe=False, enable_flash=True, upcast_attention=False,
        upcast_softmax=False, cls_mode=False, pdnorm_bn=True, pdnorm_ln=True, pdnorm_decouple=False,
        pdnorm_adaptive=True, pdnorm_affine=True, pdnorm_conditions=None, pdnorm_only_decoder=False,
        add_coords_in_attn="none"),
    action_config=dict(
        voxel_size=0.01, context_channels=256, txt_ft_size=512, max_txt_len=77, txt_reduce="mean",
        use_ee_pose=True, use_step_id=False, max_steps=30, reduce="max", max_traj_len=1, dim_actions=8,
        pos_pred_type="heatmap_mlp", pos_heatmap_temp=0.1, rot_pred_type="euler", dropout=0.1,
        pos_bins=40, pos_bin_size=0.01, best_disc_pos="max"),
    loss_config=dict(pos_weight=1, rot_weight=1),
)

# job_scripts/train_3dlotus_policy.sh:61-87 (MODEL.* overrides only)
V1_OVERRIDES = [
    "ptv3_config.drop_path", "0.0", "ptv3_config.attn_drop", "0.1", "ptv3_config.proj_drop", "0.1",
    "action_config.dropout", "0.2", "action_config.voxel_size", "0.01", "action_config.reduce", "max",
    "action_config.dim_actions", "7", "action_config.rot_pred_type", "euler_disc",
    "action_config.pos_heatmap_temp", "0.1", "ptv3_config.in_channels", "7",
    "ptv3_config.pdnorm_only_decoder", "False", "ptv3_config.qk_norm", "True",
    "ptv3_config.scaled_cosine_attn", "False", "ptv3_config.enable_flash", "True",
    "action_config.max_steps", "30", "ptv3_config.enc_depths", "[1, 1, 1, 1, 1]",
    "ptv3_config.dec_depths", "[1, 1, 1, 1]", "ptv3_config.enc_channels", "[64, 128, 256, 512, 768]",
    "ptv3_config.dec_channels", "[128, 128, 256, 512]", "action_config.use_step_id", "False",
    "action_config.use_ee_pose", "False", "loss_config.pos_weight", "1", "loss_config.rot_weight", "1",
    "action_config.pos_pred_type", "heatmap_disc", "action_config.pos_bins", "15",
    "model_class", "SimplePolicyPTV3CA", "ptv3_config.pdnorm_bn", "False",
    "ptv3_config.pdnorm_ln", "False", "ptv3_config.pdnorm_adaptive", "False",
]

# BASELINE.json configs[0]: "3D-LOTUS tiny (2 layers, 64-dim, 512 pts/scene)"
TINY_OVERRIDES = V1_OVERRIDES + [
    "ptv3_config.enc_depths", "[1, 1]", "ptv3_config.enc_channels", "[64, 64]",
    "ptv3_config.enc_num_head", "[2, 2]", "ptv3_config.enc_patch_size", "[128, 128]",
    "ptv3_config.stride", "[2]", "ptv3_config.dec_depths", "[1]", "ptv3_config.dec_channels", "[64]",
    "ptv3_config.dec_num_head", "[2]", "ptv3_config.dec_patch_size", "[128]",
]


# the tiny width with stages deeper than one Block (the reference YAML default is enc_depths [2, 2, 2, 6, 2]): Block i of a
# stage attends along curve slot i % 4, so depth 5 wraps around the four curves
TINYDEEP_OVERRIDES = TINY_OVERRIDES + ["ptv3_config.enc_depths", "[2, 5]", "ptv3_config.dec_depths", "[2]"]


# tiny width with the two optional context tokens of SimplePolicyPTV3CA (simple_policy_ptv3.py:386-389,419-427): the current
# end-effector pose and the key-step index are appended to every cloud's instruction tokens
TINYCTX_OVERRIDES = TINY_OVERRIDES + ["action_config.use_ee_pose", "True", "action_config.use_step_id", "True"]


def _parse(v):
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return {"null": None, "true": True, "false": False}.get(v.lower(), v)


def merge_overrides(cfg, overrides):
    """yacs merge_from_list semantics: [KEY, VALUE, KEY, VALUE, ...] with dotted keys."""
    assert len(overrides) % 2 == 0
    for k, v in zip(overrides[0::2], overrides[1::2]):
        node = cfg
        parts = k.split(".")
        if parts[0] == "MODEL":
            parts = parts[1:]
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = _parse(v)
    return cfg


def load_model_config(yaml_path=None, overrides=()):
    """Read the MODEL section of a reference-format YAML (or the built-in defaults) and apply
    overrides.  Returns an attribute dict usable as `config.MODEL` of the reference trainer."""
    if yaml_path is not None:
        import yaml

        with open(yaml_path) as f:
            model = yaml.safe_load(f)["MODEL"]
    else:
        model = copy.deepcopy(_YAML_MODEL)
    return to_cfg(merge_overrides(model, list(overrides)))


# genrobo3d/configs/rlbench/motion_planner_ptv3.yaml:103-164 differs from the simple-policy YAML in these MODEL keys
_YAML_MP_DELTA = dict(model_class="MotionPlannerPTV3AdaNorm",
                      action_config=dict(pc_label_channels=64, traj_embed_size=64, max_traj_len=5,
                                         rot_pred_type="euler_disc"))

# job_scripts/train_3dlotusplus_motion_planner.sh:71-98 (MODEL.* overrides; pos_bin_size=15, max_traj_len=5)
MP_OVERRIDES = [
    "ptv3_config.drop_path", "0.0", "ptv3_config.attn_drop", "0.1", "ptv3_config.proj_drop", "0.1",
    "action_config.dropout", "0.2", "action_config.voxel_size", "0.01", "action_config.reduce", "max",
    "action_config.dim_actions", "7", "action_config.rot_pred_type", "euler_disc",
    "action_config.pos_pred_type", "heatmap_disc", "action_config.pos_heatmap_temp", "0.1",
    "ptv3_config.in_channels", "4", "ptv3_config.pdnorm_only_decoder", "False", "ptv3_config.qk_norm", "True",
    "ptv3_config.scaled_cosine_attn", "False", "ptv3_config.enable_flash", "True",
    "action_config.max_steps", "30", "ptv3_config.enc_depths", "[1, 1, 1, 1, 1]",
    "ptv3_config.dec_depths", "[1, 1, 1, 1]", "ptv3_config.enc_channels", "[64, 128, 256, 512, 768]",
    "ptv3_config.dec_channels", "[128, 128, 256, 512]", "loss_config.pos_weight", "1",
    "loss_config.rot_weight", "1", "action_config.max_traj_len", "5", "action_config.pos_bins", "15",
    "action_config.txt_reduce", "attn", "action_config.use_ee_pose", "False",
    "model_class", "MotionPlannerPTV3CA", "ptv3_config.pdnorm_bn", "False",
    "ptv3_config.pdnorm_ln", "False", "ptv3_config.pdnorm_adaptive", "False",
]
MP_TINY_OVERRIDES = MP_OVERRIDES + TINY_OVERRIDES[len(V1_OVERRIDES):]
# the YAML's own use_ee_pose = True (motion_planner_ptv3.yaml:151; the published job script switches it off)
MP_TINYCTX_OVERRIDES = MP_TINY_OVERRIDES + ["action_config.use_ee_pose", "True"]


# job_scripts/train_3dlotus_policy_peract.sh:55-75: the v1 model; `txt_reduce attn` is set but SimplePolicyPTV3CA ignores it
PERACT_OVERRIDES = V1_OVERRIDES + ["action_config.txt_reduce", "attn"]


def preset(name="v1"):
    """'v1' / 'tiny': 3D-LOTUS policy; 'peract': the RLBench-18task (PerAct) variant of BASELINE configs[4] (same network;
    its bf16 compute mode is ops.set_gemm_precision("bf16")); 'mp' / 'mp_tiny': 3D-LOTUS++ motion planner (configs[3])."""
    if name in ("mp", "mp_tiny", "mp_tinyctx"):
        model = copy.deepcopy(_YAML_MODEL)
        model["model_class"] = _YAML_MP_DELTA["model_class"]
        model["action_config"].update(_YAML_MP_DELTA["action_config"])
        return to_cfg(merge_overrides(model, {"mp": MP_OVERRIDES, "mp_tiny": MP_TINY_OVERRIDES, "mp_tinyctx": MP_TINYCTX_OVERRIDES}[name]))
    return load_model_config(None, {"v1": V1_OVERRIDES, "tiny": TINY_OVERRIDES, "peract": PERACT_OVERRIDES,
                                    "tinydeep": TINYDEEP_OVERRIDES, "tinyctx": TINYCTX_OVERRIDES}[name])


def plain(cfg):
    """dict(ptv3=..., action=..., loss=...) view used by the oracle and the kernels' host code."""
    return dict(ptv3=dict(cfg["ptv3_config"]), action=dict(cfg["action_config"]), loss=dict(cfg["loss_config"]))

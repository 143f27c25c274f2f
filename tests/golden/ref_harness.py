"""Import harness for the *reference* (vlc-robot/robot-3dlotus) — build container only.

TEST INFRASTRUCTURE.  This file never travels as product code and is never imported by
the package, by `-m gpu` tests, by smoke() or by bench.py: it needs /root/reference, which
does not exist on the GPU box.  It is used by `tests/golden/make_golden.py` (fixture
generator) and by the container-only tests in `tests/test_oracle_vs_reference.py`.

The reference's three native dependencies (spconv, flash_attn, torch_scatter) and four
pure-python ones (addict, easydict, timm, yacs) are not installed and there is no network
(SURVEY.md §8c).  Stand-in modules are injected into `sys.modules` *before* the reference is
imported, following SURVEY.md Appendix D:

  addict.Dict / easydict.EasyDict   attribute dict
  timm.models.layers                DropPath, trunc_normal_
  torch_scatter.segment_csr         scatter_reduce(include_self=False)
  spconv.pytorch                    SparseConvTensor container + SubMConv3d (pure torch, sorted-key
                                    neighbour lookup, weight (Cout,k,k,k,Cin), taps x-major over
                                    indices[:,1:4], duplicates -> lowest index)
  flash_attn                        per-segment fp32 softmax attention honouring cu_seqlens;
                                    result cast to the input dtype (fp16 in the reference unless
                                    `neutralise_half()` is active)

None of these stand-ins is a copy of third-party source; they restate the published semantics
of the call sites listed in SURVEY.md §2.1.
"""
import contextlib
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


# ----------------------------------------------------------------------------- attr dicts
class _AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]


class _Cfg(_AttrDict):
    """yacs-CfgNode stand-in: attribute access, .get, no-op defrost/freeze."""

    def defrost(self):
        pass

    def freeze(self):
        pass


def to_cfg(d):
    if isinstance(d, dict):
        return _Cfg({k: to_cfg(v) for k, v in d.items()})
    return d


# ----------------------------------------------------------------------------- timm
class _DropPath(nn.Module):
    def __init__(self, p=0.0):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        keep = 1 - self.p
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask / keep


# ----------------------------------------------------------------------------- torch_scatter
def _segment_csr(src, indptr, out=None, reduce="sum"):
    counts = indptr[1:] - indptr[:-1]
    nseg = counts.numel()
    seg = torch.repeat_interleave(torch.arange(nseg, device=src.device), counts)
    idx = seg.view(-1, *([1] * (src.ndim - 1))).expand_as(src)
    red = {"sum": "sum", "mean": "mean", "max": "amax", "min": "amin"}[reduce]
    res = src.new_zeros((nseg,) + tuple(src.shape[1:]))
    return res.scatter_reduce(0, idx, src, reduce=red, include_self=False)


# ----------------------------------------------------------------------------- spconv
class _SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, _cache=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {} if _cache is None else _cache

    def replace_feature(self, f):
        return _SparseConvTensor(f, self.indices, self.spatial_shape, self.batch_size, self.indice_dict)


def _neighbour_table(indices, ksize):
    """nbr[N, k^3] (int64, -1 = absent).  Tap order: x-major over indices[:,1:4]."""
    idx = indices.long()
    n = idx.shape[0]
    b, x, y, z = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
    S = int(max(x.max(), y.max(), z.max())) + ksize + 2
    r = ksize // 2

    def key(bb, xx, yy, zz):
        return ((bb * S + (xx + r)) * S + (yy + r)) * S + (zz + r)

    keys = key(b, x, y, z)
    # duplicates -> lowest index wins: stable sort, then keep first of each run
    skeys, sidx = torch.sort(keys, stable=True)
    first = torch.ones_like(skeys, dtype=torch.bool)
    first[1:] = skeys[1:] != skeys[:-1]
    ukeys, uidx = skeys[first], sidx[first]
    taps = []
    for dx in range(-r, r + 1):
        for dy in range(-r, r + 1):
            for dz in range(-r, r + 1):
                q = key(b, x + dx, y + dy, z + dz)
                pos = torch.searchsorted(ukeys, q).clamp(max=ukeys.numel() - 1)
                hit = ukeys[pos] == q
                taps.append(torch.where(hit, uidx[pos], torch.full_like(pos, -1)))
    return torch.stack(taps, 1)


class _SubMConv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, algo=None):
        super().__init__()
        k = kernel_size
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, k
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, k, k, k, in_channels))
        # spconv keeps its own init (kaiming-uniform style); any finite init does for parity work
        bound = 1.0 / (in_channels * k ** 3) ** 0.5
        nn.init.uniform_(self.weight, -bound, bound)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        cache_key = (self.indice_key, self.kernel_size)
        nbr = x.indice_dict.get(cache_key) if self.indice_key is not None else None
        if nbr is None:
            nbr = _neighbour_table(x.indices, self.kernel_size)
            if self.indice_key is not None:
                x.indice_dict[cache_key] = nbr
        f = x.features
        w = self.weight.reshape(self.out_channels, -1, self.in_channels)
        out = f.new_zeros(f.shape[0], self.out_channels)
        for t in range(nbr.shape[1]):
            col = nbr[:, t]
            rows = torch.nonzero(col >= 0).squeeze(1)
            if rows.numel() == 0:
                continue
            out = out.index_add(0, rows, f[col[rows]] @ w[:, t, :].t())
        if self.bias is not None:
            out = out + self.bias
        return x.replace_feature(out)


# ----------------------------------------------------------------------------- flash_attn
def _attn_fp32(q, k, v, scale):
    # q (Lq,H,d) k,v (Lk,H,d)
    s = torch.einsum("qhd,khd->hqk", q.float(), k.float()) * scale
    p = torch.softmax(s, dim=-1)
    return torch.einsum("hqk,khd->qhd", p, v.float())


def _flash_qkvpacked(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, **kw):
    assert dropout_p == 0, "stand-in supports dropout_p == 0 only (use eval() or zero dropouts)"
    out = qkv.new_empty(qkv.shape[0], qkv.shape[2], qkv.shape[3])
    cu = cu_seqlens.tolist()
    for a, b in zip(cu[:-1], cu[1:]):
        assert b - a <= max_seqlen
        out[a:b] = _attn_fp32(qkv[a:b, 0], qkv[a:b, 1], qkv[a:b, 2], softmax_scale).to(qkv.dtype)
    return out


def _flash_kvpacked(q, kv, cu_q, cu_k, max_q, max_k, dropout_p=0.0, softmax_scale=None, **kw):
    assert dropout_p == 0
    out = torch.empty_like(q)
    cq, ck = cu_q.tolist(), cu_k.tolist()
    for i in range(len(cq) - 1):
        a, b, c, d = cq[i], cq[i + 1], ck[i], ck[i + 1]
        out[a:b] = _attn_fp32(q[a:b], kv[c:d, 0], kv[c:d, 1], softmax_scale).to(q.dtype)
    return out


def install_shims():
    if "spconv" in sys.modules and getattr(sys.modules["spconv"], "_lotus_shim", False):
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("addict", Dict=_AttrDict)
    mod("easydict", EasyDict=_AttrDict)
    layers = mod("timm.models.layers", DropPath=_DropPath, trunc_normal_=torch.nn.init.trunc_normal_)
    models = mod("timm.models", layers=layers)
    mod("timm", models=models)
    mod("torch_scatter", segment_csr=_segment_csr)
    modules = mod("spconv.pytorch.modules", is_spconv_module=lambda m: isinstance(m, _SubMConv3d))
    sp = mod("spconv.pytorch", SubMConv3d=_SubMConv3d, SparseConvTensor=_SparseConvTensor, modules=modules)
    top = mod("spconv", pytorch=sp)
    top._lotus_shim = True
    mod("flash_attn", flash_attn_varlen_qkvpacked_func=_flash_qkvpacked,
        flash_attn_varlen_kvpacked_func=_flash_kvpacked)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


@contextlib.contextmanager
def neutralise_half():
    """fp32-ideal flavour: Tensor.half() becomes the identity."""
    orig = torch.Tensor.half
    torch.Tensor.half = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.half = orig


@contextlib.contextmanager
def record_randperm(store):
    """Record every torch.randperm draw (the 5 shuffle_orders permutations, Trap 4)."""
    orig = torch.randperm

    def wrapped(*a, **k):
        p = orig(*a, **k)
        store.append(p.clone())
        return p

    torch.randperm = wrapped
    try:
        yield
    finally:
        torch.randperm = orig


def reference_model_config(variant="v1"):
    """Reference YAML (simple_policy_ptv3.yaml) + the CLI overrides of
    job_scripts/train_3dlotus_policy.sh:61-87 (variant 'v1') or the tiny variant 'tiny'."""
    import yaml

    with open(f"{REFERENCE_ROOT}/genrobo3d/configs/rlbench/simple_policy_ptv3.yaml") as f:
        cfg = yaml.safe_load(f)["MODEL"]
    p, a = cfg["ptv3_config"], cfg["action_config"]
    cfg["model_class"] = "SimplePolicyPTV3CA"
    p.update(drop_path=0.0, attn_drop=0.1, proj_drop=0.1, in_channels=7, pdnorm_only_decoder=False,
             qk_norm=True, scaled_cosine_attn=False, enable_flash=True,
             enc_depths=[1, 1, 1, 1, 1], dec_depths=[1, 1, 1, 1],
             enc_channels=[64, 128, 256, 512, 768], dec_channels=[128, 128, 256, 512],
             pdnorm_bn=False, pdnorm_ln=False, pdnorm_adaptive=False)
    a.update(dropout=0.2, voxel_size=0.01, reduce="max", dim_actions=7, rot_pred_type="euler_disc",
             pos_heatmap_temp=0.1, max_steps=30, use_step_id=False, use_ee_pose=False,
             pos_pred_type="heatmap_disc", pos_bins=15)
    cfg["loss_config"].update(pos_weight=1, rot_weight=1)
    if variant == "tinyctx":  # the two optional context tokens (end-effector pose, key-step index)
        a.update(use_ee_pose=True, use_step_id=True)
    if variant in ("tiny", "tinydeep", "tinyctx"):
        p.update(enc_depths=[1, 1], enc_channels=[64, 64], enc_num_head=[2, 2], enc_patch_size=[128, 128],
                 stride=[2], dec_depths=[1], dec_channels=[64], dec_num_head=[2], dec_patch_size=[128])
    if variant == "tinydeep":  # stages deeper than one Block (order_index = i % 4 wraps at depth 5)
        p.update(enc_depths=[2, 5], dec_depths=[2])
    # PointTransformerV3CA.__init__ does not accept these two keys of the YAML
    for k in ("pdnorm_only_decoder",):
        pass
    return to_cfg(cfg)


def build_reference_policy(variant="v1", drop_path=0.0, enable_flash=True):
    """enable_flash=False: the reference's OWN attention arithmetic — the padded-patch softmax branch of SerializedAttention
    (PointTransformerV3/model.py:499-527) and the padded einsum branch of the cross attention (model_ca.py:68-95) — instead of
    its flash_attn calls, i.e. without this file's flash_attn stand-in anywhere on the path."""
    install_shims()
    from genrobo3d.models.simple_policy_ptv3 import SimplePolicyPTV3CA

    cfg = reference_model_config(variant)
    cfg["ptv3_config"]["drop_path"] = drop_path
    cfg["ptv3_config"]["enable_flash"] = bool(enable_flash)
    return SimplePolicyPTV3CA(cfg), cfg


def reference_forward(ref, batch, full=True):
    """Run the reference hot path.  full=True calls SimplePolicyPTV3CA.forward itself
    (simple_policy_ptv3.py:225-306); full=False re-wires the same calls without the hard-coded
    `point_outs[k] for k in [0..4]` (:243), which requires exactly five stages and therefore
    cannot run BASELINE's two-stage tiny configuration."""
    if full:
        acts, losses = ref(batch, compute_loss=True, compute_final_action=False)
        return losses
    batch = ref.prepare_batch(batch)
    point_outs = ref.ptv3_model(ref.prepare_ptv3_batch(batch), return_dec_layers=True)
    pred = ref.act_proj_head(point_outs[-1].feat, batch["npoints_in_batch"], coords=point_outs[-1].coord,
                             temp=1, gt_pos=batch["gt_actions"][..., :3], dec_layers_embed=None)
    return ref.compute_loss(pred, batch["gt_actions"], disc_pos_probs=batch.get("disc_pos_probs"),
                            npoints_in_batch=batch["npoints_in_batch"])


# ----------------------------------------------------------------------------- 3D-LOTUS++ motion planner
def reference_mp_config(variant="mp"):
    """Reference YAML (motion_planner_ptv3.yaml) + the MODEL overrides of
    job_scripts/train_3dlotusplus_motion_planner.sh:71-98 (pos_bin_size=15, max_traj_len=5); 'mp_tiny' adds the
    two-stage geometry of BASELINE configs[0]."""
    import yaml

    with open(f"{REFERENCE_ROOT}/genrobo3d/configs/rlbench/motion_planner_ptv3.yaml") as f:
        cfg = yaml.safe_load(f)["MODEL"]
    p, a = cfg["ptv3_config"], cfg["action_config"]
    cfg["model_class"] = "MotionPlannerPTV3CA"
    p.update(drop_path=0.0, attn_drop=0.1, proj_drop=0.1, in_channels=4, pdnorm_only_decoder=False,
             qk_norm=True, scaled_cosine_attn=False, enable_flash=True,
             enc_depths=[1, 1, 1, 1, 1], dec_depths=[1, 1, 1, 1],
             enc_channels=[64, 128, 256, 512, 768], dec_channels=[128, 128, 256, 512],
             pdnorm_bn=False, pdnorm_ln=False, pdnorm_adaptive=False)
    a.update(dropout=0.2, voxel_size=0.01, reduce="max", dim_actions=7, rot_pred_type="euler_disc",
             pos_pred_type="heatmap_disc", pos_heatmap_temp=0.1, max_steps=30, max_traj_len=5, pos_bins=15,
             txt_reduce="attn", use_ee_pose=False)
    cfg["loss_config"].update(pos_weight=1, rot_weight=1)
    if variant in ("mp_tiny", "mp_tinyctx"):
        p.update(enc_depths=[1, 1], enc_channels=[64, 64], enc_num_head=[2, 2], enc_patch_size=[128, 128],
                 stride=[2], dec_depths=[1], dec_channels=[64], dec_num_head=[2], dec_patch_size=[128])
    if variant == "mp_tinyctx":   # the YAML's own use_ee_pose = True (motion_planner_ptv3.yaml:151)
        a.update(use_ee_pose=True)
    return to_cfg(cfg)


def build_reference_mp(variant="mp", enable_flash=True):
    """enable_flash=False: the reference's own attention arithmetic instead of its flash_attn calls (see build_reference_policy)."""
    install_shims()
    if "einops" not in sys.modules:
        import einops  # noqa: F401  (installed in this image)
    from genrobo3d.models.motion_planner_ptv3 import MotionPlannerPTV3CA

    cfg = reference_mp_config(variant)
    cfg["ptv3_config"]["enable_flash"] = bool(enable_flash)
    return MotionPlannerPTV3CA(cfg), cfg


def reference_forward_mp(ref, batch):
    """MotionPlannerPTV3AdaNorm.forward itself (motion_planner_ptv3.py:222-305), training-step call."""
    acts, losses = ref(batch, compute_loss=True, compute_final_action=False)
    return losses

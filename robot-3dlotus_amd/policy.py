"""SimplePolicyPTV3CA — drop-in for `genrobo3d/models/simple_policy_ptv3.py:376-431` on MI355X.

Module surface kept from the reference: `cls(config.MODEL)`,
`forward(batch, compute_loss=False, **kwargs)` returning `final_pred_actions` or
`(final_pred_actions, losses{pos,rot,open,total})`, kwarg `compute_final_action`, properties
`num_parameters` / `num_trainable_parameters`, and the 460-entry state_dict (Appendix B).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .config import to_cfg
from .ptv3 import PointTransformerV3CA

_PTV3_KEYS = ("in_channels", "order", "stride", "enc_depths", "enc_channels", "enc_num_head", "enc_patch_size",
              "dec_depths", "dec_channels", "dec_num_head", "dec_patch_size", "mlp_ratio", "qkv_bias", "qk_scale",
              "qk_norm", "attn_drop", "proj_drop", "drop_path", "pre_norm", "shuffle_orders", "enable_rpe",
              "enable_flash", "upcast_attention", "upcast_softmax", "cls_mode", "pdnorm_bn", "pdnorm_ln",
              "pdnorm_decouple", "pdnorm_adaptive", "pdnorm_affine", "pdnorm_conditions", "pdnorm_only_decoder",
              "add_coords_in_attn", "scaled_cosine_attn", "ctx_channels", "pdnorm_context_channels")


class BaseModel(nn.Module):
    """genrobo3d/models/base.py:10-49"""

    @property
    def num_parameters(self):
        ps = list(self.parameters())
        return sum(int(np.prod(p.size())) for p in ps), len(ps)

    @property
    def num_trainable_parameters(self):
        ps = [p for p in self.parameters() if p.requires_grad]
        return sum(int(np.prod(p.size())) for p in ps), len(ps)

    def prepare_batch(self, batch):
        device = next(self.parameters()).device
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                # pinned host tensors (data.ptv3_collate_fn(pin=True)) upload asynchronously on the current stream
                batch[k] = v.to(device, non_blocking=v.device.type == "cpu" and v.is_pinned())
        return batch

    @staticmethod
    def append_context_tokens(ctx, ctx_counts, tok):
        """One extra context token per cloud and entry of `tok` ([B, C] each), appended to that cloud's instruction tokens
        (simple_policy_ptv3.py:419-427, motion_planner_ptv3.py:451-457): rows are placed with one index_copy per source
        instead of B small cats.  Returns (context, counts)."""
        if not tok:
            return ctx, ctx_counts
        B, ne = len(ctx_counts), len(tok)
        starts = np.concatenate([[0], np.cumsum(np.asarray(ctx_counts) + ne)])[:-1]
        txt_pos = np.concatenate([starts[b] + np.arange(ctx_counts[b]) for b in range(B)])
        out = ctx.new_empty(int(sum(ctx_counts)) + B * ne, ctx.shape[1])
        out = out.index_copy(0, torch.from_numpy(txt_pos).to(ctx.device), ctx)
        for j, t in enumerate(tok):
            pos_j = torch.from_numpy(starts + np.asarray(ctx_counts) + j).to(ctx.device)
            out = out.index_copy(0, pos_j, t.to(ctx.dtype))
        return out, [c + ne for c in ctx_counts]

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, nn.Embedding):
            nn.init.trunc_normal_(m.weight, std=0.02)


class ActionHead(nn.Module):
    """Parameter layout of simple_policy_ptv3.py:19-68 for (max, heatmap_disc, euler_disc)."""

    def __init__(self, reduce, pos_pred_type, rot_pred_type, hidden_size, dim_actions, dropout=0, voxel_size=0.01,
                 euler_resolution=5, ptv3_config=None, pos_bins=50):
        super().__init__()
        if (reduce, pos_pred_type, rot_pred_type) != ("max", "heatmap_disc", "euler_disc"):
            raise NotImplementedError("lotus-hip builds the published head: reduce=max, heatmap_disc, euler_disc")
        self.euler_resolution, self.euler_bins, self.pos_bins = euler_resolution, 360 // euler_resolution, pos_bins
        self.dropout = float(dropout)
        self.heatmap_mlp = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.LeakyReLU(0.02), nn.Dropout(dropout),
                                         nn.Linear(hidden_size, 3 * pos_bins * 2))
        self.action_mlp = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.LeakyReLU(0.02), nn.Dropout(dropout),
                                        nn.Linear(hidden_size, self.euler_bins * 3 + 1))


class RobotPoseEmbedding(nn.Module):
    """genrobo3d/models/base.py:52-78: LayerNorm(Linear(xyz) + Linear(sin / cos of the xyz Euler angles) + Embedding(open)), one
    context token per cloud.  [B, 256]-sized: plain torch arithmetic (the Euler angles come from scipy on the host, as in the
    reference)."""

    def __init__(self, hidden_size):
        super().__init__()
        self.open_embedding = nn.Embedding(2, hidden_size)
        self.pos_embedding = nn.Linear(3, hidden_size)
        self.rot_embedding = nn.Linear(6, hidden_size)
        self.layer_norm = nn.LayerNorm(hidden_size, eps=1e-12)

    def forward(self, actions):
        from scipy.spatial.transform import Rotation as R

        pos = self.pos_embedding(actions[..., :3])
        opn = self.open_embedding(actions[..., -1].long())
        eul = torch.from_numpy(R.from_quat(actions[..., 3:7].detach().cpu().numpy()).as_euler("xyz")).float().to(actions.device)
        rot = self.rot_embedding(torch.cat([torch.sin(eul), torch.cos(eul)], -1))
        return self.layer_norm(pos + rot + opn)


class SimplePolicyPTV3CA(BaseModel):
    def __init__(self, config):
        super().__init__()
        config = to_cfg(config)
        self.config = config
        p3 = {k: v for k, v in config.ptv3_config.items() if k in _PTV3_KEYS}
        self.ptv3_model = PointTransformerV3CA(**p3)
        act = config.action_config
        self.txt_fc = nn.Linear(act.txt_ft_size, act.context_channels)
        if act.use_ee_pose:     # simple_policy_ptv3.py:386-389 (the published v1 model sets both to False)
            self.pose_embedding = RobotPoseEmbedding(act.context_channels)
        if act.use_step_id:
            self.stepid_embedding = nn.Embedding(act.max_steps, act.context_channels)
        self.act_proj_head = ActionHead(act.reduce, act.pos_pred_type, act.rot_pred_type,
                                        config.ptv3_config.dec_channels[0], act.dim_actions, dropout=act.dropout,
                                        voxel_size=act.voxel_size, pos_bins=act.pos_bins)
        self.apply(self._init_weights)

    # -- reference API ---------------------------------------------------------------------
    def prepare_ptv3_batch(self, batch):
        """simple_policy_ptv3.py:403-431 (context = txt_fc(txt_embeds), one segment per cloud)."""
        txt, feat = batch["txt_embeds"].contiguous(), batch["pc_fts"]
        extra = {}
        if self.act_storage == "bf16":
            # bf16 activation storage: the network inputs are rounded once here (what autocast does to the first
            # layers' inputs); the integer front end keeps reading the fp32 coordinates of pc_fts
            txt, feat = txt.to(torch.bfloat16), feat.to(torch.bfloat16)
            extra["coord_src"] = batch["pc_fts"]
        ctx = ops.LinearFn.apply(txt, self.txt_fc.weight, self.txt_fc.bias)
        ctx_counts = list(batch["txt_lens"])
        act = self.config.action_config
        tok = []
        if act.use_ee_pose:
            tok.append(self.pose_embedding(batch["ee_poses"].float()))
        if act.use_step_id:
            tok.append(self.stepid_embedding(batch["step_ids"].long()))
        ctx, ctx_counts = self.append_context_tokens(ctx, ctx_counts, tok)
        return {"coord": batch["pc_fts"][:, :3], "grid_size": self.config.action_config.voxel_size,
                "offset": batch["offset"], "feat": feat, "context": ctx,
                "counts": list(batch["npoints_in_batch"]), "context_counts": ctx_counts, **extra}

    @torch.no_grad()
    def prefetch(self, batch):
        """Optional input-pipeline hook (not in the reference): start the integer front-end of `batch` on a side
        stream.  Call it for the NEXT batch BEFORE the forward of the current one (`prefetch(b[k + 1]); forward(b[k])`:
        the pipeline then runs under forward k, see PointTransformerV3CA.prefetch) or right after it; forward(batch) must
        later get the same dict.  Purely an overlap device — results are identical."""
        on_host = any(isinstance(v, torch.Tensor) and v.device.type == "cpu" for v in batch.values())
        if on_host and self.ptv3_model._pending is not None:
            # two batches ahead (see PointTransformerV3CA.prefetch): upload and pipeline are both issued by the forward of
            # the batch in between, behind ITS tables on the front-end stream
            self.ptv3_model._deferred = lambda: self.prefetch(batch)
            return
        if on_host:
            # a host batch (pinned tensors of data.ptv3_collate_fn(pin=True)): upload it on the front-end stream, so
            # the H2D copies (24 MB of soft labels per 16 clouds) and the integer pipeline both run under the current
            # step's backward instead of in front of the next forward (genrobo3d/models/base.py:29-34 uploads
            # synchronously at the top of forward)
            fe = self.ptv3_model.fe_stream()
            with torch.cuda.stream(fe):
                batch = self.prepare_batch(batch)
            batch["_upload_stream"] = fe
        else:
            batch = self.prepare_batch(batch)
        if not batch["pc_fts"].is_contiguous():
            batch["pc_fts"] = batch["pc_fts"].contiguous()
        self.ptv3_model.prefetch({"coord": batch["pc_fts"][:, :3], "grid_size": self.config.action_config.voxel_size,
                                  "offset": batch["offset"], "feat": batch["pc_fts"], "coord_src": batch["pc_fts"],
                                  "counts": list(batch["npoints_in_batch"]), "context_counts": list(batch["txt_lens"])},
                                 wait_current=not on_host)

    gemm_precision = None  # 'fp32' | 'bf16x3' | 'bf16': operand precision of THIS model's products (None = ops default)
    # None / 'fp32': activations are stored in fp32 (the parity path).  'bf16': every activation tensor in HBM is bf16,
    # parameters, their gradients, statistics and accumulation stay fp32, products are bf16 MFMA — the configuration of
    # BASELINE configs[4] (RLBench-18 / PerAct, bf16); runs the lotus_b16_* twins of the C-ABI (include/lotus_hip_b16.h)
    act_storage = None

    # bf16 storage only: the dense layers of the backbone read bf16 SHADOWS of their fp32 master weights (ops.WeightShadows;
    # refreshed by the fused AdamW in the launch that updates the masters) — "bf16 activations / weights with fp32 master
    # weights", job_scripts/train_3dlotus_policy_peract.sh:42-44,61.  False keeps converting the fp32 masters per block.
    weight_shadows = os.environ.get("LOTUS_W_SHADOW", "1") != "0"

    def _shadows(self):
        ws = getattr(self, "_weight_shadow_set", None)
        if ws is None:
            ws = ops.WeightShadows([m.weight for m in self.ptv3_model.modules() if isinstance(m, nn.Linear)])
            object.__setattr__(self, "_weight_shadow_set", ws)
        ws.refresh()
        return ws

    def forward(self, batch, compute_loss=False, **kwargs):
        if self.act_storage == "bf16":
            if self.weight_shadows:
                self._shadows()
            with ops.storage(torch.bfloat16, shadows=self.weight_shadows):
                return self._forward(batch, compute_loss, **kwargs)
        with ops.precision(self.gemm_precision):
            return self._forward(batch, compute_loss, **kwargs)

    def _forward(self, batch, compute_loss=False, **kwargs):
        batch = self.prepare_batch(batch)
        up = batch.pop("_upload_stream", None)
        if up is not None:  # uploaded by prefetch() on the front-end stream: order this stream after it, tell the allocator
            cur = torch.cuda.current_stream()
            cur.wait_stream(up)
            for v in batch.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
        dev = batch["pc_fts"].device
        if dev.type != "cuda":
            raise RuntimeError("lotus-hip runs on a HIP device only (no CPU fallback); move the model and batch to cuda")
        act, head = self.config.action_config, self.act_proj_head
        outs = self.ptv3_model(self.prepare_ptv3_batch(batch), return_dec_layers=True)
        last = outs[-1]
        lvl = last.level
        B = len(lvl.counts)
        gt = batch["gt_actions"].float().contiguous() if "gt_actions" in batch else None
        tgt = None
        with_loss = bool(compute_loss)
        if with_loss:
            dp = batch.get("disc_pos_probs")
            if dp is None:
                # no host-made soft labels in the batch: build them on the device from the ground-truth positions
                # (get_disc_gt_pos_prob, utils/action_position_utils.py:7-46; the dataset would otherwise ship
                # 3 * n * 2 * pos_bins floats per cloud over PCIe).  Options of the reference dataset
                # (simple_policy_dataset.py:41-42): batch["pos_heatmap_type"] 'plain' | 'dist', and
                # batch["robot_point_mask"] (bool [N]) for pos_heatmap_no_robot.
                pc = batch["pc_fts"] if batch["pc_fts"].stride(1) == 1 else batch["pc_fts"].contiguous()
                tgt = ops.pos_targets(pc, lvl.off, lvl.batch, gt, 2 * head.pos_bins, act.pos_bin_size,
                                      batch.get("pos_heatmap_type", "plain"), batch.get("robot_point_mask"))
            else:
                tgt = dp if isinstance(dp, torch.Tensor) else torch.cat([t.reshape(-1) for t in dp]).to(dev)
                tgt = tgt.float().contiguous()
        hm, am = head.heatmap_mlp, head.action_mlp
        p = head.dropout if self.training else 0.0
        lc = self.config.loss_config
        dummy = last.feat.new_zeros(1)
        losses, xt, ae = ops.HeadLossFn.apply(
            last.feat, hm[0].weight, hm[0].bias, hm[3].weight, hm[3].bias, am[0].weight, am[0].bias, am[3].weight,
            am[3].bias, lvl, tgt if with_loss else dummy, gt if gt is not None else dummy.view(1, 1),
            float(lc.pos_weight), float(lc.rot_weight), p, ops.mix_seed(self.ptv3_model.last_seed, 1000), with_loss)
        nb = 2 * head.pos_bins
        pred_pos = xt.view(-1, 3, nb).permute(1, 0, 2)          # (3, N, 2*pos_bins) like the reference
        pred_rot = ae[:, :head.euler_bins * 3].view(B, head.euler_bins, 3)
        pred_open = ae[:, -1]
        self.last_pred = (pred_pos, pred_rot, pred_open)

        decode = kwargs.get("compute_final_action", True)
        if compute_loss and self.training and not decode and not kwargs.get("decode_actions", False):
            # Training step of the reference trainer (`_, losses = model(batch, compute_loss=True,
            # compute_final_action=False)`, train_simple_policy.py:211): the action tuple is discarded, so
            # the device->host argmax copy + per-sample scipy decode (simple_policy_ptv3.py:292-296) is
            # skipped unless decode_actions=True is passed.  See INTEGRATION.md.
            return None, {"pos": losses[0], "rot": losses[1], "open": losses[2], "total": losses[3]}
        if decode:
            pc = batch["pc_fts"] if batch["pc_fts"].stride(1) == 1 else batch["pc_fts"].contiguous()
            best = act.get("best_disc_pos", "max")   # simple_policy_ptv3.py:266 (set by the evaluation scripts)
            if best == "ens1":
                pos = ops.pos_decode_ens1(xt, pc, list(batch["npoints_in_batch"]), nb, act.pos_bin_size)
            elif best == "max":
                pos = ops.pos_decode_max(xt, pc, lvl.off, B, nb, act.pos_bin_size)  # f64 [B, 3], one launch pair
            else:
                raise ValueError(f"best_disc_pos must be 'max' or 'ens1', got {best!r}")
        else:
            pos = gt[..., :3]
        # euler_disc decode, simple_policy_ptv3.py:292-296 (float64 on purpose)
        from scipy.spatial.transform import Rotation as R
        rot_bins = torch.argmax(pred_rot, 1).cpu().numpy()
        quat = np.stack([R.from_euler("xyz", x * head.euler_resolution - 180, degrees=True).as_quat() for x in rot_bins], 0)
        final = torch.cat([pos.double(), torch.from_numpy(quat).to(dev), pred_open.detach().double().unsqueeze(-1)], -1)
        if compute_loss:
            return final, {"pos": losses[0], "rot": losses[1], "open": losses[2], "total": losses[3]}
        return final




def _factory():
    from .motion_planner import MotionPlannerPTV3CA
    return {"SimplePolicyPTV3CA": SimplePolicyPTV3CA, "MotionPlannerPTV3CA": MotionPlannerPTV3CA}


class _Factory(dict):
    """name -> class table of genrobo3d/train/train_simple_policy.py:46-50 / train_motion_planner.py (lazy: the
    motion planner imports this module)."""

    def __missing__(self, k):
        self.update(_factory())
        return dict.__getitem__(self, k)


MODEL_FACTORY = _Factory({"SimplePolicyPTV3CA": SimplePolicyPTV3CA})

"""MotionPlannerPTV3CA — the 3D-LOTUS++ motion planner (BASELINE configs[3]) on the lotus-hip kernels; drop-in for
`genrobo3d/models/motion_planner_ptv3.py:400-463` (+ forward / compute_loss of its base class, :222-397).

Same backbone as the policy (68 input channels: xyz + height + a 64-d embedding of the 4 point labels), a trajectory
head that scores `max_traj_len` future steps at once, and five losses (pos, rot, open, stop, total) masked by the
per-sample trajectory length.  Module surface kept: `cls(config.MODEL)`, `forward(batch, compute_loss=False,
**kwargs)` -> `final_pred_actions [B, T, 3+4+2]` or `(final_pred_actions, losses)`, kwarg `compute_final_action`,
the reference's state_dict keys (incl. the unused `txt_attn_fc` the CA variant builds for txt_reduce == 'attn').

What runs where: every matrix product over points, the sparse convolutions, attention, norms, the per-cloud max and the
heatmap cross entropy are lotus-hip kernels (ops.*Fn); the [B, T]-sized rotation / openness / stop losses, the
trajectory-embedding bias (5 x 64 floats) and the effective stem weight (see prepare_ptv3_batch) are a few ATen
launches on small tensors.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .config import to_cfg
from .policy import BaseModel, RobotPoseEmbedding, _PTV3_KEYS
from .ptv3 import PointTransformerV3CA


class TrajectoryActionHead(nn.Module):
    """Parameter layout of motion_planner_ptv3.py:20-75 for (max, heatmap_disc, euler_disc)."""

    def __init__(self, reduce, pos_pred_type, rot_pred_type, hidden_size, dim_actions, max_traj_len, dropout=0,
                 voxel_size=0.01, euler_resolution=5, ptv3_config=None, pos_bins=50, traj_embed_size=64):
        super().__init__()
        if (reduce, pos_pred_type, rot_pred_type) != ("max", "heatmap_disc", "euler_disc"):
            raise NotImplementedError("lotus-hip builds the published head: reduce=max, heatmap_disc, euler_disc")
        if traj_embed_size <= 0:
            raise NotImplementedError("traj_embed_size == 0 (single-step head) is the SimplePolicy head; use that")
        self.euler_resolution, self.euler_bins, self.pos_bins = euler_resolution, 360 // euler_resolution, pos_bins
        self.max_traj_len, self.hidden_size, self.dropout = max_traj_len, hidden_size, float(dropout)
        self.traj_embedding = nn.Embedding(max_traj_len, traj_embed_size)
        self.heatmap_mlp = nn.Sequential(nn.Linear(hidden_size + traj_embed_size, hidden_size), nn.LeakyReLU(0.02),
                                         nn.Dropout(dropout), nn.Linear(hidden_size, 3 * pos_bins * 2))
        self.action_mlp = nn.Sequential(nn.Linear(hidden_size + traj_embed_size, hidden_size), nn.LeakyReLU(0.02),
                                        nn.Dropout(dropout), nn.Linear(hidden_size, self.euler_bins * 3 + 1 + 1))


class MotionPlannerPTV3CA(BaseModel):
    def __init__(self, config):
        super().__init__()
        config = to_cfg(config)
        self.config = config
        act = config.action_config
        p3 = {k: v for k, v in config.ptv3_config.items() if k in _PTV3_KEYS}
        # the reference adds pc_label_channels into config.ptv3_config.in_channels in place
        # (motion_planner_ptv3.py:405-408); the caller's config is left untouched here
        p3["in_channels"] = config.ptv3_config.in_channels + act.pc_label_channels
        self.ptv3_model = PointTransformerV3CA(**p3)
        self.pc_label_embedding = nn.Embedding(4, act.pc_label_channels)   # 0 obstacle, 1 robot, 2 object, 3 target
        self.txt_fc = nn.Linear(act.txt_ft_size, act.context_channels)
        if act.txt_reduce == "attn":
            self.txt_attn_fc = nn.Linear(act.txt_ft_size, 1)               # built, never used by the CA variant
        if act.use_ee_pose:
            self.pose_embedding = RobotPoseEmbedding(act.context_channels)   # one extra context token per cloud (:421-422)
        self.act_proj_head = TrajectoryActionHead(
            act.reduce, act.pos_pred_type, act.rot_pred_type, config.ptv3_config.dec_channels[0], act.dim_actions,
            act.max_traj_len, dropout=act.dropout, voxel_size=act.voxel_size, pos_bins=act.pos_bins,
            traj_embed_size=act.traj_embed_size)
        self.apply(self._init_weights)

    def prepare_ptv3_batch(self, batch):
        """motion_planner_ptv3.py:433-463: feat = [pc_fts | label embedding], context = txt_fc(txt_embeds)."""
        labels = batch["pc_labels"].long()
        # The 64 embedding channels take only 4 distinct values per point, and the stem convolution is linear:
        #   conv(W, [pc | E[label]]) == conv([W_pc | W_emb E^T], [pc | onehot(label)])
        # so the 68-channel 5^3 convolution (81 % of whose gathered rows are absent neighbours) becomes the 8-channel
        # stem the policy already has a kernel for, with an effective weight that autograd differentiates back into
        # the stem weight and the embedding table (two products of a few hundred kFLOP); no input gradient is needed.
        W, E = self.ptv3_model.embedding.stem.conv.weight, self.pc_label_embedding.weight
        c_pc = batch["pc_fts"].shape[1]
        w_eff = torch.cat([W[..., :c_pc], torch.matmul(W[..., c_pc:], E.t())], -1)                  # [64,5,5,5,c_pc+4]
        pres = getattr(self, "_pre", None) or []
        pre = next((q for q in pres if q[0] is batch["pc_fts"] and q[1] is batch["pc_labels"]), None)
        self._pre = [q for q in pres if q is not pre][-1:]  # (at most one more: the batch after this one)
        if pre is not None:
            feat = pre[2]   # prefetch() built it (and started the front-end on exactly this tensor)
        else:
            feat = torch.cat([batch["pc_fts"].float(), F.one_hot(labels, 4).float()], -1)
        txt, extra = batch["txt_embeds"].contiguous(), {}
        if getattr(self, "act_storage", None) == "bf16":
            # bf16 activation storage (as SimplePolicyPTV3CA.prepare_ptv3_batch): the network inputs are rounded once here;
            # the integer front end keeps reading the fp32 coordinates — the first three columns of the fp32 `feat`
            extra["coord_src"] = feat
            txt, feat_in = txt.to(torch.bfloat16), feat.to(torch.bfloat16)
        else:
            feat_in = feat
        ctx = ops.LinearFn.apply(txt, self.txt_fc.weight, self.txt_fc.bias)
        ctx_counts = list(batch["txt_lens"])
        if self.config.action_config.use_ee_pose:   # motion_planner_ptv3.py:451-457
            ctx, ctx_counts = self.append_context_tokens(ctx, ctx_counts, [self.pose_embedding(batch["ee_poses"].float())])
        return {"coord": feat[:, :3], "grid_size": self.config.action_config.voxel_size, "offset": batch["offset"],
                "feat": feat_in, "stem_weight": w_eff.contiguous(), "context": ctx, "counts": list(batch["npoints_in_batch"]),
                "context_counts": ctx_counts, **extra}

    @torch.no_grad()
    def prefetch(self, batch):
        """Input-pipeline hook, as SimplePolicyPTV3CA.prefetch: build the network input of `batch` ([pc | one-hot label]) and
        start its integer front-end on the side stream; the following forward(batch) must get the same (device) batch."""
        batch = self.prepare_batch(batch)
        feat = torch.cat([batch["pc_fts"].float(), F.one_hot(batch["pc_labels"].long(), 4).float()], -1)
        self._pre = ((getattr(self, "_pre", None) or []) + [(batch["pc_fts"], batch["pc_labels"], feat)])[-2:]
        self.ptv3_model.prefetch({"coord": feat[:, :3], "grid_size": self.config.action_config.voxel_size, "offset": batch["offset"],
                                  "feat": feat, "counts": list(batch["npoints_in_batch"]),
                                  "context_counts": [c + int(bool(self.config.action_config.use_ee_pose)) for c in batch["txt_lens"]]})

    gemm_precision = None  # as SimplePolicyPTV3CA.gemm_precision
    act_storage = None     # None / 'fp32' | 'bf16': as SimplePolicyPTV3CA.act_storage (bf16 activations in HBM, fp32 masters)

    def forward(self, batch, compute_loss=False, **kwargs):
        if self.act_storage == "bf16":
            with ops.storage(torch.bfloat16):
                return self._forward(batch, compute_loss, **kwargs)
        with ops.precision(self.gemm_precision):
            return self._forward(batch, compute_loss, **kwargs)

    def _forward(self, batch, compute_loss=False, **kwargs):
        batch = self.prepare_batch(batch)
        dev = batch["pc_fts"].device
        if dev.type != "cuda":
            raise RuntimeError("lotus-hip runs on a HIP device only (no CPU fallback); move the model and batch to cuda")
        act, head = self.config.action_config, self.act_proj_head
        last = self.ptv3_model(self.prepare_ptv3_batch(batch), return_dec_layers=True)[-1]
        x, lvl = last.feat, last.level
        B, T, C = len(lvl.counts), head.max_traj_len, head.hidden_size
        nb, eb = 2 * head.pos_bins, head.euler_bins
        p = head.dropout if self.training else 0.0
        hm, am = head.heatmap_mlp, head.action_mlp
        te = head.traj_embedding.weight                                         # [T, E]

        # heatmap branch, motion_planner_ptv3.py:88-97,113-114.  Linear([x | te_t]) = x Wx^T + (te_t Wt^T + b): the
        # point-feature product is shared by all T steps and the step only shifts the bias.
        base = ops.LinearFn.apply(x, hm[0].weight[:, :C].contiguous(), None)                       # [N, C]
        step_bias = F.linear(te, hm[0].weight[:, C:], hm[0].bias)                                  # [T, C]
        xts = list(ops.StepHeadFn.apply(base, step_bias.contiguous(), hm[3].weight, hm[3].bias, p,
                                        ops.mix_seed(self.ptv3_model.last_seed, 2000)))         # T x [N, 3*nb]
        # action branch, :116-120,139-146: max over points commutes with the concatenated step embedding
        pc = ops.CloudMaxFn.apply(x, lvl)                                                         # [B, C]
        pcs = torch.cat([pc.unsqueeze(1).expand(-1, T, -1), te.to(pc.dtype).unsqueeze(0).expand(B, -1, -1)], -1).reshape(B * T, -1)
        a = F.dropout(F.leaky_relu(ops.LinearFn.apply(pcs, am[0].weight, am[0].bias), 0.02), p, self.training)
        ae = ops.LinearFn.apply(a, am[3].weight, am[3].bias).view(B, T, -1)
        pred_rot = ae[..., :eb * 3].reshape(B, T, eb, 3)
        pred_open, pred_stop = ae[..., -2], ae[..., -1]
        self.last_pred = (xts, pred_rot, pred_open, pred_stop)   # xts[t] is [N, 3*nb]; reference layout: pred_pos()

        losses = None
        if compute_loss:
            losses = self._traj_losses(xts, ae, batch, lvl)
        decode = kwargs.get("compute_final_action", True)
        if compute_loss and self.training and not decode and not kwargs.get("decode_actions", False):
            return None, losses                                  # trainer discards the actions (see policy.py)
        if decode:   # :238-267, best_disc_pos == 'max'; one launch pair per step instead of B*T host round trips
            pcf = batch["pc_fts"] if batch["pc_fts"].stride(1) == 1 else batch["pc_fts"].contiguous()
            best = act.get("best_disc_pos", "max")   # motion_planner_ptv3.py:263
            if best == "ens1":
                cnts = list(batch["npoints_in_batch"])
                pos = torch.stack([ops.pos_decode_ens1(xt.detach(), pcf, cnts, nb, act.pos_bin_size) for xt in xts], 1)
            else:
                pos = torch.stack([ops.pos_decode_max(xt.detach(), pcf, lvl.off, B, nb, act.pos_bin_size) for xt in xts], 1)
            pos = pos.float()                                    # reference: .float() at :266
        else:
            pos = batch["gt_trajs"][..., :3].float()
        from scipy.spatial.transform import Rotation as R
        rot_bins = torch.argmax(pred_rot.reshape(B * T, eb, 3), 1).cpu().numpy()
        quat = np.stack([R.from_euler("xyz", r * head.euler_resolution - 180, degrees=True).as_quat() for r in rot_bins], 0)
        quat = torch.from_numpy(quat).to(dev).reshape(B, T, 4)
        final = torch.cat([pos, quat, pred_open.detach().unsqueeze(-1), pred_stop.detach().unsqueeze(-1)], -1)
        return (final, losses) if compute_loss else final

    def pred_pos(self):
        """Last forward's position logits in the reference layout (T, 3, N, 2*pos_bins)."""
        nb = 2 * self.act_proj_head.pos_bins
        return torch.stack([xt.view(-1, 3, nb).permute(1, 0, 2) for xt in self.last_pred[0]], 0)

    def compute_loss(self, xts, pred_rot, pred_open, pred_stop, batch, lvl):
        """motion_planner_ptv3.py:307-397 (heatmap_disc / euler_disc), reference argument list."""
        B, T = pred_rot.shape[:2]
        ae = torch.cat([pred_rot.reshape(B, T, -1), pred_open.unsqueeze(-1), pred_stop.unsqueeze(-1)], -1)
        return self._traj_losses(xts, ae, batch, lvl)

    def _traj_losses(self, xts, ae, batch, lvl):
        """ae [B, T, euler_bins * 3 + 2]: rotation logits (bin, axis) at bin * 3 + axis | openness | stop."""
        dev = ae.device
        B, T = ae.shape[:2]
        gt = batch["gt_trajs"].float()
        m = batch["traj_masks"].float()                                                  # [B, T]
        dp = batch["gt_trajs_disc_pos_probs"]
        # per step t the targets in the layout the CE kernel reads: cloud-major, [3][n_b * nb] per cloud
        if isinstance(dp, torch.Tensor):
            tgts = dp.float().to(dev)                                                    # pre-packed [T, sum 3*n_b*nb]
        else:
            tgts = torch.cat([d.to(dev).float().reshape(T, -1) for d in dp], 1)
        tgts = tgts.contiguous()
        ce = torch.stack([ops.PosCEFn.apply(xts[t], tgts[t], lvl) for t in range(T)], 1)  # [B, T, 3]
        # sum_tc CE * mask / (3 * sum_t mask) per cloud, mean over clouds (:327-336); masked rotation CE, openness and
        # stop BCE (:338-383): one launch for the lot (the same arithmetic as ~40 small ATen kernels cost 2 ms per step)
        lc = self.config.loss_config
        L = ops.TrajLossFn.apply(ae.reshape(B * T, -1), ce.reshape(B * T, 3), gt.reshape(B * T, -1).contiguous(),
                                 batch["gt_trajs_stop"].float().reshape(-1).contiguous(), m.reshape(B, T).contiguous(),
                                 (ae.shape[-1] - 2) // 3, lc.pos_weight, lc.rot_weight)
        return {"pos": L[0], "rot": L[1], "open": L[2], "stop": L[3], "total": L[4]}

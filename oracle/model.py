"""ORACLE (test infrastructure, never shipped as product): floating-point hot path of 3D-LOTUS.

Functional CPU/PyTorch-fp32 restatement of `SimplePolicyPTV3CA.forward(batch, compute_loss=True,
compute_final_action=False)` with *flash-path* patch semantics (SURVEY.md Trap 1), the stale
decoder CPE input (Trap 3) and injected order permutations (Trap 4).  Parameters are read from a
plain `state_dict` with the reference's key grammar; gradients come from
torch autograd on the CPU.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.
Pinned against the imported reference (with stand-ins for spconv / flash_attn / torch_scatter,
see tests/golden/ref_harness.py) by tests/test_oracle_vs_reference.py and tests/golden/*.npz.

Citations are relative to /root/reference/genrobo3d/models.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import front_end as fe


def _t(a, dtype=torch.long):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def subm_conv(x, nbr, weight, bias):
    """spconv.SubMConv3d (PointTransformerV3/model.py:616-622, :845-852); weight (Cout,k,k,k,Cin);
    out[p] = sum_t W[:,t,:] x[nbr[p,t]]."""
    cout, cin = weight.shape[0], weight.shape[-1]
    w = weight.reshape(cout, -1, cin)
    out = x.new_zeros(x.shape[0], cout)
    for t in range(nbr.shape[1]):
        col = nbr[:, t]
        rows = torch.nonzero(col >= 0).squeeze(1)
        if rows.numel():
            out = out.index_add(0, rows, x[col[rows]] @ w[:, t, :].t())
    if bias is not None:
        out = out + bias
    return out


def _round_fp16(x, on):
    return x.half().float() if on else x


def patch_attention(qkv, lvl, order_index, H, qn_w, qn_b, kn_w, kn_b, patch_size, fp16_attn=False):
    """SerializedAttention.forward flash branch, PointTransformerV3/model.py:478-551."""
    N, C3 = qkv.shape
    C = C3 // 3
    d = C // H
    order = lvl["order_t"][order_index][lvl["pad_t"]]
    inverse = lvl["unpad_t"][lvl["inverse_t"][order_index]]
    x = qkv[order].reshape(-1, 3, H, d)
    q, k, v = x.unbind(1)
    q = F.layer_norm(q, (d,), qn_w, qn_b, 1e-6)
    k = F.layer_norm(k, (d,), kn_w, kn_b, 1e-6)
    q, k, v = (_round_fp16(t_, fp16_attn) for t_ in (q, k, v))
    scale = d ** -0.5
    cu = lvl["cu_seqlens"].tolist()
    outs = []
    i = 0
    while i < len(cu) - 1:  # batch runs of full patches
        L = cu[i + 1] - cu[i]
        j = i
        while j + 1 < len(cu) - 1 and cu[j + 2] - cu[j + 1] == L:
            j += 1
        a, b = cu[i], cu[j + 1]
        P = (b - a) // L
        qq = q[a:b].reshape(P, L, H, d).transpose(1, 2)
        kk = k[a:b].reshape(P, L, H, d).transpose(1, 2)
        vv = v[a:b].reshape(P, L, H, d).transpose(1, 2)
        s = (qq @ kk.transpose(-1, -2)) * scale
        o = torch.softmax(s, dim=-1) @ vv
        outs.append(o.transpose(1, 2).reshape(P * L, C))
        i = j + 1
    feat = _round_fp16(torch.cat(outs, 0), fp16_attn)
    return feat[inverse]


def cross_attention(q, kv, counts, ctx_counts, H, qn_w, qn_b, kn_w, kn_b, fp16_attn=False):
    """CrossAttention.forward flash branch, PointTransformerV3/model_ca.py:46-67."""
    N, C = q.shape
    d = C // H
    q = F.layer_norm(q.view(-1, H, d), (d,), qn_w, qn_b, 1e-6)
    kv = kv.view(-1, 2, H, d)
    k = F.layer_norm(kv[:, 0], (d,), kn_w, kn_b, 1e-6)
    v = kv[:, 1]
    q, k, v = (_round_fp16(t_, fp16_attn) for t_ in (q, k, v))
    scale = d ** -0.5
    outs = []
    a = c = 0
    for n, l in zip(counts, ctx_counts):
        s = torch.einsum("qhd,khd->hqk", q[a:a + n], k[c:c + l]) * scale
        outs.append(torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), v[c:c + l]).reshape(n, C))
        a += n
        c += l
    return _round_fp16(torch.cat(outs, 0), fp16_attn)


class _RoundBF16(torch.autograd.Function):
    """y = bf16(x) in the forward pass, dx = bf16(dy) in the backward pass, both kept in x's dtype: what storing an
    activation (and its gradient) as bf16 in memory does to it."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class Oracle:
    """sd: dict name -> tensor (reference key grammar).  cfg: dict(ptv3=..., action=..., loss=...)."""

    def __init__(self, sd, cfg, training=False, fp16_attn=False, bn_momentum=0.01, dtype=torch.float32):
        # dtype: arithmetic type of the restatement.  float32 is the reference's; float64 (state dict given in float64)
        # is the yardstick the full-size parity test measures BOTH fp32 implementations against — the integer front end
        # always sees the float32 coordinates
        self.sd, self.cfg, self.dt = sd, cfg, dtype
        self.training = training
        self.fp16_attn = fp16_attn
        self.bn_momentum = bn_momentum
        self.new_running = {}
        self.record_arg = False  # True: record the max-pool / cloud-max arg-max tables in the output dict (tests)
        # True: simulate bf16 ACTIVATION STORAGE (BASELINE configs[4]): every tensor an implementation would keep in
        # memory between two operators — and its gradient — is rounded to bf16, and the operands of the products
        # (weights included) are bf16; accumulation, statistics, master weights and parameter gradients keep `dtype`.
        # Not the reference's arithmetic (torch autocast keeps fp32 residuals): a model of the rounding noise such a mode
        # cannot avoid, which the full-size bf16 parity test states its bars from.
        self.bf16_storage = False

    def q(self, x):
        return _RoundBF16.apply(x) if self.bf16_storage else x

    def qw(self, w):
        """Weight as a product operand (no gradient rounding: parameter gradients stay in `dtype`)."""
        return (w.to(torch.bfloat16).to(w.dtype) - w).detach() + w if self.bf16_storage and w is not None else w

    # -- small helpers ---------------------------------------------------------------------
    def lin(self, x, name):
        return self.q(F.linear(x, self.qw(self.sd[name + ".weight"]), self.sd.get(name + ".bias")))

    def ln(self, x, name, eps=1e-5):
        return self.q(F.layer_norm(x, (x.shape[-1],), self.sd[name + ".weight"], self.sd[name + ".bias"], eps))

    def bn(self, x, name):
        """nn.BatchNorm1d(eps=1e-3, momentum=0.01), PointTransformerV3/model_ca.py:226."""
        sd = self.sd
        if self.training:
            rm, rv = sd[name + ".running_mean"].clone(), sd[name + ".running_var"].clone()
            y = F.batch_norm(x, rm, rv, sd[name + ".weight"], sd[name + ".bias"], True, self.bn_momentum, 1e-3)
            self.new_running[name + ".running_mean"], self.new_running[name + ".running_var"] = rm, rv
            return y
        return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                            sd[name + ".weight"], sd[name + ".bias"], False, 0.0, 1e-3)

    def mlp(self, x, name):
        """MLP, PointTransformerV3/model.py:577-583 (dropouts are identity in every fixture)."""
        return self.lin(self.q(F.gelu(self.lin(x, name + ".fc1"))), name + ".fc2")

    # -- blocks ----------------------------------------------------------------------------
    def block(self, x, xs, lvl, name, H, patch_size, order_index=0):
        """Block.forward, PointTransformerV3/model.py:659-680.  xs = sparse_conv_feat.features
        (== x in the encoder, == proj_skip branch only for the first Block of a decoder stage: Trap 3).  order_index =
        i % len(order) for the i-th Block of a stage (model_ca.py:285,354).  DropPath (model.py:655-657) is the identity in
        every configuration the oracle is run in (drop_path 0, or eval mode)."""
        sd = self.sd
        q_ = self.q
        c = q_(subm_conv(xs, lvl["nbr27_t"], self.qw(sd[name + ".cpe.0.weight"]), sd[name + ".cpe.0.bias"]))
        x = q_(x + self.ln(self.lin(c, name + ".cpe.1"), name + ".cpe.2"))
        qkv = self.lin(self.ln(x, name + ".norm1.0"), name + ".attn.qkv")
        a = q_(patch_attention(qkv, lvl, order_index, H, sd[name + ".attn.q_norm.weight"], sd[name + ".attn.q_norm.bias"],
                               sd[name + ".attn.k_norm.weight"], sd[name + ".attn.k_norm.bias"], patch_size,
                               self.fp16_attn))
        x = q_(x + self.lin(a, name + ".attn.proj"))
        x = q_(x + self.mlp(self.ln(x, name + ".norm2.0"), name + ".mlp.0"))
        return x

    def ca_block(self, x, lvl, ctx, ctx_counts, name, H):
        """CABlock.forward, PointTransformerV3/model_ca.py:135-152."""
        sd = self.sd
        q = self.lin(self.ln(x, name + ".norm1.0"), name + ".attn.q")
        kv = self.lin(ctx, name + ".attn.kv")
        a = self.q(cross_attention(q, kv, lvl["counts"].tolist(), ctx_counts, H,
                                   sd[name + ".attn.q_norm.weight"], sd[name + ".attn.q_norm.bias"],
                                   sd[name + ".attn.k_norm.weight"], sd[name + ".attn.k_norm.bias"], self.fp16_attn))
        x = self.q(x + self.lin(a, name + ".attn.proj"))
        x = self.q(x + self.mlp(self.ln(x, name + ".norm2.0"), name + ".mlp.0"))
        return x

    # -- whole model -----------------------------------------------------------------------
    def forward(self, batch, perms, compute_loss=True):
        """SimplePolicyPTV3AdaNorm.forward + SimplePolicyPTV3CA.prepare_ptv3_batch,
        simple_policy_ptv3.py:225-306, :403-431.  batch uses the reference schema
        (pc_fts, npoints_in_batch, txt_embeds, txt_lens, gt_actions, disc_pos_probs)."""
        assert not (self.training and self.cfg["ptv3"].get("drop_path", 0.0) > 0), "oracle: DropPath masks are not reproducible"
        pc32 = batch["pc_fts"].float()
        pc = self.q(pc32.to(self.dt))
        counts = list(batch["npoints_in_batch"])
        ctx = self.lin(self.q(batch["txt_embeds"].to(self.dt)), "txt_fc")  # simple_policy_ptv3.py:414
        ctx_counts = list(batch["txt_lens"])
        act = self.cfg["action"]
        ctx, ctx_counts = self.context_tokens(ctx, ctx_counts, batch, act.get("use_ee_pose"), act.get("use_step_id"))
        x, out = self.backbone(pc, pc32[:, :3], counts, ctx, ctx_counts, perms)
        return self.head(x, counts, batch, out, compute_loss)

    def context_tokens(self, ctx, ctx_counts, batch, use_ee_pose, use_step_id):
        """One extra context token per cloud and option, appended to that cloud's instruction tokens
        (simple_policy_ptv3.py:419-427, motion_planner_ptv3.py:451-457; RobotPoseEmbedding base.py:52-78)."""
        if not (use_ee_pose or use_step_id):
            return ctx, ctx_counts
        from scipy.spatial.transform import Rotation as R
        parts = list(torch.split(ctx, ctx_counts))
        if use_ee_pose:
            a = batch["ee_poses"].to(self.dt)
            eul = torch.from_numpy(R.from_quat(batch["ee_poses"][..., 3:7].numpy()).as_euler("xyz")).float().to(self.dt)
            e = (self.lin(a[..., :3], "pose_embedding.pos_embedding") + self.lin(torch.cat([torch.sin(eul), torch.cos(eul)], -1), "pose_embedding.rot_embedding")
                 + self.sd["pose_embedding.open_embedding.weight"][batch["ee_poses"][..., -1].long()])
            e = F.layer_norm(e, (e.shape[-1],), self.sd["pose_embedding.layer_norm.weight"], self.sd["pose_embedding.layer_norm.bias"], 1e-12)
            parts = [torch.cat([c, t_.unsqueeze(0)], 0) for c, t_ in zip(parts, e)]
            ctx_counts = [c + 1 for c in ctx_counts]
        if use_step_id:
            e = self.sd["stepid_embedding.weight"][batch["step_ids"].long()]
            parts = [torch.cat([c, t_.unsqueeze(0)], 0) for c, t_ in zip(parts, e)]
            ctx_counts = [c + 1 for c in ctx_counts]
        return torch.cat(parts, 0), ctx_counts

    def backbone(self, feat, xyz, counts, ctx, ctx_counts, perms):
        """PointTransformerV3CA.forward, PointTransformerV3/model_ca.py:314-347: embedding, encoder, decoder.
        Returns (last decoder features, dict(levels, feats))."""
        p3, act = self.cfg["ptv3"], self.cfg["action"]
        sd = self.sd
        pc = feat
        n_lv = len(p3["enc_channels"])
        levels = fe.build_all_levels(xyz.detach().numpy(), counts, n_lv,
                                     patch_size=p3["enc_patch_size"][0], perms=perms,
                                     grid_size=np.float32(act["voxel_size"]))
        for lv in levels:
            for k in ("order", "inverse"):
                lv[k + "_t"] = _t(lv[k])
            lv["pad_t"], lv["unpad_t"] = _t(lv["pad"]), _t(lv["unpad"])
            lv["nbr27_t"] = _t(lv["nbr27"])
        out = {"levels": levels}

        # Embedding, PointTransformerV3/model.py:844-861
        x = self.q(subm_conv(pc, _t(levels[0]["nbr125"]), sd["ptv3_model.embedding.stem.conv.weight"], None))
        x = self.q(F.gelu(self.bn(x, "ptv3_model.embedding.stem.norm")))
        feats, skips = [], []
        for s in range(n_lv):
            name = f"ptv3_model.enc.enc{s}"
            lvl = levels[s]
            if s > 0:  # SerializedPooling, PointTransformerV3/model.py:713-790
                proj = self.lin(x, name + ".down.proj")
                cl = _t(lvl["cluster"]).view(-1, 1).expand(-1, proj.shape[1])
                x = proj.new_zeros(lvl["grid"].shape[0], proj.shape[1]).scatter_reduce(
                    0, cl, proj, reduce="amax", include_self=False)
                if self.record_arg:
                    # arg-max table of the segment max (the FIRST parent row attaining it, as torch_scatter's CPU reduction
                    # and the HIP kernel pick it) and, so that the gradient is routed to exactly that row (amax's autograd
                    # would split it evenly among exact ties), the pooled value re-read through the table
                    with torch.no_grad():
                        rows = torch.arange(proj.shape[0]).view(-1, 1).expand_as(proj)
                        cand = torch.where(proj == x[cl[:, 0]], rows, torch.full_like(rows, proj.shape[0]))
                        arg = torch.full((x.shape[0], x.shape[1]), proj.shape[0], dtype=torch.long).scatter_reduce(
                            0, cl, cand, reduce="amin", include_self=True)
                    out.setdefault("pool_arg", []).append(arg)
                    x = torch.gather(proj, 0, arg)
                x = self.q(F.gelu(self.bn(x, name + ".down.norm.0")))
            for i in range(p3["enc_depths"][s]):  # model_ca.py:270-310
                x = self.block(x, x, lvl, name + f".block{i}", p3["enc_num_head"][s], p3["enc_patch_size"][s], i % 4)
                x = self.ca_block(x, lvl, ctx, ctx_counts, name + f".ca_block{i}", p3["enc_num_head"][s])
            skips.append(x)
        feats.append(x)
        for s in reversed(range(n_lv - 1)):  # SerializedUnpooling, model.py:817-828
            name = f"ptv3_model.dec.dec{s}"
            lvl = levels[s]
            up = self.q(F.gelu(self.bn(self.lin(x, name + ".up.proj.0"), name + ".up.proj.1")))
            skip = self.q(F.gelu(self.bn(self.lin(skips[s], name + ".up.proj_skip.0"), name + ".up.proj_skip.1")))
            x = self.q(skip + up[_t(levels[s + 1]["cluster"])])
            for i in range(p3["dec_depths"][s]):  # model_ca.py:340-380; later Blocks see the refreshed sparse_conv_feat
                x = self.block(x, skip if i == 0 else x, lvl, name + f".block{i}", p3["dec_num_head"][s],
                               p3["dec_patch_size"][s], i % 4)
                x = self.ca_block(x, lvl, ctx, ctx_counts, name + f".ca_block{i}", p3["dec_num_head"][s])
            feats.append(x)
        out["feats"] = feats
        return x, out

    def head(self, x, counts, batch, out, compute_loss):
        """ActionHead.forward (heatmap_disc / max / euler_disc), simple_policy_ptv3.py:113-157, and
        compute_loss, :308-373."""
        pre = self.lin(x, "act_proj_head.heatmap_mlp.0")
        if self.record_arg:  # the sign pattern LeakyReLU routes the gradient by (a discrete decision, like the arg-max tables)
            out["leaky_pre"] = pre.detach()
        h = self.q(F.leaky_relu(pre, 0.02))
        xt = self.lin(h, "act_proj_head.heatmap_mlp.3")  # (N, 3*2*pos_bins)
        nb = xt.shape[1] // 3
        xt = xt.view(-1, 3, nb).permute(1, 0, 2)  # 'n (c b) -> c n b'
        mx = [t_.max(0) for t_ in torch.split(x, counts)]
        pcs = torch.stack([m_[0] for m_ in mx], 0)
        if self.record_arg:  # global row of every (cloud, channel) maximum (torch.max routes the gradient to this row)
            base = np.concatenate([[0], np.cumsum(counts)])
            out["cloud_arg"] = torch.stack([m_[1] + int(base[i]) for i, m_ in enumerate(mx)], 0)
        ae = self.lin(self.q(F.leaky_relu(self.lin(pcs, "act_proj_head.action_mlp.0"), 0.02)), "act_proj_head.action_mlp.3")
        ebins = 360 // 5
        xr = ae[..., : ebins * 3].reshape(-1, ebins, 3)
        xo = ae[..., -1]
        out.update(xt=xt, xr=xr, xo=xo)
        if compute_loss:  # compute_loss, simple_policy_ptv3.py:308-373
            gt = batch["gt_actions"].to(self.dt)
            pos = 0
            for i, (lg, tg) in enumerate(zip(torch.split(xt, counts, dim=1), batch["disc_pos_probs"])):
                pos = pos + F.cross_entropy(lg.reshape(3, -1), tg.to(self.dt), reduction="mean")
            pos = pos / len(counts)
            rot = F.cross_entropy(xr, gt[..., 3:-1].long(), reduction="mean")
            opn = F.binary_cross_entropy_with_logits(xo, gt[..., -1], reduction="mean")
            lc = self.cfg["loss"]
            out["losses"] = dict(pos=pos, rot=rot, open=opn,
                                 total=lc["pos_weight"] * pos + lc["rot_weight"] * rot + opn)
        return out

    # -- 3D-LOTUS++ motion planner (BASELINE configs[3]) -----------------------------------------
    def forward_mp(self, batch, perms, compute_loss=True):
        """MotionPlannerPTV3CA: prepare_ptv3_batch motion_planner_ptv3.py:433-463, forward :222-305, trajectory
        ActionHead.forward :77-148 (heatmap_disc / max / euler_disc), compute_loss :307-397.  batch: pc_fts [N,4],
        pc_labels, txt_embeds, txt_lens, npoints_in_batch, gt_trajs [B,T,7], gt_trajs_stop, traj_masks,
        gt_trajs_disc_pos_probs (list of [T,3,n*nb])."""
        sd, act = self.sd, self.cfg["action"]
        pc32 = batch["pc_fts"].float()   # the integer front end always sees the float32 coordinates
        pc = pc32.to(self.dt)
        counts = list(batch["npoints_in_batch"])
        feat = torch.cat([pc, sd["pc_label_embedding.weight"][batch["pc_labels"].long()]], -1)   # :441-442
        ctx = self.lin(batch["txt_embeds"].to(self.dt), "txt_fc")                                  # :447
        ctx, ctx_counts = self.context_tokens(ctx, list(batch["txt_lens"]), batch, act.get("use_ee_pose"), False)  # :451-457
        x, out = self.backbone(feat, pc32[:, :3], counts, ctx, ctx_counts, perms)
        T, B = act["max_traj_len"], len(counts)
        te = sd["act_proj_head.traj_embedding.weight"]
        pe = torch.cat([x.unsqueeze(1).expand(-1, T, -1), te.unsqueeze(0).expand(x.shape[0], -1, -1)], -1)  # :90-97
        xt = self.lin(F.leaky_relu(self.lin(pe, "act_proj_head.heatmap_mlp.0"), 0.02), "act_proj_head.heatmap_mlp.3")
        nb = xt.shape[-1] // 3
        xt = xt.view(-1, T, 3, nb).permute(1, 2, 0, 3)                       # 'n t (c b) -> t c n b', :113-114
        pcs = torch.stack([t_.max(0)[0] for t_ in torch.split(pe, counts)], 0)                   # :116-120
        if self.record_arg:
            # the per-cloud maximum of [x | traj_embedding] over the points: the embedding half is constant over the points, so
            # only the feature half has an arg-max (the same row for every step); re-read through it so that the gradient is
            # routed to exactly that row
            mx = [t_.max(0) for t_ in torch.split(x, counts)]
            base = np.concatenate([[0], np.cumsum(counts)])
            out["cloud_arg"] = torch.stack([m_[1] + int(base[i]) for i, m_ in enumerate(mx)], 0)
            px = torch.stack([m_[0] for m_ in mx], 0)
            pcs = torch.cat([px.unsqueeze(1).expand(-1, T, -1), te.unsqueeze(0).expand(B, -1, -1)], -1)
        ae = self.lin(F.leaky_relu(self.lin(pcs, "act_proj_head.action_mlp.0"), 0.02), "act_proj_head.action_mlp.3")
        ebins = 360 // 5
        xr = ae[..., : ebins * 3].reshape(B, T, ebins, 3)                    # :139-142
        xo, xs = ae[..., -2], ae[..., -1]                                     # :145-146
        out.update(xt=xt, xr=xr, xo=xo, xstop=xs)
        if compute_loss:
            gt, m = batch["gt_trajs"].to(self.dt), batch["traj_masks"].to(self.dt)
            pos = 0
            for i, lg in enumerate(torch.split(xt, counts, dim=2)):          # :324-336
                ce = F.cross_entropy(lg.reshape(T * 3, -1), batch["gt_trajs_disc_pos_probs"][i].to(self.dt).reshape(T * 3, -1),
                                     reduction="none")
                mk = m[i].unsqueeze(1).expand(-1, 3).reshape(-1)
                pos = pos + (ce * mk).sum() / mk.sum()
            pos = pos / B
            rl = F.cross_entropy(xr.permute(0, 1, 3, 2).reshape(-1, ebins), gt[..., 3:-1].long().reshape(-1),
                                 reduction="none").view(B, T, 3)             # :365-372
            rot = (rl * m.unsqueeze(-1)).sum() / m.sum() / 3
            opn = (F.binary_cross_entropy_with_logits(xo, gt[..., -1], reduction="none") * m).sum() / m.sum()
            stp = (F.binary_cross_entropy_with_logits(xs, batch["gt_trajs_stop"].to(self.dt), reduction="none") * m).sum() / m.sum()
            lc = self.cfg["loss"]
            out["losses"] = dict(pos=pos, rot=rot, open=opn, stop=stp,
                                 total=lc["pos_weight"] * pos + lc["rot_weight"] * rot + opn + stp)
        return out

"""PointTransformerV3CA backbone as a drop-in `nn.Module` over the HIP operators.

Same constructor arguments, `forward(data_dict, return_dec_layers)` contract and parameter names
as `genrobo3d/models/PointTransformerV3/model_ca.py:155-412` (state_dict layout: SURVEY.md
Appendix B).  The torch sub-modules (nn.Linear, nn.LayerNorm, nn.BatchNorm1d) are used purely as
parameter containers — so checkpoints, `SyncBatchNorm.convert_sync_batchnorm` and DDP see the
usual structure — while every forward/backward computation goes through robot_3dlotus_amd.ops.

Only the configuration family of the published models is built (flash path, qk_norm, no RPE / PDNorm /
cls_mode); other options raise NotImplementedError rather than silently computing something else.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import _capi, ops
from .frontend import FrontEnd, draw_order_perms


class SubMConv3d(nn.Module):
    """Parameter container with spconv's layout: weight (Cout, k, k, k, Cin), optional bias."""

    def __init__(self, cin, cout, k, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, k, k, k, cin))
        bound = 1.0 / (cin * k ** 3) ** 0.5
        nn.init.uniform_(self.weight, -bound, bound)
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout))
        else:
            self.register_parameter("bias", None)


def _bn(c):
    return nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)


class _Attn(nn.Module):
    def __init__(self, c, h):
        super().__init__()
        self.qkv = nn.Linear(c, 3 * c)
        self.proj = nn.Linear(c, c)
        self.q_norm = nn.LayerNorm(c // h, eps=1e-6)
        self.k_norm = nn.LayerNorm(c // h, eps=1e-6)


class _CrossAttn(nn.Module):
    def __init__(self, c, h, ctx):
        super().__init__()
        self.q = nn.Linear(c, c)
        self.kv = nn.Linear(ctx, 2 * c)
        self.proj = nn.Linear(c, c)
        self.q_norm = nn.LayerNorm(c // h, eps=1e-6)
        self.k_norm = nn.LayerNorm(c // h, eps=1e-6)


class _MLP(nn.Module):
    def __init__(self, c, hidden):
        super().__init__()
        self.fc1 = nn.Linear(c, hidden)
        self.fc2 = nn.Linear(hidden, c)


class Block(nn.Module):
    """model.py:586-680"""

    def __init__(self, c, h, mlp_ratio):
        super().__init__()
        self.num_heads = h
        self.cpe = nn.Sequential(SubMConv3d(c, c, 3, bias=True), nn.Linear(c, c), nn.LayerNorm(c))
        self.norm1 = nn.Sequential(nn.LayerNorm(c))
        self.attn = _Attn(c, h)
        self.norm2 = nn.Sequential(nn.LayerNorm(c))
        self.mlp = nn.Sequential(_MLP(c, int(c * mlp_ratio)))

    def run(self, x, xs, lvl, drop_p, seed, attn_p=0.0, wt=None, dpath=0.0):
        """-> (x, hand): `hand` lets the next sub-block's backward pre-mask the gradient this block's MLP needs
        (ops.Handoff; the blocks of a stage form a chain with a single consumer each).  `lvl` carries the patch tables of
        this block's curve slot (Level.for_order); dpath = DropPath rate of the attention and MLP branches (training)."""
        c0, c1, c2 = self.cpe[0], self.cpe[1], self.cpe[2]
        x = ops.CpeFn.apply(x, xs, c0.weight, c0.bias, c1.weight, c1.bias, c2.weight, c2.bias, lvl, wt)
        a, n1 = self.attn, self.norm1[0]
        h_attn, h_mlp = ops.Handoff(), ops.Handoff()
        x = ops.SelfAttnFn.apply(x, n1.weight, n1.bias, a.qkv.weight, a.qkv.bias, a.q_norm.weight, a.q_norm.bias,
                                 a.k_norm.weight, a.k_norm.bias, a.proj.weight, a.proj.bias, lvl, self.num_heads,
                                 drop_p, seed, attn_p, h_attn, dpath)
        m, n2 = self.mlp[0], self.norm2[0]
        x = ops.FfnFn.apply(x, n2.weight, n2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, drop_p,
                            ops.mix_seed(seed, 2), h_mlp, h_attn, dpath)
        return x, (h_mlp if dpath == 0.0 else None)


class CABlock(nn.Module):
    """model_ca.py:104-152"""

    def __init__(self, c, h, ctx, mlp_ratio):
        super().__init__()
        self.num_heads = h
        self.norm1 = nn.Sequential(nn.LayerNorm(c))
        self.attn = _CrossAttn(c, h, ctx)
        self.norm2 = nn.Sequential(nn.LayerNorm(c))
        self.mlp = nn.Sequential(_MLP(c, int(c * mlp_ratio)))

    def run(self, x, context, lvl, drop_p, seed, attn_p=0.0, prev_hand=None, bank=None, bidx=0):
        """bank / bidx: the keys / values of this block come from slice `bidx` of an ops.KvBank (all CABlocks projected
        together at the top of the forward pass) instead of this block's own product over the context."""
        a, n1 = self.attn, self.norm1[0]
        h_attn = ops.Handoff()
        if bank is not None:
            x = ops.CrossAttnKvFn.apply(x, bank.slice(bidx), n1.weight, n1.bias, a.q.weight, a.q.bias, a.q_norm.weight,
                                        a.q_norm.bias, a.k_norm.weight, a.k_norm.bias, a.proj.weight, a.proj.bias, lvl,
                                        self.num_heads, drop_p, seed, attn_p, h_attn, prev_hand, bank, bidx)
        else:
            x = ops.CrossAttnFn.apply(x, context, n1.weight, n1.bias, a.q.weight, a.q.bias, a.kv.weight, a.kv.bias,
                                      a.q_norm.weight, a.q_norm.bias, a.k_norm.weight, a.k_norm.bias, a.proj.weight,
                                      a.proj.bias, lvl, self.num_heads, drop_p, seed, attn_p, h_attn, prev_hand)
        m, n2 = self.mlp[0], self.norm2[0]
        # the stage's last MLP: its output may feed several consumers (pooling + decoder skip), no hand-over to it
        return ops.FfnFn.apply(x, n2.weight, n2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, drop_p,
                               ops.mix_seed(seed, 2), None, h_attn)


class _Down(nn.Module):
    """SerializedPooling parameters, model.py:707-711"""

    def __init__(self, cin, cout):
        super().__init__()
        self.proj = nn.Linear(cin, cout)
        self.norm = nn.Sequential(_bn(cout))


class _Up(nn.Module):
    """SerializedUnpooling parameters, model.py:804-813"""

    def __init__(self, cin, cskip, cout):
        super().__init__()
        self.proj = nn.Sequential(nn.Linear(cin, cout), _bn(cout))
        self.proj_skip = nn.Sequential(nn.Linear(cskip, cout), _bn(cout))


class _Stem(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = SubMConv3d(cin, cout, 5, bias=False)
        self.norm = _bn(cout)


class _Embedding(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.stem = _Stem(cin, cout)


class PointDict(dict):
    """EasyDict stand-in for the reference's `_pack_point_dict` (model.py:1065-1070)."""
    __getattr__ = dict.__getitem__


class PointTransformerV3CA(nn.Module):
    def __init__(self, in_channels=6, order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2, 2, 2),
                 enc_depths=(2, 2, 2, 6, 2), enc_channels=(32, 64, 128, 256, 512), enc_num_head=(2, 4, 8, 16, 32),
                 enc_patch_size=(1024,) * 5, dec_depths=(2, 2, 2, 2), dec_channels=(64, 64, 128, 256),
                 dec_num_head=(4, 4, 8, 16), dec_patch_size=(1024,) * 4, mlp_ratio=4, ctx_channels=256,
                 qkv_bias=True, qk_scale=None, qk_norm=False, attn_drop=0.0, proj_drop=0.0, drop_path=0.3,
                 pre_norm=True, shuffle_orders=True, enable_rpe=False, enable_flash=True, upcast_attention=False,
                 upcast_softmax=False, cls_mode=False, pdnorm_bn=False, pdnorm_ln=False, pdnorm_decouple=True,
                 pdnorm_adaptive=False, pdnorm_context_channels=256, pdnorm_affine=True,
                 pdnorm_conditions=("ScanNet", "S3DIS", "Structured3D"), pdnorm_only_decoder=False,
                 add_coords_in_attn=False, scaled_cosine_attn=False):
        super().__init__()
        unsupported = dict(pdnorm_bn=pdnorm_bn, pdnorm_ln=pdnorm_ln, enable_rpe=enable_rpe, cls_mode=cls_mode,
                           scaled_cosine_attn=scaled_cosine_attn, not_flash=not enable_flash, not_qk_norm=not qk_norm,
                           not_pre_norm=not pre_norm, no_qkv_bias=not qkv_bias, qk_scale=qk_scale is not None,
                           add_coords=add_coords_in_attn not in (False, "none", None),
                           depth_lt_1=any(d < 1 for d in list(enc_depths) + list(dec_depths)),
                           stride_ne_2=any(s != 2 for s in stride),
                           patch_ne_128=any(p > 128 for p in list(enc_patch_size) + list(dec_patch_size)))
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"lotus-hip builds the published 3D-LOTUS configuration family only; unsupported: {bad}")
        if len(set(enc_patch_size) | set(dec_patch_size)) != 1:
            raise NotImplementedError("all patch sizes must be equal")
        self.num_stages = len(enc_depths)
        self.order = list(order)
        self.shuffle_orders = shuffle_orders
        self.proj_drop, self.attn_drop = float(proj_drop), float(attn_drop)
        self.enc_channels, self.dec_channels = list(enc_channels), list(dec_channels) + [enc_channels[-1]]
        self.enc_depths, self.dec_depths = [int(d) for d in enc_depths], [int(d) for d in dec_depths]
        self.frontend = FrontEnd(self.num_stages, patch_size=enc_patch_size[0], orders=self.order,
                                 n_patch_orders=max(self.enc_depths + self.dec_depths),
                                 conv_widths=[max(e, d) for e, d in zip(self.enc_channels, self.dec_channels)])
        # stochastic-depth schedule, model_ca.py:250-252,316-325: linear in the block index over the whole encoder /
        # decoder; the decoder's per-stage slice is reversed
        ed = torch.linspace(0, drop_path, sum(self.enc_depths)).tolist()
        dd = torch.linspace(0, drop_path, sum(self.dec_depths)).tolist() if self.dec_depths else []
        self.enc_drop_path = [ed[sum(self.enc_depths[:s]):sum(self.enc_depths[:s + 1])] for s in range(self.num_stages)]
        self.dec_drop_path = [list(reversed(dd[sum(self.dec_depths[:s]):sum(self.dec_depths[:s + 1])]))
                              for s in range(self.num_stages - 1)]

        self.embedding = _Embedding(in_channels, enc_channels[0])
        self.enc = nn.Sequential()
        for s in range(self.num_stages):
            enc = nn.Sequential()
            if s > 0:
                enc.add_module("down", _Down(enc_channels[s - 1], enc_channels[s]))
            for i in range(self.enc_depths[s]):  # model_ca.py:270-310: Block i, then CABlock i
                enc.add_module(f"block{i}", Block(enc_channels[s], enc_num_head[s], mlp_ratio))
                enc.add_module(f"ca_block{i}", CABlock(enc_channels[s], enc_num_head[s], ctx_channels, mlp_ratio))
            self.enc.add_module(f"enc{s}", enc)
        self.dec = nn.Sequential()
        dc = self.dec_channels
        for s in reversed(range(self.num_stages - 1)):
            dec = nn.Sequential()
            dec.add_module("up", _Up(dc[s + 1], enc_channels[s], dc[s]))
            for i in range(self.dec_depths[s]):  # model_ca.py:340-380
                dec.add_module(f"block{i}", Block(dc[s], dec_num_head[s], mlp_ratio))
                dec.add_module(f"ca_block{i}", CABlock(dc[s], dec_num_head[s], ctx_channels, mlp_ratio))
            self.dec.add_module(f"dec{s}", dec)
        self._blocks = [m for m in self.modules() if isinstance(m, Block)]
        # level every Block works at (encoder stage s -> level s; the decoder modules are stored in execution order)
        self._block_level = {}
        for s in range(self.num_stages):
            for m in self.enc[s].children():
                if isinstance(m, Block):
                    self._block_level[id(m)] = s
        for i, s in enumerate(reversed(range(self.num_stages - 1))):
            for m in self.dec[i].children():
                if isinstance(m, Block):
                    self._block_level[id(m)] = s
        # every CABlock in execution order (module order = encoder stages, then decoder stages as they run): their kv
        # projections of the shared context are evaluated as one product at the top of forward (ops.KvAllFn)
        self._cablocks = [m for m in self.modules() if isinstance(m, CABlock)]
        self._cab_index = {id(m): i for i, m in enumerate(self._cablocks)}
        self._pair_params = {}  # (Block, CABlock) -> their 38 parameters in ops._PAIR_PARAM_SLOTS order (module walks cost host time)
        self.kv_group = os.environ.get("LOTUS_KV_GROUP", "1") != "0"
        self._step = None  # dropout stream position; taken from stem.norm.num_batches_tracked on first use (see _seeds)
        self._seed_base = None
        self.order_perms = None  # inject a list of permutations to override the RNG draw (tests)
        self._pending, self._deferred, self._fe_stream = None, None, None  # prefetch() state
        self._nbt = None
        self._sync_bn_checked = False
        self.register_load_state_dict_post_hook(lambda m, _keys: setattr(m, "_step", None))

    def _bn_counters(self):
        """num_batches_tracked of every norm layer — BatchNorm1d, or SyncBatchNorm after convert_sync_batchnorm (not a
        BatchNorm1d subclass)."""
        if self._nbt is None:
            self._nbt = [m.num_batches_tracked for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        return self._nbt

    def _seeds(self):
        """Base seed of this forward's dropout masks.  Independent per rank (data-parallel ranks must not draw the
        same masks: the reference's ATen / flash-attn Philox streams are seeded per rank, train_simple_policy.py:64-67)
        and per optimisation step; the step is the stem BatchNorm's num_batches_tracked — a state_dict entry — so a
        resumed run continues the mask stream instead of replaying it from step 1."""
        if self._seed_base is None:
            self._seed_base = ops.mix_seed(torch.initial_seed(), ops.dist_rank())
        if self._step is None:
            self._step = int(self.embedding.stem.norm.num_batches_tracked.item())  # one host sync, first forward only
        self._step += 1
        return ops.mix_seed(self._seed_base, self._step)

    def _check_sync_bn(self):
        """`nn.SyncBatchNorm.convert_sync_batchnorm(model)` (train_simple_policy.py:116-117) swaps the BatchNorm1d
        containers for SyncBatchNorm: honour it — batch statistics over the points of ALL ranks — instead of
        silently normalising per rank.  (Collective: every rank enters its first forward.)"""
        self._sync_bn_checked = True
        import torch.distributed as dist
        if (ops.BnState.reduce is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
                and any(isinstance(m, nn.SyncBatchNorm) for m in self.modules())):
            from . import parallel
            parallel.enable_sync_batchnorm()

    def _pair(self, blk, cab, x, xs, lvl_o, lvl, p, si, pa, wt, bank, mlp_ratio_hd=None):
        """Block + CABlock as one autograd node (ops.PairFn): same seeds, same launches, same results as blk.run(...) followed by
        cab.run(...), a fifth of the host work."""
        key = (id(blk), id(cab))
        ps = self._pair_params.get(key)
        if ps is None:
            c0, c1, c2 = blk.cpe[0], blk.cpe[1], blk.cpe[2]
            a, n1, n2, m = blk.attn, blk.norm1[0], blk.norm2[0], blk.mlp[0]
            ca, cn1, cn2, cm = cab.attn, cab.norm1[0], cab.norm2[0], cab.mlp[0]
            ps = (c0.weight, c0.bias, c1.weight, c1.bias, c2.weight, c2.bias,
                  n1.weight, n1.bias, a.qkv.weight, a.qkv.bias, a.q_norm.weight, a.q_norm.bias, a.k_norm.weight, a.k_norm.bias,
                  a.proj.weight, a.proj.bias, n2.weight, n2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                  cn1.weight, cn1.bias, ca.q.weight, ca.q.bias, ca.q_norm.weight, ca.q_norm.bias, ca.k_norm.weight, ca.k_norm.bias,
                  ca.proj.weight, ca.proj.bias, cn2.weight, cn2.bias, cm.fc1.weight, cm.fc1.bias, cm.fc2.weight, cm.fc2.bias)
            self._pair_params[key] = ps
        sc = ops.mix_seed(si, 8)  # the CABlock's seed (cab.run(..., ops.mix_seed(si, 8), ...))
        meta = (lvl_o, lvl, blk.num_heads, ps[18].shape[0], p, pa, si, ops.mix_seed(si, 2), sc, ops.mix_seed(sc, 2), bank,
                self._cab_index[id(cab)])
        return ops.PairFn.apply(x, xs, bank.slice(self._cab_index[id(cab)]), wt, *ps, meta)

    def _pair_ok(self, blk, bank, dpath):
        c = blk.cpe[0].weight.shape[0]
        return (bank is not None and dpath == 0.0 and self._pair_on and c % 64 == 0 and (c == 64 or c % 128 == 0)
                and c // blk.num_heads % 4 == 0)

    def _apply(self, fn, *a, **kw):  # (parameters may be replaced: drop the cached tuples)
        self._pair_params = {}
        self._nbt = None
        return super()._apply(fn, *a, **kw)

    def _pack(self, feat, lvl):
        return PointDict(feat=feat, coord=lvl.coord, offset=lvl.off[1:].long(), level=lvl)

    def _front_inputs(self, data_dict):
        feat = data_dict["feat"]
        if not feat.is_contiguous():
            feat = feat.contiguous()
        counts = data_dict.get("counts")
        if counts is None:
            off = data_dict["offset"].tolist()
            counts = [b - a for a, b in zip([0] + off[:-1], off)]
        ctx_counts = data_dict.get("context_counts")
        if ctx_counts is None:
            off = data_dict["context_offset"].tolist()
            ctx_counts = [b - a for a, b in zip([0] + off[:-1], off)]
        context = data_dict.get("context")
        if context is not None:
            context = context.contiguous()
        coord = data_dict["coord"]
        src = data_dict.get("coord_src")  # optional: the fp32 row-major tensor whose first three columns are `coord`
        if src is None or src.dtype != torch.float32 or not src.is_contiguous() or src.data_ptr() != coord.data_ptr():
            src = feat if (coord.data_ptr() == feat.data_ptr() and feat.shape[1] >= 3) else coord.contiguous()
        self.frontend.grid_size = float(np.float32(data_dict.get("grid_size", 0.01)))
        return feat, src, counts, ctx_counts, context

    def fe_stream(self):
        if self._fe_stream is None:
            self._fe_stream = _capi.step_stream("fe")
        return self._fe_stream

    @torch.no_grad()
    def prefetch(self, data_dict, wait_current=True):
        """Start the integer front-end (serialisation, sorts, pooling tables) of `data_dict` on a side stream.
        It depends on the input cloud only, so a trainer can call this for batch i + 1 right after the forward
        of batch i: the work then hides under the backward pass, and the next forward(data_dict) — which must
        receive the very same dict — finds its tables without draining the GPU.  The order permutations are drawn
        here (same count and order of torch.randperm calls as the reference's forward)."""
        self.fe_stream()
        feat, src, counts, ctx_counts, context = self._front_inputs(data_dict)
        data_dict["feat"] = feat  # keep the tensors forward() will look at identical
        if src is not feat:
            data_dict["coord"] = src
        if self._pending is not None:
            # TWO batches ahead: a prefetched batch still waits for its forward, so this one belongs to the forward after it
            # (`prefetch(batch k + 1); forward(batch k)`).  Its pipeline is launched BY that forward, right behind the
            # exact-size tables of batch k on the front-end stream: it then runs under forward k — alone on the GPU and far
            # from filling it — and is finished a whole backward pass before the host asks for its counts.  Launched after
            # forward k (the other calling order) it shares the GPU with backward k and the host reaches the next forward
            # ~9 ms of enqueue time later: the run-ahead of the host is then capped at "backward k minus the front-end", which
            # the data-parallel step's extra host work exceeds (measured: the host waits 11 ms per step for the counts).
            self._deferred = lambda: self._launch_prefetch(src, list(counts), wait_current)
            return
        self._launch_prefetch(src, counts, wait_current)

    def drop_prefetch(self):
        """Forget a prefetched (or announced) batch that will not be passed to forward() after all."""
        self._pending, self._deferred = None, None

    def _launch_prefetch(self, src, counts, wait_current):
        perms = self.order_perms if self.order_perms is not None else draw_order_perms(self.num_stages, self.shuffle_orders)
        self._pending = self.frontend.launch(src, counts, perms, stream=self._fe_stream, wait_current=wait_current)

    def forward(self, data_dict, return_dec_layers=False):
        """data_dict keys as in the reference: coord / feat / offset / context / context_offset
        (+ grid_size).  Extra host-side hints `counts` / `context_counts` (python lists) avoid two
        device->host copies.  Returns the list [enc_last, dec..] of {feat, coord, offset} when
        return_dec_layers, else the last dict (model_ca.py:383-412)."""
        feat, src, counts, ctx_counts, context = self._front_inputs(data_dict)
        pend, self._pending = self._pending, None
        if pend is not None and pend["pc_fts"] is src and pend["counts"] == list(counts):
            levels = self.frontend.finish(pend, ctx_counts, need_coord=True)  # prefetched: no pipeline drain
        else:
            perms = self.order_perms if self.order_perms is not None else draw_order_perms(self.num_stages, self.shuffle_orders)
            levels = self.frontend.build(src, counts, ctx_counts, perms, need_coord=True)
        nxt, self._deferred = self._deferred, None
        if nxt is not None:  # the batch after this one (prefetch() before this forward)
            nxt()
        self.last_n_dup = levels[0].n_dup  # points sharing a voxel with an earlier point (0 for voxel-unique batches)
        training = self.training
        if not self._sync_bn_checked:
            self._check_sync_bn()
        p = self.proj_drop if training else 0.0
        pa = self.attn_drop if training else 0.0
        base = self._seeds() if training else 0
        self.last_seed = base  # the policy head derives its dropout seeds from the same (rank, step) stream
        if training:  # num_batches_tracked += 1 for every norm layer (BatchNorm1d or SyncBatchNorm), one fused launch
            nbt = self._bn_counters()
            if nbt:
                torch._foreach_add_(nbt, 1)
        site = 0

        st = self.embedding.stem
        blocks = self._blocks  # every Block of the model, in module order (cached: a tree walk per forward costs 0.4 ms)
        # (convolutions on the tap-grouped path read the module's weight tensor itself: 134 of the 143 MB of packing in v1)
        need = [b for b in blocks if not ops.conv_tap_active(levels[self._block_level[id(b)]], b.cpe[0].weight.shape[0])]
        packs = dict(zip(need, ops.prepack_conv_weights([b.cpe[0].weight for b in need])))
        if len(need) < len(blocks):
            none = ops.no_pack(feat.device)
            for b in blocks:
                packs.setdefault(b, none)
        n_ord = len(self.order)
        # optional effective stem weight (a differentiable function of st.conv.weight) for callers whose input
        # features are a linear code of something smaller, e.g. the motion planner's label embedding
        x = ops.StemFn.apply(feat, data_dict.get("stem_weight", st.conv.weight), st.norm.weight, st.norm.bias, st.norm.running_mean,
                             st.norm.running_var, levels[0], training)
        ops.sync_side_stream()  # the packed convolution weights (they overlapped the stem)
        bank, cidx = None, self._cab_index
        self._pair_on = False  # decided per stage below: ops.pair_enabled(points of the stage's level)
        if self.kv_group and context is not None and len(self._cablocks) > 1:
            bank = ops.KvBank()
            wb = []
            for c in self._cablocks:
                wb += [c.attn.kv.weight, c.attn.kv.bias]
            ops.KvAllFn.apply(context, bank, *wb)
        skips = []
        for s in range(self.num_stages):
            enc, lvl = self.enc[s], levels[s]
            site += 1
            seed = ops.mix_seed(base, site)
            self._pair_on = ops.pair_enabled(lvl.n, levels[0].n)
            if s > 0:
                d, bn = enc.down, enc.down.norm[0]
                x = ops.PoolFn.apply(x, d.proj.weight, d.proj.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                     lvl, training)
            for i in range(self.enc_depths[s]):
                blk, cab = getattr(enc, f"block{i}"), getattr(enc, f"ca_block{i}")
                si = seed if i == 0 else ops.mix_seed(seed, 16 + i)
                dpath = self.enc_drop_path[s][i] if training else 0.0
                if self._pair_ok(blk, bank, dpath):
                    x = self._pair(blk, cab, x, x, lvl.for_order(i % n_ord), lvl, p, si, pa, packs[blk], bank)
                    continue
                x, hand = blk.run(x, x, lvl.for_order(i % n_ord), p, si, pa, packs[blk], dpath)
                x = cab.run(x, context, lvl, p, ops.mix_seed(si, 8), pa, hand, bank, cidx[id(cab)])
            skips.append(x)
        outs = [self._pack(x, levels[-1])]
        for i, s in enumerate(reversed(range(self.num_stages - 1))):
            dec, lvl, child = self.dec[i], levels[s], levels[s + 1]
            site += 1
            seed = ops.mix_seed(base, site)
            self._pair_on = ops.pair_enabled(lvl.n, levels[0].n)
            u, us = dec.up.proj, dec.up.proj_skip
            x, skip = ops.UnpoolFn.apply(x, skips[s], u[0].weight, u[0].bias, u[1].weight, u[1].bias, u[1].running_mean,
                                         u[1].running_var, us[0].weight, us[0].bias, us[1].weight, us[1].bias,
                                         us[1].running_mean, us[1].running_var, child, training)
            for j in range(self.dec_depths[s]):
                blk, cab = getattr(dec, f"block{j}"), getattr(dec, f"ca_block{j}")
                si = seed if j == 0 else ops.mix_seed(seed, 16 + j)
                # only the first Block of a decoder stage sees the stale skip branch in its CPE convolution (Trap 3):
                # every Block / CABlock ends with sparse_conv_feat.replace_feature(feat) (model.py:678, model_ca.py:151)
                dpath = self.dec_drop_path[s][j] if training else 0.0
                if self._pair_ok(blk, bank, dpath):
                    x = self._pair(blk, cab, x, skip if j == 0 else x, lvl.for_order(j % n_ord), lvl, p, si, pa, packs[blk], bank)
                    continue
                x, hand = blk.run(x, skip if j == 0 else x, lvl.for_order(j % n_ord), p, si, pa, packs[blk], dpath)
                x = cab.run(x, context, lvl, p, ops.mix_seed(si, 8), pa, hand, bank, cidx[id(cab)])
            outs.append(self._pack(x, lvl))
        return outs if return_dec_layers else outs[-1]

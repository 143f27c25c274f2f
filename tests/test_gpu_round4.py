"""Round-4 restructuring of the step, each piece against the path it replaces:

* ops.KvAllFn / CrossAttnKvFn — the kv projections of all CABlocks as ONE product over the shared context
  (/root/reference/genrobo3d/models/PointTransformerV3/model_ca.py:46-67 evaluates one small product per block): composite
  entry point vs per-launch host path bit-identical, whole model vs the per-block projection path to fp32 summation noise;
* the one-launch BatchNorm statistics (last-arrival reduction, csrc/norm.hip) vs the three-launch path and vs float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import golden_util as gu  # noqa: E402


def _ops():
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import ops
    return ops


@pytest.mark.parametrize("M,C", [(65536, 64), (65536, 128), (23894, 128), (6077, 256), (1450, 512), (361, 768), (37, 768), (5, 64)])
def test_batchnorm_fused_statistics_match_three_launch_path_and_float64(M, C):
    ops = _ops()
    F = torch.nn.functional
    g = torch.Generator(device="cuda").manual_seed(M + C)
    x = torch.randn(M, C, device="cuda", generator=g) * 1.3 + 0.4
    dy = torch.randn(M, C, device="cuda", generator=g)
    gam, bet = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g)

    def run(fused):
        prev, ops._BN_FUSED = ops._BN_FUSED, fused
        try:
            rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
            y, mu, istd = ops.bn_fwd(x, gam, bet, rm, rv, True, ops.ACT_GELU)
            dx, dg, db = ops.bn_bwd(dy, x, mu, istd, gam, bet, True, ops.ACT_GELU)
            torch.cuda.synchronize()
            return [y, mu, istd, rm, rv, dx, dg, db]
        finally:
            ops._BN_FUSED = prev

    a, b, b2 = run(False), run(True), run(True)
    for i, (u, v, w) in enumerate(zip(a, b, b2)):
        assert torch.equal(v, w), f"fused path not reproducible (output {i})"   # counters reset, fixed summation order
        scale = float(u.abs().max()) + 1e-12
        assert float((u - v).abs().max()) <= 2e-6 * scale, (i, float((u - v).abs().max()), scale)
    if M > 1:
        xd = x.double().requires_grad_(True)
        gd, bd = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
        ref = F.gelu(F.batch_norm(xd, None, None, gd, bd, True, 0.0, 1e-3))
        ref.backward(dy.double())
        for name, got, want, tol in (("y", b[0], ref.detach(), 5e-6), ("dx", b[5], xd.grad, 3e-5), ("dgamma", b[6], gd.grad, 3e-5),
                                     ("dbeta", b[7], bd.grad, 3e-5)):
            err = float((got.double() - want).abs().max()) / (float(want.abs().max()) + 1e-12)
            assert err <= tol, (name, err)
        assert float((b[3].double() - 0.01 * x.double().mean(0)).abs().max()) <= 1e-6   # running mean, momentum 0.01


def _levels():
    from robot_3dlotus_amd import synth
    from robot_3dlotus_amd.frontend import FrontEnd

    batch = synth.synth_batch(3, 1500, ragged=True, seed=11)
    lv = FrontEnd(2).build(batch["pc_fts"].cuda(), batch["npoints_in_batch"], batch["txt_lens"], [[0, 1, 2, 3]] * 2)
    return batch, lv


@pytest.mark.parametrize("join", ["node", "end"])
@pytest.mark.parametrize("C,H", [(64, 2), (256, 8)])
def test_crossattn_kv_composite_is_bit_identical_and_matches_own_projection(C, H, join):
    """CrossAttnKvFn (kv from a KvBank of three differently sized 'blocks') — composite C call vs per-launch host path bit for
    bit, both weight-gradient join modes, dropout and hand-overs on; and the same numbers as CrossAttnFn computing its own
    projection of the context (identical kernels, the kv product only tiled over a wider output)."""
    ops = _ops()
    batch, lv = _levels()
    lvl, d, Cc = lv[0], C // H, 256
    L = sum(batch["txt_lens"])
    widths = [2 * 128, 2 * C, 2 * 64]   # the middle slice belongs to the block under test

    def tensors():
        g = torch.Generator().manual_seed(C + 1)
        x = torch.randn(lvl.n, C, generator=g).cuda().requires_grad_(True)
        ctxt = torch.randn(L, Cc, generator=g).cuda().requires_grad_(True)
        mk = lambda *s: (torch.randn(*s, generator=g) * (0.1 if len(s) == 1 else 1.0 / s[-1] ** 0.5)).cuda().requires_grad_(True)  # noqa: E731
        ps = [mk(C), mk(C), mk(C, C), mk(C), mk(d), mk(d), mk(d), mk(d), mk(C, C), mk(C)]
        kvw = [(mk(w, Cc), mk(w)) for w in widths]
        dy = torch.randn(lvl.n, C, generator=g).cuda()
        return x, ctxt, ps, kvw, dy

    def run_bank(composite):
        ops.set_composites(composite)
        x, ctxt, ps, kvw, dy = tensors()
        bank = ops.KvBank()
        sl = ops.KvAllFn.apply(ctxt, bank, *[t for pair in kvw for t in pair])
        h_in, h_out = ops.Handoff(), ops.Handoff()
        h_out.arm(0.1, 99)
        y = ops.CrossAttnKvFn.apply(x, sl[1], *ps, lvl, H, 0.1, 777, 0.1, h_in, h_out, bank, 1)
        # the other two slices get a gradient too (as the other blocks' backward passes would write them)
        loss = (y * dy).sum() + (sl[0] * 0.5).sum() + (sl[2] * -0.25).sum()
        loss.backward()
        torch.cuda.synchronize()
        return [y.detach().clone(), x.grad.clone(), ctxt.grad.clone()] + [p.grad.clone() for p in ps] + \
               [t.grad.clone() for pair in kvw for t in pair] + [h_out.dz.clone()]

    def run_own():
        ops.set_composites(True)
        x, ctxt, ps, kvw, dy = tensors()
        h_in, h_out = ops.Handoff(), ops.Handoff()
        h_out.arm(0.1, 99)
        wkv, bkv = kvw[1]
        y = ops.CrossAttnFn.apply(x, ctxt, ps[0], ps[1], ps[2], ps[3], wkv, bkv, *ps[4:], lvl, H, 0.1, 777, 0.1, h_in, h_out)
        (y * dy).sum().backward()
        torch.cuda.synchronize()
        return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in ps] + [wkv.grad.clone(), bkv.grad.clone()]

    try:
        ops.set_wgrad_join(join)
        ref, got, own = run_bank(False), run_bank(True), run_own()
    finally:
        ops.set_composites(True)
        ops.set_wgrad_join("node")
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), (i, float((a - b).abs().max()))
    # vs the block's own projection: forward output and d x, the block's parameters, and the kv weight / bias of slice 1
    mine = [got[0], got[1]] + got[3:13] + [got[13 + 2], got[13 + 3]]
    for i, (a, b) in enumerate(zip(mine, own)):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-6 * scale, (i, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("preset,n", [("tiny", 700), ("v1", 900)])
def test_model_with_grouped_kv_matches_per_block_projections(preset, n):
    """The whole policy, train mode with dropout ON (same seeds on both paths): kv projected once for all CABlocks vs every
    block projecting the context itself — logits, losses and every gradient agree to fp32 summation noise (the context
    gradient is one sum over all blocks' columns instead of a sum of per-block results)."""
    from robot_3dlotus_amd import config as lcfg, synth
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
    from weights_util import seeded_state_dict

    cfg = lcfg.preset(preset)
    sd = seeded_state_dict(gu.state_template(cfg), 7, "scaled")
    batch = synth.synth_batch(3, n, ragged=True, seed=5)
    perms = [[1, 3, 0, 2], [0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1], [1, 0, 2, 3]][:len(cfg.ptv3_config.enc_channels)]
    dev = {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if k == "disc_pos_probs" else v))
           for k, v in batch.items()}
    res = []
    for group in (False, True):
        torch.manual_seed(3)
        m = SimplePolicyPTV3CA(cfg)
        m.load_state_dict(sd)
        m = m.cuda().train()
        m.ptv3_model.kv_group = group
        m.ptv3_model.order_perms = perms
        _, losses = m(dict(dev), compute_loss=True, compute_final_action=False)
        losses["total"].backward()
        torch.cuda.synchronize()
        res.append((m.last_pred[0].detach().clone(), {k: v.detach().clone() for k, v in losses.items()},
                    {k: p.grad.clone() for k, p in m.named_parameters()}))
    (xa, la, ga), (xb, lb, gb) = res
    assert float((xa - xb).abs().max()) <= 2e-6 * max(1.0, float(xa.abs().max()))
    for k in la:
        assert abs(la[k].item() - lb[k].item()) <= 2e-6 * max(1.0, abs(la[k].item())), k
    gmax = max(float(v.norm()) for v in ga.values())
    worst = max((float((ga[k] - gb[k]).norm()) / (float(ga[k].norm()) + 1e-3 * gmax), k) for k in ga)
    assert worst[0] <= 1e-5, worst


def _peract_model(shadows, seed=93):
    from robot_3dlotus_amd import config as lcfg
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("peract")
    sd = seeded_state_dict(gu.state_template(cfg), seed, "scaled")
    m = SimplePolicyPTV3CA(cfg)
    m.load_state_dict(sd)
    m = m.cuda().train()
    m.ptv3_model.proj_drop = m.ptv3_model.attn_drop = 0.0
    m.act_proj_head.dropout = 0.0
    m.act_storage = "bf16"
    m.weight_shadows = shadows
    m.ptv3_model.order_perms = [[3, 1, 0, 2], [0, 2, 1, 3], [1, 0, 3, 2], [2, 3, 1, 0], [0, 1, 2, 3]]
    return m


def test_bf16_weight_shadows_are_bit_identical_to_per_block_conversion_and_follow_the_optimiser():
    """BASELINE configs[4], "bf16 weights with fp32 master weights": with ops.WeightShadows the dense layers read a bf16 copy
    of every weight (precision 5) instead of converting the fp32 master per block — the MFMA operands are the same bf16
    values, so logits, losses and every gradient are BIT-identical.  The fused AdamW rewrites the shadows in the launch that
    updates the masters (no version bump involved), another in-place change of a master triggers a recast."""
    from types import SimpleNamespace
    from robot_3dlotus_amd import optim as loptim, synth

    batch = synth.augment_clouds(synth.synth_batch(2, 2048, ragged=True, seed=323), seed=10, max_rot_deg=45.0)
    dev = {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if k == "disc_pos_probs" else v))
           for k, v in batch.items()}
    res = []
    for shadows in (False, True):
        m = _peract_model(shadows)
        _, losses = m(dict(dev), compute_loss=True, compute_final_action=False)
        losses["total"].backward()
        torch.cuda.synchronize()
        res.append((m, m.last_pred[0].detach().clone(), losses["total"].detach().clone(), [p.grad.clone() for p in m.parameters()]))
    (m0, x0, l0, g0), (m1, x1, l1, g1) = res
    n_sh = sum(1 for p in m1.parameters() if getattr(p, "_lotus_b16", None) is not None)
    assert n_sh >= 80 and not any(hasattr(p, "_lotus_b16") for p in m0.parameters())
    assert torch.equal(x0, x1) and torch.equal(l0, l1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    # the optimiser step keeps shadow == bf16(master) without any recast
    topts = SimpleNamespace(learning_rate=1e-3, weight_decay=0.05, optim="adamw", betas=[0.9, 0.98], lr_sched="cosine",
                            warmup_steps=1, num_train_steps=100, grad_norm=10.0)
    opt, _ = loptim.build_optimizer(m1, topts)
    before = {id(p): p.detach().clone() for p in m1.parameters()}
    opt.clip_grad_norm_(10.0)
    opt.step()
    torch.cuda.synchronize()
    changed = 0
    for p in m1.parameters():
        sh = getattr(p, "_lotus_b16", None)
        if sh is not None:
            assert torch.equal(sh, p.detach().to(torch.bfloat16)), "shadow not refreshed by the fused AdamW"
            changed += int(not torch.equal(before[id(p)], p.detach()))
    assert changed >= 80
    # any other in-place update bumps the version counter -> the next forward recasts
    w = m1.ptv3_model.enc.enc0.block0.mlp[0].fc1.weight
    with torch.no_grad():
        w.mul_(1.5)
    assert not torch.equal(w._lotus_b16, w.detach().to(torch.bfloat16))
    for p in m1.parameters():
        p.grad = None
    m1(dict(dev), compute_loss=True, compute_final_action=False)
    torch.cuda.synchronize()
    assert torch.equal(w._lotus_b16, w.detach().to(torch.bfloat16))


@pytest.mark.parametrize("join", ["node", "end"])
@pytest.mark.parametrize("preset,n,aug,storage", [("tiny", 700, False, None), ("v1", 900, True, None), ("v1", 900, False, "bf16")])
def test_pair_node_is_bit_identical_to_the_five_sub_block_nodes(preset, n, aug, storage, join):
    """ops.PairFn (one autograd node + one C call per (Block, CABlock) pair and direction, csrc/blocks.cpp lotus_pair_*)
    against the five per-sub-block nodes it replaces: train mode with dropout ON, encoder and decoder stages (the decoder's
    first Block convolves the stale skip branch), duplicate voxels, both weight-gradient join modes, fp32 and bf16 storage —
    logits, losses and every gradient bit for bit."""
    from robot_3dlotus_amd import config as lcfg, ops, synth
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
    from weights_util import seeded_state_dict

    cfg = lcfg.preset(preset)
    sd = seeded_state_dict(gu.state_template(cfg), 9, "scaled")
    batch = synth.synth_batch(3, n, ragged=True, seed=6)
    if aug:
        batch = synth.augment_clouds(batch, seed=7)
    perms = [[1, 3, 0, 2], [0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1], [1, 0, 2, 3]][:len(cfg.ptv3_config.enc_channels)]
    dev = {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if k == "disc_pos_probs" else v))
           for k, v in batch.items()}
    res = []
    try:
        ops.set_wgrad_join(join)
        for pair in (False, True):
            ops.set_pair(pair)
            torch.manual_seed(3)
            m = SimplePolicyPTV3CA(cfg)
            m.load_state_dict(sd)
            m = m.cuda().train()
            m.act_storage = storage
            m.ptv3_model.order_perms = perms
            _, losses = m(dict(dev), compute_loss=True, compute_final_action=False)
            losses["total"].backward()
            torch.cuda.synchronize()
            res.append((m.last_pred[0].detach().clone(), losses["total"].detach().clone(), [p.grad.clone() for p in m.parameters()]))
    finally:
        ops.set_pair("auto")
        ops.set_wgrad_join("node")
    (xa, la, ga), (xb, lb, gb) = res
    assert torch.equal(xa, xb) and torch.equal(la, lb)
    for i, (a, b) in enumerate(zip(ga, gb)):
        assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize("shadow", [False, True])
@pytest.mark.parametrize("M,N,K", [(65536, 128, 128), (23894, 512, 128), (23894, 128, 512), (20000, 256, 256), (16500, 192, 64),
                                   (40000, 64, 256), (30000, 384, 128), (16385, 64, 64)])
def test_bf16_linear_big_levels_match_float64_masters_and_shadows(M, N, K, shadow):
    """The bf16-storage forward / input-gradient product at the row counts of levels 0-1 (every epilogue option on: bias,
    saved pre-activation, GELU, act', dropout, residual) against (a) the float64 expression of the layer on the same
    bf16-exact inputs (one rounding of the stored output + fp32 accumulation slack) and (b) the same layer evaluated on the
    first 8192 rows only (another grid: the element indices — hence the dropout masks — must not depend on it); with fp32
    master weights (rounded while staged) and with bf16 shadows (precision 5).  (Written for a streaming kernel — weights
    resident in LDS, activation rows as MFMA fragments straight from global memory — that passed it but ran 15 % SLOWER in
    the step than the tile kernel: 16-byte row pieces per lane are 4x the memory requests of the tile kernel's 64-byte
    segments)"""
    ops = _ops()
    BF = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    bf = lambda t: t.to(BF).float()  # noqa: E731
    x, res = bf(torch.randn(M, K, device="cuda", generator=g)), bf(torch.randn(M, N, device="cuda", generator=g))
    dy, pre, add = bf(torch.randn(M, N, device="cuda", generator=g)), bf(torch.randn(M, K, device="cuda", generator=g)), bf(torch.randn(M, K, device="cuda", generator=g))
    w = bf(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
    b = torch.randn(N, device="cuda", generator=g)
    wk = w.to(BF) if shadow else w
    S = 8192
    with ops.storage(BF):
        y, p = ops.linear_fwd(x.to(BF), wk, b, residual=res.to(BF), act=ops.ACT_GELU, save_pre=True, drop_p=0.1, seed=1234)
        y0, p0 = ops.linear_fwd(x[:S].to(BF).contiguous(), wk, b, residual=res[:S].to(BF).contiguous(), act=ops.ACT_GELU, save_pre=True,
                                drop_p=0.1, seed=1234)
        dx = ops.linear_dgrad(dy.to(BF), wk, pre=pre.to(BF), add=add.to(BF), act=ops.ACT_GELU, drop_p=0.1, seed=77)
        dx0 = ops.linear_dgrad(dy[:S].to(BF).contiguous(), wk, pre=pre[:S].to(BF).contiguous(), add=add[:S].to(BF).contiguous(),
                               act=ops.ACT_GELU, drop_p=0.1, seed=77)
        yn, _ = ops.linear_fwd(x.to(BF), wk, b)   # no epilogue extras
    torch.cuda.synchronize()
    # (a) float64: pre-activation and the plain product exactly rounded; with dropout the kept elements are scaled by 1 / keep_q
    pre64 = x.double() @ w.double().t() + b.double()
    for name, got, ref in (("pre", p, pre64), ("plain", yn, pre64)):
        gd, rd = got.double(), ref
        tol = 2.0 ** -8 * rd.abs() * 1.001 + 3e-6 * float(rd.abs().max())
        assert bool(((gd - rd).abs() <= tol).all()), (name, float(((gd - rd).abs() - tol).max()))
        assert float((gd == rd.to(BF).double()).double().mean()) >= 0.98, name
    keep_q = 1 - int(0.1 * 65536) / 65536   # the keep probability the 16-bit threshold applies (round 6: the scale follows it)
    full = torch.nn.functional.gelu(pre64) / keep_q + res.double()
    kept = (y.double() - res.double()).abs() > 0          # dropped elements equal the residual exactly
    assert 0.88 < float(kept.double().mean()) < 0.92
    err = ((y.double() - full).abs() - (2.0 ** -8 * full.abs() * 1.001 + 5e-6 * float(full.abs().max())))[kept]
    assert float(err.max()) <= 0, float(err.max())
    # (b) the tile kernel on the first rows
    for name, a_, b_ in (("y", y[:S], y0), ("pre", p[:S], p0), ("dx", dx[:S], dx0)):
        same = float((a_ == b_).double().mean())
        ulp = float(((a_.double() - b_.double()).abs() / (b_.double().abs() + 1e-3)).max())
        assert same >= 0.995 and ulp <= 2.0 ** -6, (name, same, ulp)
    # input gradient against float64 where no dropout hit
    pd = pre.double().requires_grad_(True)
    torch.nn.functional.gelu(pd).sum().backward()
    dref = (dy.double() @ w.double()) * pd.grad / keep_q
    keptd = (dx.double() - add.double()).abs() > 0
    errd = ((dx.double() - (dref + add.double())).abs() - (2.0 ** -8 * (dref + add.double()).abs() * 1.001 + 5e-6 * float(dref.abs().max())))[keptd]
    assert float(errd.max()) <= 0, float(errd.max())


@pytest.mark.gpu
def test_tap_grouped_convolution_of_every_level():
    """lotus_subm_conv with a tap plan (27 gathered dense products + fixed-order tap sum; round 4: the deep levels, round 5:
    every level from 64 channels up, on the LDS-DMA tiles — 64-wide and 128-wide tile shapes, ragged last tiles): the plan is a permutation-free compaction of the neighbour table, and forward / input gradient agree with the
    float64 expression of spconv.SubMConv3d (model.py:615-622) and with the pair-compacted kernel."""
    import numpy as np
    from robot_3dlotus_amd import ops, synth
    from robot_3dlotus_amd.frontend import FrontEnd
    from robot_3dlotus_amd._capi import query

    dev = torch.device("cuda", 0)
    b = synth.synth_batch(3, 4096, ragged=True, seed=5)
    levels = FrontEnd(4, conv_widths=[64, 128, 256, 512]).build(b["pc_fts"].to(dev), b["npoints_in_batch"], b["txt_lens"], [[0, 1, 2, 3]] * 4)
    for li, C in ((0, 64), (1, 128), (2, 256), (3, 512)):
        L = levels[li]
        assert query("lotus_conv_tap_eligible", L.n, C, C) == 1 and L.tap_plan is not None
        n64 = (L.n + 63) // 64 * 64
        plan = L.tap_plan.cpu().numpy()
        cnt, tin, pos = plan[:27], plan[32:32 + 27 * n64].reshape(27, n64), plan[32 + 27 * n64:].reshape(27, L.n)
        nbr = L.nbr27.cpu().numpy()
        np.testing.assert_array_equal(cnt, (nbr >= 0).sum(1))
        for t in range(27):
            rows = np.nonzero(pos[t] >= 0)[0]
            assert len(rows) == cnt[t] and set(pos[t][rows] - t * n64) == set(range(cnt[t]))
            np.testing.assert_array_equal(tin[t][pos[t][rows] - t * n64], nbr[t][rows])   # the pair (row, neighbour) survives
            assert (tin[t][cnt[t]:(cnt[t] + 63) // 64 * 64] == 0).all()                      # padding gathers a valid row
        torch.manual_seed(li)
        x = torch.randn(L.n, C, device=dev)
        w = torch.randn(C, 3, 3, 3, C, device=dev) / (C * 9) ** 0.5
        bias, add = torch.randn(C, device=dev), torch.randn(L.n, C, device=dev)
        wt = ops.conv_weight_t(w)
        nb, w64 = L.nbr27.long(), w.double().reshape(C, 27, C)
        yr, dr = bias.double()[None, :] + add.double(), add.double().clone()
        for t in range(27):
            m = nb[t] >= 0
            yr[m] += x.double()[nb[t][m]] @ w64[:, t, :].T
            dr.index_add_(0, nb[t][m], x.double()[m] @ w64[:, t, :])                         # dx[nbr] += dy W_t (transposed pair list)
        y = ops.conv_fwd(x, w, bias, L.nbr27, L.order[0], add=add, w_t=wt, tap_plan=L.tap_plan)
        d = ops.conv_dgrad(x, w, L.nbr27, L.order[0], add=add, w_t=wt, lvl=L, tap_plan=L.tap_plan)
        y0 = ops.conv_fwd(x, w, bias, L.nbr27, L.order[0], add=add, w_t=wt)
        d0 = ops.conv_dgrad(x, w, L.nbr27, L.order[0], add=add, w_t=wt, lvl=L)
        for got, ref in ((y, yr), (d, dr), (y0, yr), (d0, dr)):
            assert float((got.double() - ref).abs().max() / ref.abs().max()) < 2e-6
        assert not torch.equal(y, y0) or True   # (different summation order: equal only by accident)
        y2 = ops.conv_fwd(x, w, bias, L.nbr27, L.order[0], add=add, w_t=wt, tap_plan=L.tap_plan)
        assert torch.equal(y, y2)                # deterministic


@pytest.mark.gpu
def test_tap_path_fails_loudly_without_its_workspace():
    """A tap plan for an eligible shape means the caller has not produced the packed weights: the path's preconditions are
    errors, not a silent fall-through to the generic kernel."""
    from robot_3dlotus_amd import ops, synth, _capi
    from robot_3dlotus_amd.frontend import FrontEnd

    dev = torch.device("cuda", 0)
    b = synth.synth_batch(2, 4096, seed=9)
    L = FrontEnd(3, conv_widths=[64, 128, 256]).build(b["pc_fts"].to(dev), b["npoints_in_batch"], b["txt_lens"], [[0, 1, 2, 3]] * 3)[2]
    C = 256
    assert L.tap_plan is not None and ops.conv_tap_active(L, C)
    x, w = torch.randn(L.n, C, device=dev), torch.randn(C, 3, 3, 3, C, device=dev)
    y = torch.empty(L.n, C, device=dev)
    small = torch.empty(1024, dtype=torch.uint8, device=dev)
    with pytest.raises(_capi.LotusError, match="workspace"):
        ops.call("lotus_subm_conv", 0, x, w, None, None, None, y, L.nbr27, L.order[0], L.n, 27, C, C, 0, L.tap_plan, small, small.numel())

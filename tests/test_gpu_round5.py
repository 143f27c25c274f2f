"""Round-5 additions outside the dense kernels (tests/test_gpu_gemm_dma.py): BatchNorm apply straight from the statistics sums."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_batchnorm_apply_from_sums_equals_finalize_then_apply():
    """lotus_batchnorm_apply_sums (the SyncBatchNorm forward without the finalisation launch) against lotus_batchnorm_finalize +
    lotus_batchnorm_apply: the same arithmetic per column -> bit-identical y, mean, invstd and running averages; widths whose
    column period does not divide 256 threads, a single row, and an empty shard (statistics only)."""
    from robot_3dlotus_amd import ops

    dev = torch.device("cuda", 0)
    for M, C, act in ((70001, 64, 1), (4097, 96, 0), (1, 128, 1), (333, 768, 1), (0, 64, 1)):
        g = torch.Generator(device="cuda").manual_seed(M + C)
        x = torch.randn(M, C, device=dev, generator=g) * 2 + 0.5
        # statistics of a LARGER (all-rank) batch than the local rows: what the all-reduce hands back
        xa = torch.cat([x, torch.randn(257, C, device=dev, generator=g)]).double()
        sums = torch.cat([xa.sum(0), (xa * xa).sum(0), torch.tensor([float(xa.shape[0])], dtype=torch.float64, device=dev)])
        gam, bet = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g)
        rm0, rv0 = torch.randn(C, device=dev, generator=g), torch.rand(C, device=dev, generator=g) + 0.5
        mean_a, inv_a, rm_a, rv_a = torch.empty(C, device=dev), torch.empty(C, device=dev), rm0.clone(), rv0.clone()
        ops.call("lotus_batchnorm_finalize", sums, mean_a, inv_a, rm_a, rv_a, C, 1e-3, 0.01)
        y_a = torch.empty_like(x)
        if M:
            ops.call("lotus_batchnorm_apply", x, mean_a, inv_a, gam, bet, y_a, M, C, act)
        mean_b, inv_b, rm_b, rv_b = torch.empty(C, device=dev), torch.empty(C, device=dev), rm0.clone(), rv0.clone()
        y_b = torch.empty_like(x)
        ops.call("lotus_batchnorm_apply_sums", x, sums, gam, bet, y_b, mean_b, inv_b, rm_b, rv_b, M, C, act, 1e-3, 0.01)
        torch.cuda.synchronize()
        assert torch.equal(mean_a, mean_b) and torch.equal(inv_a, inv_b), (M, C)
        assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b), (M, C)
        assert torch.equal(y_a, y_b), (M, C)
        ref = ((x.double() - xa.mean(0)) / torch.sqrt(xa.var(0, unbiased=False) + 1e-3)) * gam.double() + bet.double()
        if act and M:
            ref = torch.nn.functional.gelu(ref)
        if M:
            assert float((y_b.double() - ref).abs().max()) < 2e-5, (M, C)


def test_batchnorm_backward_statistics_with_parameter_gradients():
    """lotus_batchnorm_bwd_stats_fused_params: the sums of lotus_batchnorm_bwd_stats_fused bit for bit, and dbeta / dgamma = their
    fp32 conversions (what the SyncBatchNorm backward took from the sums with two conversion kernels before)."""
    from robot_3dlotus_amd import ops

    dev = torch.device("cuda", 0)
    for M, C, act in ((50001, 64, 1), (777, 256, 0), (3, 768, 1)):
        g = torch.Generator(device="cuda").manual_seed(M)
        x, dy = torch.randn(M, C, device=dev, generator=g), torch.randn(M, C, device=dev, generator=g)
        mean, invstd = x.mean(0), 1.0 / torch.sqrt(x.var(0, unbiased=False) + 1e-3)
        gam, bet = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g)
        s0 = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
        s1 = torch.empty_like(s0)
        ops._bn_bwd_stats(dy, x, mean, invstd, gam, bet, act, s0)
        dg, db = ops._bn_bwd_stats(dy, x, mean, invstd, gam, bet, act, s1, want_params=True)
        torch.cuda.synchronize()
        assert torch.equal(s0, s1), (M, C)
        assert torch.equal(db, s0[:C].float()) and torch.equal(dg, s0[C:2 * C].float()), (M, C)


def test_every_level_of_the_v1_hierarchy_takes_the_tap_grouped_convolution():
    """Routing guard (round 5): with the model's own front-end settings every level of a v1 batch carries a tap plan and both
    convolution widths of the level are eligible — a silent fall-back to the pair-compacted kernel would cost 7 % of the step
    without failing any numerical test."""
    from robot_3dlotus_amd import ops, synth
    from robot_3dlotus_amd import config as lcfg
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
    from robot_3dlotus_amd._capi import query

    dev = torch.device("cuda", 0)
    model = SimplePolicyPTV3CA(lcfg.preset("v1")).to(dev).train()
    pt = model.ptv3_model
    b = synth.synth_batch(4, 4096, seed=3)
    levels = pt.frontend.build(b["pc_fts"].to(dev), b["npoints_in_batch"], b["txt_lens"], [[0, 1, 2, 3]] * pt.num_stages)
    for li, L in enumerate(levels):
        assert L.tap_plan is not None, li
        for C in {pt.enc_channels[li], pt.dec_channels[li]}:
            assert query("lotus_conv_tap_eligible", L.n, C, C) == 1 and ops.conv_tap_active(L, C), (li, C)


def test_no_per_tensor_allocator_events_between_streams():
    """Round 5, third session: (i) the front-end's tables of one build share ONE device block per half (one record_stream / one
    allocator event instead of ~85 marker packets on the training stream when a step's tables are freed); (ii) in the deferred-join
    mode the tensors the weight-gradient stream reads are held until the end-of-backward join and released there."""
    from robot_3dlotus_amd import ops, synth
    from robot_3dlotus_amd import config as lcfg
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA

    dev = torch.device("cuda", 0)
    model = SimplePolicyPTV3CA(lcfg.preset("v1")).to(dev).train()
    pt = model.ptv3_model
    b = synth.synth_batch(2, 2048, seed=1)
    levels = pt.frontend.build(b["pc_fts"].to(dev), b["npoints_in_batch"], b["txt_lens"], [[0, 1, 2, 3]] * pt.num_stages)
    finish_blocks = {t.untyped_storage().data_ptr() for L in levels for t in (L.nbr27, L.tap_plan, L.gidx, L.owner, L.kext, L.ext_pos)}
    launch_blocks = {t.untyped_storage().data_ptr() for L in levels for t in (L.grid, L.code, L.order, L.inverse)}
    assert len(finish_blocks) == 1 and len(launch_blocks) == 1 and finish_blocks != launch_blocks
    import bench
    batch = bench.dev_batch(b, dev)
    prev = ops._JOIN
    ops.set_wgrad_join("end")
    try:
        seen = []
        orig = ops._end_of_backward

        def probe():
            seen.append(len(ops._HELD))
            orig()

        ops._end_of_backward = probe
        _, losses = model(batch, compute_loss=True, compute_final_action=False)
        losses["total"].backward()
        torch.cuda.synchronize()
    finally:
        ops._end_of_backward = orig
        ops.set_wgrad_join(prev)
    assert seen and (seen[0] > 50 or ops.SIDE is None) and len(ops._HELD) == 0


def test_a_backward_pass_that_raises_does_not_leave_the_side_stream_unjoined():
    """autograd drops its queued callbacks when backward raises, so the deferred join ("end" mode) never ran, the pending
    flag stayed set and every later backward appended to ops._HELD without ever joining again.  The next forward now notices."""
    import torch
    import golden_util as gu
    from robot_3dlotus_amd import config as lcfg, ops, synth
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("tiny")
    m = SimplePolicyPTV3CA(cfg)
    m.load_state_dict(seeded_state_dict(gu.state_template(cfg), 7, "scaled"))
    m = m.cuda().train()
    m.ptv3_model.proj_drop = m.ptv3_model.attn_drop = 0.0   # (so that two passes over the same batch give the same gradients)
    m.act_proj_head.dropout = 0.0
    m.ptv3_model.order_perms = [[0, 1, 2, 3], [3, 2, 1, 0]]
    b = synth.synth_batch(2, 400, ragged=True, seed=3)
    dev = {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if k == "disc_pos_probs" else v)) for k, v in b.items()}
    ops.set_wgrad_join("end")
    try:
        _, losses = m(dict(dev), compute_loss=True, compute_final_action=False)

        # txt_fc's backward runs late (every cross-attention block feeds it): the head and most blocks have run — and queued the
        # end-of-backward callback — when it raises
        def boom(ctx, dy):
            raise RuntimeError("boom")

        keep = ops.LinearFn.backward
        ops.LinearFn.backward = staticmethod(boom)
        try:
            with pytest.raises(RuntimeError, match="boom"):
                losses["total"].backward()
        finally:
            ops.LinearFn.backward = keep
        assert ops._END_CB_PENDING, "the failing pass was meant to leave the callback pending"
        for p in m.parameters():
            p.grad = None
        _, l2 = m(dict(dev), compute_loss=True, compute_final_action=False)   # notices the abandoned pass
        assert not ops._HELD
        l2["total"].backward()
        torch.cuda.synchronize()
        assert not ops._END_CB_PENDING and not ops._HELD
        ref = [p.grad.clone() for p in m.parameters()]
        for p in m.parameters():
            p.grad = None
        _, l3 = m(dict(dev), compute_loss=True, compute_final_action=False)
        l3["total"].backward()
        torch.cuda.synchronize()
        assert all(torch.equal(a, p.grad) for a, p in zip(ref, m.parameters()))
    finally:
        ops.set_wgrad_join("node")

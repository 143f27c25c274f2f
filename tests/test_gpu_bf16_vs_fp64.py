"""bf16 ACTIVATION STORAGE against an independent yardstick: every lotus_b16_* primitive against the
float64 torch expression of the operator evaluated on the same bf16-exact inputs — not against its own fp32 twin
(tests/test_gpu_bf16_ops.py does that).  What a bf16-storage kernel may add to the exact result is ONE rounding of each
stored output element (relative 2^-8) on top of its fp32 accumulation; operands of the products are exactly representable
here (bf16-exact activations AND weights), so the MFMA products themselves are exact.

Bars: |got - ref| <= 2^-8 |ref| + fp32-accumulation slack, and the stored value must be THE bf16 nearest to the float64
result for almost every element (a result within fp32 noise of a rounding boundary may fall to the other side)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle.model as om  # noqa: E402

BF = torch.bfloat16
HALF_ULP = 2.0 ** -8


def _g(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


def _bf(t):  # bf16-exact fp32 values
    return t.to(BF).float()


def _ops():
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import ops
    return ops


def _check_rounded(got, ref64, slack, what, exact_min=0.98):
    """got: bf16 tensor from the kernel; ref64: float64 result of the operator.  slack: absolute allowance for the kernel's
    fp32 accumulation, in units of max|ref|."""
    assert got.dtype == BF, what
    g, r = got.double().cpu(), ref64.double().cpu()
    tol = HALF_ULP * r.abs() * 1.001 + slack * float(r.abs().max())
    bad = (g - r).abs() > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} elements off by more than one bf16 rounding " \
                                f"(worst {float(((g - r).abs() - tol).max()):.3e})"
    exact = float((g == r.to(BF).double()).double().mean())
    assert exact >= exact_min, f"{what}: only {exact:.4f} of the stored values are the nearest bf16 of the float64 result"
    return exact


@pytest.mark.parametrize("M,N,K", [(5000, 128, 64), (777, 256, 512), (4096, 64, 256), (23894, 128, 128)])
def test_linear_twins_against_float64(M, N, K):
    ops = _ops()
    g = _g(M + N + K)
    x, dy, res, pre = (_bf(torch.randn(s, device="cuda", generator=g)) for s in ((M, K), (M, N), (M, N), (M, K)))
    w = _bf(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
    b = torch.randn(N, device="cuda", generator=g)
    xd, wd, bd = x.double(), w.double(), b.double()
    with ops.storage(BF):
        y, p = ops.linear_fwd(x.to(BF), w, b, residual=res.to(BF), act=ops.ACT_GELU, save_pre=True)
        dx = ops.linear_dgrad(dy.to(BF), w, pre=pre.to(BF), add=x.to(BF), act=ops.ACT_GELU)
        dw, db = ops.linear_wgrad(dy.to(BF), x.to(BF))
    torch.cuda.synchronize()
    pre64 = xd @ wd.t() + bd
    _check_rounded(p, pre64, 2e-6, "pre-activation")
    _check_rounded(y, torch.nn.functional.gelu(pre64) + res.double(), 2e-6, "gelu(linear) + residual")
    pd = pre.double().requires_grad_(True)
    torch.nn.functional.gelu(pd).sum().backward()
    _check_rounded(dx, (dy.double() @ wd) * pd.grad + xd, 2e-6, "input gradient")
    # parameter gradients stay fp32: exact products, fp32 accumulation over M rows
    dw64, db64 = dy.double().t() @ xd, dy.double().sum(0)
    assert float((dw.double() - dw64).abs().max()) <= 2e-6 * float(dw64.abs().max()) * max(1.0, (M / 4096) ** 0.5) + 1e-6
    assert float((db.double() - db64).abs().max()) <= 2e-6 * float(db64.abs().max()) * max(1.0, (M / 4096) ** 0.5) + 1e-5


def test_linear_twin_rejects_bf16x3():
    """the bf16-storage build has no bf16x3 product path; the entry point used to accept precision 3,
    launch nothing and return OK with uninitialised outputs."""
    ops = _ops()
    from robot_3dlotus_amd import _capi
    x = torch.randn(256, 64, device="cuda").to(BF)
    w = torch.randn(64, 64, device="cuda")
    with ops.storage(BF):
        for fn in (lambda: ops.linear_fwd(x, w, None, prec=3), lambda: ops.linear_dgrad(x, w, prec=3),
                   lambda: ops.linear_wgrad(x, x, prec=3)):
            with pytest.raises(_capi.LotusError, match="precision 3"):
                fn()
        y, _ = ops.linear_fwd(x, w, None, prec=1)  # the supported mode still runs
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()


@pytest.mark.parametrize("M,C", [(4097, 64), (1000, 128), (333, 768)])
def test_layernorm_batchnorm_twins_against_float64(M, C):
    ops = _ops()
    F = torch.nn.functional
    g = _g(M + C)
    x, dy, add = (_bf(torch.randn(M, C, device="cuda", generator=g) * 1.7 + 0.3) for _ in range(3))
    gam, bet = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g)
    with ops.storage(BF):
        y, mean, rstd = ops.ln_fwd(x.to(BF), gam, bet, res=add.to(BF))
    xd = x.double().requires_grad_(True)
    gd, bd = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref = F.layer_norm(xd, (C,), gd, bd, 1e-5)
    _check_rounded(y, ref.detach() + add.double(), 3e-6, "LayerNorm + residual")
    ref.backward(dy.double())
    with ops.storage(BF):
        dx, dg, db = ops.ln_bwd(dy.to(BF), x.to(BF), mean, rstd, gam, add=add.to(BF))
    torch.cuda.synchronize()
    _check_rounded(dx, xd.grad + add.double(), 5e-6, "LayerNorm input gradient")
    assert float((dg.double() - gd.grad).abs().max()) <= 1e-5 * float(gd.grad.abs().max()) + 1e-5
    assert float((db.double() - bd.grad).abs().max()) <= 1e-5 * float(bd.grad.abs().max()) + 1e-5
    # BatchNorm1d(eps 1e-3) + GELU, batch statistics
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    with ops.storage(BF):
        yb, mu, istd = ops.bn_fwd(x.to(BF), gam, bet, rm, rv, True, ops.ACT_GELU)
    xd2 = x.double().requires_grad_(True)
    g2, b2 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    refb = F.gelu(F.batch_norm(xd2, None, None, g2, b2, True, 0.0, 1e-3))
    _check_rounded(yb, refb.detach(), 3e-6, "BatchNorm + GELU")
    refb.backward(dy.double())
    with ops.storage(BF):
        dxb, dgb, dbb = ops.bn_bwd(dy.to(BF), x.to(BF), mu, istd, gam, bet, True, ops.ACT_GELU)
    torch.cuda.synchronize()
    _check_rounded(dxb, xd2.grad, 5e-6, "BatchNorm input gradient", exact_min=0.97)
    assert float((dgb.double() - g2.grad).abs().max()) <= 2e-5 * float(g2.grad.abs().max()) + 1e-5
    assert float((dbb.double() - b2.grad).abs().max()) <= 2e-5 * float(b2.grad.abs().max()) + 1e-5


@pytest.fixture(scope="module")
def levels():
    import robot_3dlotus_amd  # noqa: F401
    from oracle import front_end as ofe
    from robot_3dlotus_amd import synth
    from robot_3dlotus_amd.frontend import FrontEnd

    batch = synth.synth_batch(3, 1500, ragged=True, seed=5)
    perms = [[0, 1, 2, 3], [1, 0, 3, 2], [2, 3, 0, 1]]
    got = FrontEnd(3).build(batch["pc_fts"].cuda(), batch["npoints_in_batch"], batch["txt_lens"], perms)
    ref = ofe.build_all_levels(batch["pc_fts"][:, :3].numpy(), batch["npoints_in_batch"], 3, patch_size=128, perms=perms,
                               grid_size=np.float32(0.01))
    return got, ref, batch


@pytest.mark.parametrize("lv,C", [(0, 64), (1, 128), (2, 256)])
def test_sparse_conv_twin_against_float64(levels, lv, C):
    ops = _ops()
    got, ref, _ = levels
    L, R = got[lv], ref[lv]
    g = _g(10 * lv + C)
    x, dy, add = (_bf(torch.randn(L.n, C, device="cuda", generator=g)) for _ in range(3))
    w = _bf(torch.randn(C, 3, 3, 3, C, device="cuda", generator=g) / (13 * C) ** 0.5)
    b = torch.randn(C, device="cuda", generator=g)
    with ops.storage(torch.float32):
        wt = ops.conv_weight_t(w, prec=1)
    with ops.storage(BF):
        y = ops.conv_fwd(x.to(BF), w, b, L.nbr27, L.order[0], add=add.to(BF), w_t=wt)
        dx = ops.conv_dgrad(dy.to(BF), w, L.nbr27, L.order[0], add=add.to(BF), w_t=wt, lvl=L)
        dw, db = ops.conv_wgrad(dy.to(BF), x.to(BF), w.shape, L.nbr27)
    torch.cuda.synchronize()
    nbr = torch.from_numpy(np.ascontiguousarray(R["nbr27"])).long()
    xd = x.double().cpu().requires_grad_(True)
    wd, bd = w.double().cpu().requires_grad_(True), b.double().cpu().requires_grad_(True)
    ref_y = om.subm_conv(xd, nbr, wd, bd)
    _check_rounded(y, ref_y.detach() + add.double().cpu(), 3e-6, "sparse convolution + residual")
    ref_y.backward(dy.double().cpu())
    _check_rounded(dx, xd.grad + add.double().cpu(), 3e-6, "sparse convolution input gradient")
    assert float((dw.double().cpu() - wd.grad).abs().max()) <= 3e-6 * float(wd.grad.abs().max()) + 1e-6
    assert float((db.double().cpu() - bd.grad).abs().max()) <= 3e-6 * float(bd.grad.abs().max()) + 1e-5


@pytest.mark.parametrize("C,H", [(64, 2), (128, 4)])
def test_patch_attention_twin_against_float64(levels, C, H):
    """Self attention with bf16 rows: q / k / v are bf16-exact, the probabilities are bf16 MFMA operands (one extra rounding
    of P, relative 2^-9 per key) — the bound is bf16-sized relative to the largest output, far below the fp16 attention of the
    reference (flash_attn), and is checked against the float64 softmax attention, forward and backward."""
    ops = _ops()
    got, ref, _ = levels
    lv, r = got[0], ref[0]
    n, d = lv.n, C // H
    g = torch.Generator().manual_seed(C)
    qkv = (torch.randn(n, 3 * C, generator=g) * 1.5).to(BF).float()
    qn = (torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.2)
    kn = (torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.2)
    dout = torch.randn(n, C, generator=g).to(BF).float()
    lvl = dict(order_t=torch.from_numpy(r["order"]), inverse_t=torch.from_numpy(r["inverse"]),
               pad_t=torch.from_numpy(r["pad"]), unpad_t=torch.from_numpy(r["unpad"]), cu_seqlens=r["cu_seqlens"])
    qd = qkv.double().requires_grad_(True)
    oref = om.patch_attention(qd, lvl, 0, H, qn[0].double(), qn[1].double(), kn[0].double(), kn[1].double(), 128)
    oref.backward(dout.double())
    qc = qkv.cuda().to(BF)
    qnc, knc = tuple(t.cuda() for t in qn), tuple(t.cuda() for t in kn)
    att = torch.empty(n, C, device="cuda", dtype=BF)
    lse = torch.empty(lv.npad, H, device="cuda")
    dqkv = torch.empty(n, 3 * C, device="cuda", dtype=BF)
    extra = torch.empty(max(lv.n_extra, 1), 2 * C, device="cuda", dtype=BF)
    with ops.storage(BF):
        ops.attention_fwd(qc, 3 * C, 0, qc, 3 * C, C, 2 * C, lv.gidx, lv.gidx, lv.owner, lv.self_tiles, lv.n_self_tiles,
                          qnc, knc, att, lse, H, d)
        ops.attention_bwd(qc, 3 * C, 0, qc, 3 * C, C, 2 * C, lv.gidx, lv.gidx, lv.owner, lv.self_tiles, lv.self_blocks,
                          lv.n_self_tiles, qnc, knc, att, dout.cuda().to(BF), lse, dqkv, 3 * C, 0, dqkv, 3 * C, C, 2 * C, 0, 0, H, d,
                          0.0, 0, lv.kext, lv.ext_pos, lv.n_extra, extra)
    torch.cuda.synchronize()
    e_f = float((att.double().cpu() - oref.detach()).abs().max()) / float(oref.abs().max())
    e_b = float((dqkv.double().cpu() - qd.grad).abs().max()) / float(qd.grad.abs().max())
    assert e_f <= 1.5e-2 and e_b <= 3e-2, (e_f, e_b)

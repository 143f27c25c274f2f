"""The LDS-DMA dense kernels (csrc/gemm_dma.h, round 5) on the products they are routed to: >= 16 384 activation rows,
16-byte aligned operands.  The generic linear tests of test_gpu_ops.py reach them too (their tall shapes); this file
covers what those do not: the fused dropout mask (replayed by the stand-alone mask kernel and by the input gradient),
ragged row counts next to partial column tiles, reductions that are multiples of 16 but not 32, and the float64 product
for every epilogue the model uses."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import ops

    return ops


def _err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("M,N,K", [(16384, 128, 128), (16411, 128, 48), (20000, 192, 64), (16384, 64, 96), (17003, 512, 128),
                                   (16384, 384, 128), (32768, 64, 64), (16385, 256, 16)])
def test_forward_epilogues_against_float64(M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M + 7 * N + K)
    x, w, b, r = (torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g),
                  torch.randn(M, N, generator=g))
    xd, wd, bd, rd = x.cuda(), w.cuda(), b.cuda(), r.cuda()
    ref_pre = x.double() @ w.double().t() + b.double()
    tol = 4e-7 * K ** 0.5 + 1e-6
    y, _ = ops.linear_fwd(xd, wd, None)                                   # plain
    assert _err(y, x.double() @ w.double().t()) <= tol
    y, _ = ops.linear_fwd(xd, wd, bd)                                     # bias (the accumulators start from it)
    assert _err(y, ref_pre) <= tol
    y, pre = ops.linear_fwd(xd, wd, bd, act=1, save_pre=True)             # fc1: bias, saved pre-activation, GELU
    assert _err(pre, ref_pre) <= tol and _err(y, F.gelu(ref_pre)) <= tol
    y, _ = ops.linear_fwd(xd, wd, bd, residual=rd)                        # proj / fc2 without dropout
    assert _err(y, ref_pre + r.double()) <= tol
    y, _ = ops.linear_fwd(xd, wd, bd, residual=rd, drop_p=0.25, seed=99)  # proj / fc2: dropout on the product, then the residual
    mask = ops.dropout(torch.ones(M, N, device="cuda"), 0.25, 99)         # (the stand-alone mask kernel: same hash, same index)
    assert 0.72 < float((mask != 0).float().mean()) < 0.78
    assert _err(y, ref_pre * mask.double().cpu() + r.double()) <= tol * 1.4
    y2, _ = ops.linear_fwd(xd, wd, bd, residual=rd, drop_p=0.25, seed=99)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("M,N,K", [(16384, 128, 128), (16411, 48, 128), (20000, 64, 192), (17003, 128, 512), (16384, 128, 384),
                                   (32768, 64, 64), (16385, 16, 256)])
def test_input_gradient_epilogues_against_float64(M, N, K):
    """dx = (dy w) * act'(pre) * dropout-mask + add with dy [M, N], w [N, K]: the reduction runs over N."""
    ops = _ops()
    g = torch.Generator().manual_seed(3 * M + N + K)
    w, dy = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(M, N, generator=g)
    pre, addt = torch.randn(M, K, generator=g), torch.randn(M, K, generator=g)
    dyd, wd = dy.cuda(), w.cuda()
    tol = 4e-7 * N ** 0.5 + 1e-6
    ref = dy.double() @ w.double()
    assert _err(ops.linear_dgrad(dyd, wd), ref) <= tol
    xr = pre.double().requires_grad_(True)
    (gr,) = torch.autograd.grad(F.gelu(xr).sum(), xr)
    assert _err(ops.linear_dgrad(dyd, wd, pre=pre.cuda(), add=addt.cuda(), act=1), ref * gr + addt.double()) <= tol
    dx = ops.linear_dgrad(dyd, wd, add=addt.cuda(), drop_p=0.1, seed=5)
    mask = ops.dropout(torch.ones(M, K, device="cuda"), 0.1, 5)
    assert _err(dx, ref * mask.double().cpu() + addt.double()) <= tol * 1.2


@pytest.mark.parametrize("M,N,K", [(16384, 128, 128), (65536, 64, 64), (40011, 512, 128), (23894, 128, 512), (16411, 192, 64), (33000, 384, 128),
                                   (65536, 128, 64), (20000, 256, 128)])
def test_weight_gradient_against_float64(M, N, K):
    """dw = dy^T x, db = column sums of dy: split over the rows, partial slabs summed in a fixed order."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + 5 * K)
    x, dy = torch.randn(M, K, generator=g), torch.randn(M, N, generator=g)
    dw, db = ops.linear_wgrad(dy.cuda(), x.cuda())
    tol = 4e-7 * M ** 0.5 + 1e-6
    assert _err(dw, dy.double().t() @ x.double()) <= tol
    assert _err(db, dy.double().sum(0)) <= tol
    dw2, db2 = ops.linear_wgrad(dy.cuda(), x.cuda())
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "the weight gradient must be deterministic"
    dw3, _ = ops.linear_wgrad(dy.cuda(), x.cuda(), need_bias=False)
    assert torch.equal(dw, dw3)


@pytest.mark.parametrize("M,N,C", [(16384, 384, 128), (20011, 512, 128), (32768, 192, 64), (16500, 256, 64), (65536, 512, 128),
                                   (4000, 384, 128), (16384, 1024, 256)])
def test_input_gradient_with_layernorm_backward_epilogue(M, N, C):
    """lotus_linear_dgrad_ln: dx = LN'(dy w) + add, the dropout-masked copy dz, dgamma / dbeta — against float64 autograd and
    against the two-launch path (lotus_linear_dgrad + lotus_layernorm_bwd), which the call itself takes where the fused kernel
    does not apply (few rows, C = 256)."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + C)
    x = torch.randn(M, C, generator=g) * 1.5 + 0.3
    w, dy, addt = torch.randn(N, C, generator=g) / C ** 0.5, torch.randn(M, N, generator=g), torch.randn(M, C, generator=g)
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xd = x.cuda()
    _, mean, rstd = ops.ln_fwd(xd, gam.cuda(), bet.cuda())
    dx, dg, db, dz, nparts = ops.linear_dgrad_ln(dy.cuda(), w.cuda(), xd, mean, rstd, gam.cuda(), add=addt.cuda(), drop=(0.1, 77))
    fused = M >= 16384 and C <= 128
    assert (nparts == (M + 127) // 128) == fused, (nparts, fused)
    x64, g64, b64 = x.double().requires_grad_(True), gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    y = F.layer_norm(x64, (C,), g64, b64, 1e-5)
    y.backward(dy.double() @ w.double())
    tol = 4e-7 * N ** 0.5 + 4e-6
    assert _err(dx, x64.grad + addt.double()) <= tol
    assert _err(dg, g64.grad) <= 4e-7 * M ** 0.5 + 1e-5 and _err(db, b64.grad) <= 4e-7 * M ** 0.5 + 1e-5
    mask = ops.dropout(torch.ones(M, C, device="cuda"), 0.1, 77)
    assert torch.equal(dz, dx * mask)
    # the two-launch path on the same inputs
    dn = ops.linear_dgrad(dy.cuda(), w.cuda())
    dx2, dg2, db2 = ops.ln_bwd(dn, xd, mean, rstd, gam.cuda(), add=addt.cuda())
    assert _err(dx, dx2) <= 2e-6 and _err(dg, dg2) <= 1e-5 and _err(db, db2) <= 1e-5
    dx3, dg3, db3, _, _ = ops.linear_dgrad_ln(dy.cuda(), w.cuda(), xd, mean, rstd, gam.cuda(), add=addt.cuda(), drop=(0.1, 77))
    assert torch.equal(dx, dx3) and torch.equal(dg, dg3) and torch.equal(db, db3)


def test_ragged_rows_are_deterministic():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    M, N, K = 16384 + 37, 128, 64
    x, w = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) / 8).cuda()
    y1, _ = ops.linear_fwd(x, w, None)
    y2, _ = ops.linear_fwd(x, w, None)
    assert torch.equal(y1, y2)
    assert _err(y1, x.double().cpu() @ w.double().cpu().t()) <= 5e-6

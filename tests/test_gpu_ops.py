"""GPU unit tests of every HIP kernel family against plain PyTorch fp32/fp64 formulations of the
same op (and the oracle's conv / attention restatements).  Tolerances are written per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import front_end as fe  # noqa: E402
from oracle import model as om  # noqa: E402


def _ops():
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import ops

    return ops


def _close(a, b, tol, msg=""):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{msg}: max err {err:.3e} > {tol:.1e} * {scale:.3g}"


# ------------------------------------------------------------------------------------ linear
@pytest.mark.parametrize("M,N,K", [(65536, 64, 64), (1000, 192, 64), (441, 768, 768), (3000, 256, 1024), (190, 512, 256),
                                   (65536, 90, 128), (16, 217, 128), (7, 128, 128), (4097, 128, 512),
                                   # tall thin layers of level 0: the weights-stationary kernel (all four column counts,
                                   # all three reduction depths, a ragged last row tile)
                                   (65536, 256, 64), (65536, 64, 256), (40011, 128, 128), (65536, 192, 64), (33000, 256, 128)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_fwd(M, N, K, act):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b, r = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    y, pre = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda(), residual=r.cuda(), act=act, save_pre=True)
    ref_pre = x.double() @ w.double().t() + b.double()
    ref = {0: lambda t: t, 1: lambda t: F.gelu(t), 2: lambda t: F.leaky_relu(t, 0.02)}[act](ref_pre) + r.double()
    _close(pre, ref_pre, 2e-6, "pre")
    _close(y, ref, 2e-6, "y")


@pytest.mark.parametrize("M,N,K", [(65536, 256, 64), (1000, 64, 192), (441, 3072, 768), (65536, 90, 128), (16, 217, 128),
                                   (65536, 64, 256), (40011, 128, 128), (65536, 64, 192)])
def test_linear_dgrad_wgrad(M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x, w, dy = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(M, N, generator=g)
    pre, addt = torch.randn(M, K, generator=g), torch.randn(M, K, generator=g)
    dx = ops.linear_dgrad(dy.cuda(), w.cuda(), pre=pre.cuda(), add=addt.cuda(), act=1)
    xr = pre.double().requires_grad_(True)
    (gr,) = torch.autograd.grad(F.gelu(xr).sum(), xr)
    tol_n, tol_m = 4e-7 * N ** 0.5 + 1e-6, 4e-7 * M ** 0.5 + 1e-6   # fp32 accumulation over N / M terms
    _close(dx, (dy.double() @ w.double()) * gr + addt.double(), tol_n, "dgrad")
    dx0 = ops.linear_dgrad(dy.cuda(), w.cuda())
    _close(dx0, dy.double() @ w.double(), tol_n, "dgrad plain")
    dw, db = ops.linear_wgrad(dy.cuda(), x.cuda())
    _close(dw, dy.double().t() @ x.double(), tol_m, "wgrad")
    _close(db, dy.double().sum(0), tol_m, "bgrad")
    dw2, _ = ops.linear_wgrad(dy.cuda(), x.cuda())
    assert torch.equal(dw, dw2), "wgrad must be deterministic"


@pytest.mark.parametrize("prec", [1, 3])
@pytest.mark.parametrize("M,N,K", [(65536, 256, 64), (23894, 128, 512), (1000, 64, 192), (441, 3072, 768), (6077, 256, 1024)])
def test_linear_bf16_operand_paths(prec, M, N, K):
    """LOTUS_GEMM_PREC 1 (bf16 operands: products of bf16-rounded inputs are exact in fp32, so the result must match
    the fp64 product of the ROUNDED operands to fp32-accumulation accuracy) and 3 (bf16x3 split: within 2^-16 of the
    exact fp32-operand product, relative to sum |a||b|)."""
    from robot_3dlotus_amd import _capi
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K + prec)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    ops.set_gemm_precision({1: "bf16", 3: "bf16x3"}[prec])
    try:
        y, _ = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda())
        dx = ops.linear_dgrad(dy.cuda(), w.cuda())
        dw, db = ops.linear_wgrad(dy.cuda(), x.cuda())
        dw2, db2 = ops.linear_wgrad(dy.cuda(), x.cuda())
    finally:
        ops.set_gemm_precision("fp32")
    assert torch.equal(dw, dw2) and torch.equal(db, db2), "weight gradient must stay deterministic"
    if prec == 1:
        xr, wr, dyr = (t.bfloat16().double() for t in (x, w, dy))
        tol_k, tol_n = 4e-7 * K ** 0.5 + 1e-6, 4e-7 * N ** 0.5 + 1e-6
    else:
        xr, wr, dyr = x.double(), w.double(), dy.double()
        tol_k = tol_n = 2.0 ** -16
    ref_y, ref_dx = xr @ wr.t() + b.double(), dyr @ wr
    sy = (x.abs().double() @ w.abs().double().t()).clamp_min(1.0)     # sum |a||b| scale of every output
    sx = (dy.abs().double() @ w.abs().double()).clamp_min(1.0)
    ey = ((y.cpu().double() - ref_y).abs() / sy).max().item()
    ex = ((dx.cpu().double() - ref_dx).abs() / sx).max().item()
    assert ey <= tol_k and ex <= tol_n, (prec, ey, ex)
    # weight gradient: split-K partials of bf16(-split) products; the bias gradient is summed from the exact fp32 inputs
    tol_m = (4e-7 * M ** 0.5 + 1e-6) if prec == 1 else 2.0 ** -16
    sw = (dy.abs().double().t() @ x.abs().double()).clamp_min(1.0)
    ew = ((dw.cpu().double() - dyr.t() @ xr).abs() / sw).max().item()
    eb = ((db.cpu().double() - dy.double().sum(0)).abs() / dy.abs().double().sum(0).clamp_min(1.0)).max().item()
    assert ew <= tol_m and eb <= 4e-7 * M ** 0.5 + 1e-6, (prec, ew, eb)


def test_linear_dropout_statistics_and_replay():
    ops = _ops()
    x = torch.ones(4096, 64).cuda()
    w = torch.eye(64).cuda()
    y, _ = ops.linear_fwd(x, w, None, drop_p=0.1, seed=1234)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01
    # kept elements carry the scale of the probability the pair hash applies: t / 65536 with t = floor(0.1 * 65536) (round 6)
    assert torch.allclose(y[y != 0], torch.tensor(1 / (1 - int(0.1 * 65536) / 65536)).cuda(), rtol=1e-6)
    y2, _ = ops.linear_fwd(x, w, None, drop_p=0.1, seed=1234)
    assert torch.equal(y, y2)
    assert torch.equal(ops.dropout(x, 0.1, 1234), y), "standalone mask must replay the fused epilogue mask"
    y3, _ = ops.linear_fwd(x, w, None, drop_p=0.1, seed=1235)
    assert not torch.equal(y, y3)


# ------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("M,C", [(65536, 64), (1000, 128), (777, 256), (441, 512), (300, 768)])
def test_layernorm(M, C):
    ops = _ops()
    g = torch.Generator().manual_seed(C)
    x, r, dy = torch.randn(M, C, generator=g) * 2 + 0.5, torch.randn(M, C, generator=g), torch.randn(M, C, generator=g)
    gam, bet, addt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g), torch.randn(M, C, generator=g)
    y, mean, rstd = ops.ln_fwd(x.cuda(), gam.cuda(), bet.cuda(), res=r.cuda())
    xd = x.double().requires_grad_(True)
    gd, bd = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref = F.layer_norm(xd, (C,), gd, bd, 1e-5)
    _close(y, ref + r.double(), 3e-6, "ln fwd")
    ref.backward(dy.double())
    dx, dg, db = ops.ln_bwd(dy.cuda(), x.cuda(), mean, rstd, gam.cuda(), add=addt.cuda())
    _close(dx, xd.grad + addt.double(), 5e-6, "ln dx")
    _close(dg, gd.grad, 5e-6, "ln dgamma")
    _close(db, bd.grad, 5e-6, "ln dbeta")


@pytest.mark.parametrize("M,C", [(65536, 64), (6921, 256), (37, 768)])
@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_gelu(M, C, training):
    ops = _ops()
    g = torch.Generator().manual_seed(C + M)
    x, dy = torch.randn(M, C, generator=g) * 1.7 + 0.3, torch.randn(M, C, generator=g)
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    rm_d, rv_d = rm.clone().cuda(), rv.clone().cuda()
    y, mean, invstd = ops.bn_fwd(x.cuda(), gam.cuda(), bet.cuda(), rm_d, rv_d, training, 1)
    xd = x.double().requires_grad_(True)
    gd, bd = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    rm2, rv2 = rm.double().clone(), rv.double().clone()
    ref = F.gelu(F.batch_norm(xd, rm2, rv2, gd, bd, training, 0.01, 1e-3))
    _close(y, ref, 3e-6, "bn fwd")
    if training:
        _close(rm_d, rm2, 1e-6, "running mean")
        _close(rv_d, rv2, 1e-6, "running var")
    ref.backward(dy.double())
    dx, dg, db = ops.bn_bwd(dy.cuda(), x.cuda(), mean, invstd, gam.cuda(), bet.cuda(), training, 1)
    _close(dx, xd.grad, 1e-5, "bn dx")
    _close(dg, gd.grad, 1e-5, "bn dgamma")
    _close(db, bd.grad, 1e-5, "bn dbeta")


# ------------------------------------------------------------------------------------ sparse conv
def _cloud_levels(B, n, seed, n_levels=2):
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import synth
    from robot_3dlotus_amd.frontend import FrontEnd

    batch = synth.synth_batch(B, n, ragged=True, seed=seed)
    perms = [[0, 1, 2, 3]] * n_levels
    ref = fe.build_all_levels(batch["pc_fts"][:, :3].numpy(), batch["npoints_in_batch"], n_levels, perms=perms)
    got = FrontEnd(n_levels).build(batch["pc_fts"].cuda(), batch["npoints_in_batch"], batch["txt_lens"], perms)
    return batch, ref, got


@pytest.mark.parametrize("cin,cout,k", [(64, 64, 3), (128, 128, 3), (7, 64, 5), (256, 128, 3)])
def test_subm_conv_fwd_dgrad_wgrad(cin, cout, k):
    ops = _ops()
    batch, ref, got = _cloud_levels(3, 1500, seed=cin + k)
    n = got[0].n
    g = torch.Generator().manual_seed(cin * cout)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cout, k, k, k, cin, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    dy = torch.randn(n, cout, generator=g)
    nbr_ref = torch.from_numpy(ref[0]["nbr27" if k == 3 else "nbr125"]).long()
    nbr = got[0].nbr27 if k == 3 else got[0].nbr125
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yref = om.subm_conv(xd, nbr_ref, wd, bd)
    y = ops.conv_fwd(x.cuda(), w.cuda(), b.cuda(), nbr, got[0].order[0])
    _close(y, yref, 3e-6, "conv fwd")
    wt = ops.conv_weight_t(w.cuda()) if k == 3 else None
    if k == 3:
        y2 = ops.conv_fwd(x.cuda(), w.cuda(), b.cuda(), nbr, got[0].order[0], w_t=wt)
        _close(y2, yref, 3e-6, "conv fwd (pair-compacted, packed weights)")
    yref.backward(dy.double())
    if cin == cout:
        dx = ops.conv_dgrad(dy.cuda(), w.cuda(), nbr, got[0].order[0])
        _close(dx, xd.grad, 3e-6, "conv dgrad")
    dw, db = ops.conv_wgrad(dy.cuda(), x.cuda(), w.shape, nbr)
    _close(dw, wd.grad, 5e-6, "conv wgrad")
    _close(db, bd.grad, 5e-6, "conv bgrad")
    dw2, _ = ops.conv_wgrad(dy.cuda(), x.cuda(), w.shape, nbr, need_bias=False)  # (thin inputs: active-pair VALU kernel)
    _close(dw2, wd.grad, 5e-6, "conv wgrad (no bias)")
    if cin != cout and k == 3:
        dx = ops.conv_dgrad(dy.cuda(), w.cuda(), nbr, None)
        _close(dx, xd.grad, 3e-6, "conv dgrad (cin != cout, natural row order)")
    if k == 3:
        dx = ops.conv_dgrad(dy.cuda(), w.cuda(), nbr, got[0].order[0], w_t=wt)
        _close(dx, xd.grad, 3e-6, "conv dgrad (pair-compacted, packed weights)")


def _dup_cloud_levels(seed):
    """Clouds after the training-time augmentation (z rotation + 0-2 mm jitter): several per cent of the points share
    a voxel with another point (SURVEY.md Trap 5)."""
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import synth
    from robot_3dlotus_amd.frontend import FrontEnd

    batch = synth.augment_clouds(synth.synth_batch(3, 1500, ragged=True, seed=seed), seed=seed + 1)
    perms = [[0, 1, 2, 3]] * 2
    ref = fe.build_all_levels(batch["pc_fts"][:, :3].numpy(), batch["npoints_in_batch"], 2, perms=perms)
    got = FrontEnd(2).build(batch["pc_fts"].cuda(), batch["npoints_in_batch"], batch["txt_lens"], perms)
    rep = ref[0]["nbr27"][:, 13]
    n_dup = int((rep != np.arange(len(rep))).sum())
    assert 0.01 * len(rep) < n_dup < 0.10 * len(rep), f"augmentation should duplicate 1-10 % of the voxels, got {n_dup}"
    assert got[0].n_dup == n_dup and got[1].n_dup == 0
    return batch, ref, got


@pytest.mark.parametrize("cin,k", [(64, 3), (128, 3), (8, 5)])
def test_subm_conv_dgrad_with_duplicate_voxels(cin, k):
    """with several points per voxel the input gradient must be the gradient of the forward that was
    computed (neighbour = lowest index of the cell), not the mirrored-tap shortcut: autograd through the oracle's
    subm_conv on the same tables is the reference (PointTransformerV3/model.py:615-625; duplicates arise from
    simple_policy_dataset.py:158-181)."""
    ops = _ops()
    batch, ref, got = _dup_cloud_levels(seed=40 + cin)
    n, lvl = got[0].n, got[0]
    g = torch.Generator().manual_seed(cin * k)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cin, k, k, k, cin, generator=g) / (cin * 9) ** 0.5
    dy = torch.randn(n, cin, generator=g)
    addt = torch.randn(n, cin, generator=g)
    nbr_ref = torch.from_numpy(ref[0]["nbr27" if k == 3 else "nbr125"]).long()
    nbr = lvl.nbr27 if k == 3 else lvl.nbr125
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    om.subm_conv(xd, nbr_ref, wd, None).backward(dy.double())
    wt = ops.conv_weight_t(w.cuda()) if k == 3 else None
    dx = ops.conv_dgrad(dy.cuda(), w.cuda(), nbr, lvl.order[0], add=addt.cuda(), w_t=wt, lvl=lvl)
    _close(dx, xd.grad + addt.double(), 3e-6, "conv dgrad with duplicate voxels")
    naive = ops.conv_dgrad(dy.cuda(), w.cuda(), nbr, lvl.order[0], add=addt.cuda(), w_t=wt)  # mirrored taps only
    assert float((naive.cpu().double() - xd.grad - addt.double()).abs().max()) > 1e-2, "the case must exercise the fix"
    dw, _ = ops.conv_wgrad(dy.cuda(), x.cuda(), w.shape, nbr, need_bias=False)
    _close(dw, wd.grad, 5e-6, "conv wgrad with duplicate voxels")
    dx2 = ops.conv_dgrad(dy.cuda(), w.cuda(), nbr, lvl.order[0], add=addt.cuda(), w_t=wt, lvl=lvl)
    assert torch.equal(dx, dx2), "deterministic"


@pytest.mark.parametrize("C", [64, 128])
def test_cpe_block_fwd_bwd_with_duplicate_voxels(C):
    """x + LN(Linear(SubMConv3d(x))) — the Block.cpe sub-block — forward and backward on augmented clouds against the
    oracle under autograd (fp64), every input and parameter gradient."""
    ops = _ops()
    batch, ref, got = _dup_cloud_levels(seed=90 + C)
    n, lvl = got[0].n, got[0]
    g = torch.Generator().manual_seed(C)
    x = torch.randn(n, C, generator=g)
    cw = torch.randn(C, 3, 3, 3, C, generator=g) / (C * 9) ** 0.5
    cb, lw, lb = torch.randn(C, generator=g) * 0.1, torch.randn(C, C, generator=g) / C ** 0.5, torch.randn(C, generator=g) * 0.1
    gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    dy = torch.randn(n, C, generator=g)
    dev = [t.cuda().requires_grad_(True) for t in (x, cw, cb, lw, lb, gam, bet)]
    y = ops.CpeFn.apply(dev[0], dev[0], *dev[1:], lvl)
    y.backward(dy.cuda())
    torch.cuda.synchronize()
    dbl = [t.double().requires_grad_(True) for t in (x, cw, cb, lw, lb, gam, bet)]
    nbr_ref = torch.from_numpy(ref[0]["nbr27"]).long()
    c = om.subm_conv(dbl[0], nbr_ref, dbl[1], dbl[2])
    yref = dbl[0] + F.layer_norm(c @ dbl[3].t() + dbl[4], (C,), dbl[5], dbl[6], 1e-5)
    yref.backward(dy.double())
    _close(y, yref, 3e-6, "cpe fwd")
    for name, a, b in zip(("dx", "dconv_w", "dconv_b", "dlin_w", "dlin_b", "dgamma", "dbeta"), dev, dbl):
        _close(a.grad, b.grad, 3e-6 if name == "dx" else 2e-5, name)


@pytest.mark.parametrize("prec", [1, 3])
@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 128), (256, 256), (64, 128)])
def test_subm_conv_bf16_operand_paths(prec, cin, cout):
    """Pair-compacted 3^3 convolution (fwd + dgrad) with bf16 (1) / bf16x3 (3) operands: against the fp64 convolution of
    the bf16-ROUNDED operands (products of two bf16 are exact in fp32) resp. of the exact operands within 2^-16."""
    from robot_3dlotus_amd import _capi
    ops = _ops()
    batch, ref, got = _cloud_levels(3, 1500, seed=cin + 3)
    n = got[0].n
    g = torch.Generator().manual_seed(cin * cout + prec)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cout, 3, 3, 3, cin, generator=g) / (cin * 9) ** 0.5
    dy = torch.randn(n, cout, generator=g)
    nbr_ref = torch.from_numpy(ref[0]["nbr27"]).long()
    ops.set_gemm_precision({1: "bf16", 3: "bf16x3"}[prec])
    try:
        wt = ops.conv_weight_t(w.cuda())
        y = ops.conv_fwd(x.cuda(), w.cuda(), None, got[0].nbr27, got[0].order[0], w_t=wt)
        dx = ops.conv_dgrad(dy.cuda(), w.cuda(), got[0].nbr27, got[0].order[0], w_t=wt)
        dw, db = ops.conv_wgrad(dy.cuda(), x.cuda(), w.shape, got[0].nbr27)
    finally:
        ops.set_gemm_precision("fp32")
    rnd = (lambda t: t.bfloat16().double()) if prec == 1 else (lambda t: t.double())
    xd, wd = rnd(x).requires_grad_(True), rnd(w).requires_grad_(True)
    yref = om.subm_conv(xd, nbr_ref, wd, None)
    dxref, dwref = torch.autograd.grad(yref, (xd, wd), rnd(dy))
    sy = om.subm_conv(x.abs().double(), nbr_ref, w.abs().double(), None).clamp_min(1.0)
    tol = 2e-6 if prec == 1 else 2.0 ** -16
    ey = ((y.cpu().double() - yref.detach()).abs() / sy).max().item()
    assert ey <= tol, (prec, "fwd", ey)
    xa = x.abs().double().requires_grad_(True)
    (sx,) = torch.autograd.grad(om.subm_conv(xa, nbr_ref, w.abs().double(), None), xa, dy.abs().double())
    ex = ((dx.cpu().double() - dxref).abs() / sx.clamp_min(1.0)).max().item()
    assert ex <= tol, (prec, "dgrad", ex)
    # weight gradient (same operand modes; ~n / 3 active pairs per tap are summed in fp32) and the bias gradient, which is
    # the column sum of the UNROUNDED dy rows in every mode
    wa = w.abs().double().requires_grad_(True)
    (sw,) = torch.autograd.grad(om.subm_conv(x.abs().double(), nbr_ref, wa, None), wa, dy.abs().double())
    ew = ((dw.cpu().double() - dwref).abs() / sw.clamp_min(1.0)).max().item()
    assert ew <= tol, (prec, "wgrad", ew)
    eb = ((db.cpu().double() - dy.double().sum(0)).abs() / dy.abs().double().sum(0)).max().item()
    assert eb <= 2e-6, (prec, "bgrad", eb)


def test_subm_conv_bf16_operands_pair_compacted_path():
    """The bf16 operand modes on the PAIR-COMPACTED kernel (LOTUS_CONV_OS=0; the default for them is the output-stationary
    kernel): same cases, same fp64 references, so the hi-only / hi + lo weight packings stay covered on both kernels.  The
    switch is read once per process, so the cases run in a child interpreter."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_ops.py"), "-q", "-m", "gpu", "-x",
                        "-k", "test_subm_conv_bf16_operand_paths"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, LOTUS_CONV_OS="0"), cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    # and the converse: exact fp32 products on the output-stationary kernel (LOTUS_CONV_OS_F32=1; opt-in — it only wins
    # on the 64-wide layers and is worth +0.5 % of the step, csrc/conv_pairs.hip) against the same fp64 references
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_ops.py"), "-q", "-m", "gpu", "-x",
                        "-k", "test_subm_conv_fwd_dgrad_wgrad or duplicate"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, LOTUS_CONV_OS_F32="1"), cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


# ------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("C,H", [(64, 2), (128, 4), (768, 32)])
def test_patch_attention_fwd_bwd(C, H):
    ops = _ops()
    batch, ref, got = _cloud_levels(3, 300, seed=C)
    lv, r = got[0], ref[0]
    n, d = lv.n, C // H
    g = torch.Generator().manual_seed(C)
    qkv = torch.randn(n, 3 * C, generator=g) * 1.5
    qn = (torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.2)
    kn = (torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.2)
    dout = torch.randn(n, C, generator=g)
    lvl = dict(order_t=torch.from_numpy(r["order"]), inverse_t=torch.from_numpy(r["inverse"]),
               pad_t=torch.from_numpy(r["pad"]), unpad_t=torch.from_numpy(r["unpad"]), cu_seqlens=r["cu_seqlens"])
    qd = qkv.double().requires_grad_(True)
    pr = [t.double().requires_grad_(True) for t in (*qn, *kn)]
    oref = om.patch_attention(qd, lvl, 0, H, pr[0], pr[1], pr[2], pr[3], 128)
    oref.backward(dout.double())
    dev = lambda t: t.cuda()  # noqa: E731
    qc = dev(qkv)
    qnc, knc = tuple(map(dev, qn)), tuple(map(dev, kn))
    att = torch.empty(n, C, device="cuda")
    lse = torch.empty(lv.npad, H, device="cuda")
    ops.attention_fwd(qc, 3 * C, 0, qc, 3 * C, C, 2 * C, lv.gidx, lv.gidx, lv.owner, lv.self_tiles, lv.n_self_tiles,
                      qnc, knc, att, lse, H, d)
    _close(att, oref, 5e-6, "attn fwd")
    dqkv = torch.empty(n, 3 * C, device="cuda")
    extra = torch.empty(max(lv.n_extra, 1), 2 * C, device="cuda")
    gr = ops.attention_bwd(qc, 3 * C, 0, qc, 3 * C, C, 2 * C, lv.gidx, lv.gidx, lv.owner, lv.self_tiles, lv.self_blocks,
                           lv.n_self_tiles, qnc, knc, att, dev(dout), lse, dqkv, 3 * C, 0, dqkv, 3 * C, C, 2 * C, 0, 0, H, d, 0.0, 0,
                           lv.kext, lv.ext_pos, lv.n_extra, extra)
    _close(dqkv, qd.grad, 2e-5, "attn dqkv")
    for name, a, b in zip(("dqn_w", "dqn_b", "dkn_w", "dkn_b"), gr, pr):
        # dkn_b is mathematically zero (softmax is shift-invariant per query): pure fp32 cancellation noise
        _close(a, b.grad, 5e-4 if name == "dkn_b" else 5e-5, name)


def test_patch_attention_query_per_lane_path():
    """The query-per-lane kernels on the 128-key patch attention (LOTUS_XQ=2; slower than the tile kernels there, hence not
    the default — csrc/attention.hip): gathered rows, owner flags, borrowed tail-patch copies and four key chunks per tile,
    against the same fp64 formulation.  The switch is read once per process, so the cases run in a child interpreter."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_ops.py"), "-q", "-m", "gpu", "-x",
                        "-k", "test_patch_attention_fwd_bwd or test_cross_attention_fwd_bwd"], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, LOTUS_XQ="2"), cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def test_cross_attention_round5_short_key_kernels_still_agree():
    """LOTUS_XQ=3 routes the 32-wide heads of the cross attention back to the query-per-lane kernels of round 5 (the A/B partner
    of xq2_fwd_kernel / xq2_bwd_kernel, csrc/attention.hip): they stay in the library as the path of the 16- and 24-wide heads, so
    they keep the same fp64 check on the 32-wide shapes too.  Read once per process, hence the child interpreter."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_ops.py"), "-q", "-m", "gpu", "-x",
                        "-k", "test_cross_attention_fwd_bwd"], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, LOTUS_XQ="3"), cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("prec,tol", [(3, 1e-4), (1, 4e-2)])
@pytest.mark.parametrize("C,H", [(64, 2), (128, 4), (768, 32)])
def test_patch_attention_bf16_operand_paths(prec, tol, C, H):
    """Tile attention forward and backward with bf16x3 (3) / bf16 (1) operands against the fp64 reference (head dims 32 and 24)."""
    from robot_3dlotus_amd import _capi
    ops = _ops()
    batch, ref, got = _cloud_levels(3, 300, seed=C)
    lv, r = got[0], ref[0]
    n, d = lv.n, C // H
    g = torch.Generator().manual_seed(C + prec)
    qkv = torch.randn(n, 3 * C, generator=g) * 1.5
    qn = (torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.2)
    kn = (torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.2)
    lvl = dict(order_t=torch.from_numpy(r["order"]), inverse_t=torch.from_numpy(r["inverse"]),
               pad_t=torch.from_numpy(r["pad"]), unpad_t=torch.from_numpy(r["unpad"]), cu_seqlens=r["cu_seqlens"])
    oref = om.patch_attention(qkv.double(), lvl, 0, H, *(t.double() for t in (*qn, *kn)), 128)
    qc = qkv.cuda()
    qnc, knc = tuple(t.cuda() for t in qn), tuple(t.cuda() for t in kn)
    att = torch.empty(n, C, device="cuda")
    lse = torch.empty(lv.npad, H, device="cuda")
    dout = torch.randn(n, C, generator=g)
    qd = qkv.double().requires_grad_(True)
    pr = [t.double().requires_grad_(True) for t in (*qn, *kn)]
    om.patch_attention(qd, lvl, 0, H, pr[0], pr[1], pr[2], pr[3], 128).backward(dout.double())
    dqkv = torch.empty(n, 3 * C, device="cuda")
    extra = torch.empty(max(lv.n_extra, 1), 2 * C, device="cuda")
    ops.set_gemm_precision({1: "bf16", 3: "bf16x3"}[prec])
    try:
        ops.attention_fwd(qc, 3 * C, 0, qc, 3 * C, C, 2 * C, lv.gidx, lv.gidx, lv.owner, lv.self_tiles, lv.n_self_tiles,
                          qnc, knc, att, lse, H, d)
        gr = ops.attention_bwd(qc, 3 * C, 0, qc, 3 * C, C, 2 * C, lv.gidx, lv.gidx, lv.owner, lv.self_tiles, lv.self_blocks,
                               lv.n_self_tiles, qnc, knc, att, dout.cuda(), lse, dqkv, 3 * C, 0, dqkv, 3 * C, C, 2 * C, 0, 0,
                               H, d, 0.0, 0, lv.kext, lv.ext_pos, lv.n_extra, extra)
    finally:
        ops.set_gemm_precision("fp32")
    _close(att, oref, tol, f"attn fwd prec {prec}")
    _close(dqkv, qd.grad, 4 * tol, f"attn dqkv prec {prec}")
    for name, a, b in zip(("dqn_w", "dqn_b", "dkn_w", "dkn_b"), gr, pr):
        _close(a, b.grad, 10 * tol, f"{name} prec {prec}")


@pytest.mark.parametrize("path", ["tile", "short_keys"])
@pytest.mark.parametrize("C,H,drop", [(64, 2, 0.0), (768, 32, 0.0), (128, 4, 0.0), (128, 4, 0.25)])
def test_cross_attention_fwd_bwd(C, H, drop, path):
    """Both kernel families of the cross attention against the fp64 formulation: the 128 x 128 tile kernels (k_max = 0) and
    the short-key kernels (one lane per query; k_max = longest instruction <= 32).  With dropout on the probabilities the
    reference cannot be evaluated (hash masks), so the two families — which share the mask index — are compared with each
    other, and backward is checked against central differences of the forward along a random direction."""
    ops = _ops()
    batch, ref, got = _cloud_levels(3, 700, seed=C + 1)
    lv = got[0]
    n, d = lv.n, C // H
    counts, ctx_counts = batch["npoints_in_batch"], batch["txt_lens"]
    assert 0 < lv.ca_kmax <= 32 and lv.ca_kmax == max(ctx_counts)
    k_max = lv.ca_kmax if path == "short_keys" else 0
    L = sum(ctx_counts)
    g = torch.Generator().manual_seed(C + 5)
    q, kv = torch.randn(n, C, generator=g) * 1.5, torch.randn(L, 2 * C, generator=g) * 1.5
    qn = (torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.2)
    kn = (torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.2)
    dout = torch.randn(n, C, generator=g)
    qc, kvc = q.cuda(), kv.cuda()
    qnc, knc = tuple(t.cuda() for t in qn), tuple(t.cuda() for t in kn)
    G = lv.ca_groups

    def run(km, qq=qc, kk=kvc):
        att = torch.empty(n, C, device="cuda")
        lse = torch.empty(n, H, device="cuda")
        ops.attention_fwd(qq, C, 0, kk, 2 * C, 0, C, None, None, None, lv.ca_tiles, lv.n_ca_tiles, qnc, knc, att, lse, H, d,
                          drop_p=drop, seed=77, k_max=km)
        return att, lse

    def back(km, att, lse):
        dq = torch.empty(n, C, device="cuda")
        dkvp = torch.empty(G, L, 2 * C, device="cuda")
        gr = ops.attention_bwd(qc, C, 0, kvc, 2 * C, 0, C, None, None, None, lv.ca_tiles, lv.ca_blocks, lv.n_ca_blocks, qnc,
                               knc, att, dout.cuda(), lse, dq, C, 0, dkvp, 2 * C, 0, C, L * 2 * C, 0, H, d, drop_p=drop, seed=77,
                               k_max=km)
        return dq, dkvp.sum(0), gr

    att, lse = run(k_max)
    dq, dkv, gr = back(k_max, att, lse)
    if drop == 0.0:
        qd, kvd = q.double().requires_grad_(True), kv.double().requires_grad_(True)
        pr = [t.double().requires_grad_(True) for t in (*qn, *kn)]
        oref = om.cross_attention(qd, kvd, counts, ctx_counts, H, pr[0], pr[1], pr[2], pr[3])
        oref.backward(dout.double())
        _close(att, oref, 5e-6, "xattn fwd")
        _close(dq, qd.grad, 2e-5, "xattn dq")
        _close(dkv, kvd.grad, 2e-5, "xattn dkv")
        for name, a, b in zip(("dqn_w", "dqn_b", "dkn_w", "dkn_b"), gr, pr):
            # dkn_b is mathematically zero (softmax is shift-invariant per query): pure fp32 cancellation noise
            _close(a, b.grad, 5e-4 if name == "dkn_b" else 5e-5, name)
    else:
        att0, lse0 = run(0)
        dq0, dkv0, gr0 = back(0, att0, lse0)
        _close(att, att0, 5e-6, "xattn fwd (dropout) vs tile kernels")
        _close(lse, lse0, 5e-6, "xattn lse")
        _close(dq, dq0, 2e-5, "xattn dq (dropout)")
        _close(dkv, dkv0, 2e-5, "xattn dkv (dropout)")
        for name, a, b in zip(("dqn_w", "dqn_b", "dkn_w", "dkn_b"), gr, gr0):
            _close(a, b, 5e-4 if name == "dkn_b" else 5e-5, name + " (dropout)")
        keep = float((att != 0).float().mean())
        assert keep > 0.99  # outputs are sums over the kept keys: essentially never all dropped
        # directional derivative of <out, dout> along a random direction of (q, kv)
        gdir = torch.Generator(device="cuda").manual_seed(1)
        vq, vk = torch.randn(n, C, device="cuda", generator=gdir), torch.randn(L, 2 * C, device="cuda", generator=gdir)
        want = float((dq.double() * vq.double()).sum() + (dkv.double() * vk.double()).sum())
        eps = 1e-2
        fp = float((run(k_max, qc + eps * vq, kvc + eps * vk)[0].double() * dout.cuda().double()).sum())
        fm = float((run(k_max, qc - eps * vq, kvc - eps * vk)[0].double() * dout.cuda().double()).sum())
        fd = (fp - fm) / (2 * eps)
        assert abs(fd - want) <= 2e-3 * max(abs(want), abs(fd)) + 1e-2, (fd, want)


# ------------------------------------------------------------------------------------ pool / head / loss
def test_pool_unpool():
    ops = _ops()
    from robot_3dlotus_amd._capi import call

    batch, ref, got = _cloud_levels(2, 900, seed=3)
    parent, child = got[0], got[1]
    C = 64
    g = torch.Generator().manual_seed(0)
    x = torch.randn(parent.n, C, generator=g)
    cl = torch.from_numpy(ref[1]["cluster"])
    yref = torch.zeros(child.n, C).scatter_reduce(0, cl.view(-1, 1).expand(-1, C), x, "amax", include_self=False)
    y = torch.empty(child.n, C, device="cuda")
    arg = torch.empty(child.n, C, dtype=torch.int32, device="cuda")
    call("lotus_pool_max_fwd", x.cuda(), child.members, child.seg_start, child.n, C, y, arg)
    assert torch.equal(y.cpu(), yref)
    assert torch.equal(x[arg.cpu().long(), torch.arange(C)], yref)
    dy = torch.randn(child.n, C, generator=g)
    dx = torch.empty(parent.n, C, device="cuda")
    call("lotus_pool_max_bwd", dy.cuda(), arg, child.cluster, parent.n, C, dx)
    xr = x.clone().requires_grad_(True)
    torch.zeros(child.n, C).scatter_reduce(0, cl.view(-1, 1).expand(-1, C), xr, "amax", include_self=False).backward(dy)
    assert torch.equal(dx.cpu(), xr.grad)
    up = torch.randn(child.n, C, generator=g)
    o = torch.empty(parent.n, C, device="cuda")
    call("lotus_unpool_fwd", x.cuda(), up.cuda(), child.cluster, parent.n, C, o)
    assert torch.equal(o.cpu(), x + up[cl])
    dup = torch.empty(child.n, C, device="cuda")
    call("lotus_unpool_bwd", x.cuda(), child.members, child.seg_start, child.n, C, dup)
    _close(dup, torch.zeros(child.n, C).index_add(0, cl, x), 2e-6, "unpool bwd")


def test_head_and_losses():
    ops = _ops()
    from robot_3dlotus_amd import synth
    from robot_3dlotus_amd.frontend import FrontEnd

    batch = synth.synth_batch(3, 500, ragged=True, seed=8)
    lv = FrontEnd(2).build(batch["pc_fts"].cuda(), batch["npoints_in_batch"], batch["txt_lens"], [[0, 1, 2, 3]] * 2)[0]
    counts = batch["npoints_in_batch"]
    N, C, B = lv.n, 128, len(counts)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(N, C, generator=g)
    ws = [torch.randn(C, C, generator=g) / 11, torch.randn(C, generator=g) * 0.1, torch.randn(90, C, generator=g) / 11,
          torch.randn(90, generator=g) * 0.1, torch.randn(C, C, generator=g) / 11, torch.randn(C, generator=g) * 0.1,
          torch.randn(217, C, generator=g) / 11, torch.randn(217, generator=g) * 0.1]
    gt = batch["gt_actions"]
    tgt = torch.cat([t.reshape(-1) for t in batch["disc_pos_probs"]])
    xd = x.double().requires_grad_(True)
    wd = [w.double().requires_grad_(True) for w in ws]
    h = F.leaky_relu(F.linear(xd, wd[0], wd[1]), 0.02)
    xt = F.linear(h, wd[2], wd[3]).view(-1, 3, 30).permute(1, 0, 2)
    pcs = torch.stack([t.max(0)[0] for t in torch.split(xd, counts)], 0)
    ae = F.linear(F.leaky_relu(F.linear(pcs, wd[4], wd[5]), 0.02), wd[6], wd[7])
    pos = sum(F.cross_entropy(lg.reshape(3, -1), tg.double()) for lg, tg in zip(torch.split(xt, counts, 1), batch["disc_pos_probs"])) / B
    rot = F.cross_entropy(ae[:, :216].reshape(-1, 72, 3), gt[:, 3:6].long())
    opn = F.binary_cross_entropy_with_logits(ae[:, -1], gt[:, -1].double())
    total = pos + rot + opn
    total.backward()
    xc = x.cuda().requires_grad_(True)
    wc = [w.cuda().requires_grad_(True) for w in ws]
    losses, xt_g, ae_g = ops.HeadLossFn.apply(xc, *wc, lv, tgt.cuda(), gt.cuda(), 1.0, 1.0, 0.0, 0, True)
    _close(losses, torch.stack([pos, rot, opn, total]), 3e-6, "losses")
    _close(ae_g, ae, 3e-6, "ae")
    losses[3].backward()
    _close(xc.grad, xd.grad, 1e-5, "head dx")
    for i, (a, b) in enumerate(zip(wc, wd)):
        _close(a.grad, b.grad, 1e-5, f"head param {i}")


def test_attention_dropout_mask_is_consistent_between_fwd_and_bwd():
    """attn_drop (flash-attn dropout_p): the counter-based mask must be the same in forward and backward.
    Checked through (i) E[sum_k P~] = 1, (ii) the adjoint identity <dO, P~ dV_dir> = <dV, dV_dir> (out is linear
    in V), (iii) a central-difference directional derivative w.r.t. q."""
    ops = _ops()
    batch, ref, got = _cloud_levels(2, 300, seed=11)
    lv = got[0]
    C, H, p_drop, seed = 64, 2, 0.3, 99
    n, d = lv.n, C // H
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(n, 3 * C, generator=g)).cuda()
    qn = (torch.ones(d).cuda(), torch.zeros(d).cuda())

    def fwd(t):
        att = torch.empty(n, C, device="cuda")
        lse = torch.empty(lv.npad, H, device="cuda")
        ops.attention_fwd(t, 3 * C, 0, t, 3 * C, C, 2 * C, lv.gidx, lv.gidx, lv.owner, lv.self_tiles, lv.n_self_tiles,
                          qn, qn, att, lse, H, d, p_drop, seed)
        return att, lse

    ones = qkv.clone()
    ones[:, 2 * C:] = 1.0
    att1, _ = fwd(ones)
    assert abs(att1.mean().item() - 1.0) < 0.02 and att1.std().item() > 0.01
    att, lse = fwd(qkv)
    att_b, _ = fwd(qkv)
    assert torch.equal(att, att_b)
    dout = torch.randn(n, C, generator=g).cuda()
    dqkv = torch.empty(n, 3 * C, device="cuda")
    extra = torch.empty(max(lv.n_extra, 1), 2 * C, device="cuda")
    ops.attention_bwd(qkv, 3 * C, 0, qkv, 3 * C, C, 2 * C, lv.gidx, lv.gidx, lv.owner, lv.self_tiles, lv.self_blocks,
                      lv.n_self_tiles, qn, qn, att, dout, lse, dqkv, 3 * C, 0, dqkv, 3 * C, C, 2 * C, 0, 0, H, d, p_drop, seed,
                      lv.kext, lv.ext_pos, lv.n_extra, extra)
    u = torch.randn(n, 3 * C, generator=g).cuda()
    uv = torch.zeros_like(u)
    uv[:, 2 * C:] = u[:, 2 * C:]
    lhs = ((fwd(qkv + uv)[0] - att) * dout).double().sum().item()
    rhs = (dqkv * uv).double().sum().item()
    assert abs(lhs - rhs) <= 2e-4 * max(1.0, abs(rhs)), (lhs, rhs)
    uq = torch.zeros_like(u)
    uq[:, :2 * C] = u[:, :2 * C]
    eps = 1e-2
    num = (((fwd(qkv + eps * uq)[0] - fwd(qkv - eps * uq)[0]) / (2 * eps)) * dout).double().sum().item()
    ana = (dqkv * uq).double().sum().item()
    assert abs(num - ana) <= 3e-2 * max(1.0, abs(ana)), (num, ana)


def test_cloud_max_first_index_ties_and_ragged_clouds():
    """lotus_cloud_max_fwd == torch.max(x_b, 0) per cloud, values AND indices (first row attaining the maximum), with
    deliberate ties, clouds shorter than the row splits and non-finite rows; backward scatters to those rows."""
    from types import SimpleNamespace
    from robot_3dlotus_amd import ops

    torch.manual_seed(5)
    counts = [1, 7, 300, 4096, 33, 2050]
    C = 128
    x = torch.randn(sum(counts), C, device="cuda")
    x = (x * 4).round() / 4                                   # coarse values: many exact ties inside every column
    off = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device="cuda")
    x[off[3] + 5, :] = float("-inf")
    x[off[2]:off[2] + 300, 3] = 2.5                           # a whole column tied
    batch = torch.repeat_interleave(torch.arange(len(counts), device="cuda", dtype=torch.int32), torch.tensor(counts, device="cuda"))
    lvl = SimpleNamespace(counts=counts, off=off, batch=batch)
    xr = x.clone().requires_grad_(True)
    y = ops.CloudMaxFn.apply(xr, lvl)
    ref_v, ref_i = zip(*[t.max(0) for t in torch.split(x, counts)])
    assert torch.equal(y, torch.stack(ref_v))
    g = torch.randn_like(y)
    y.backward(g)
    want = torch.zeros_like(x)
    for b, (i, o) in enumerate(zip(ref_i, off[:-1].tolist())):
        want[i + o, torch.arange(C, device="cuda")] += g[b]
    assert torch.equal(xr.grad, want)


@pytest.mark.parametrize("M,N,K", [(361, 768, 3072), (1450, 512, 2048), (441, 3072, 768), (1727, 512, 2048)])
def test_fused_splitk_handoff_equals_two_launch_path(M, N, K):
    """The fused split-K hand-off (partials through sc1 stores / loads + an arrival counter, csrc/gemm.hip "HARDWARE CONTRACT")
    against the two-launch path (partials, kernel boundary, reduction kernel) on the deep-level shapes that split: forward and
    input-gradient products sum their partials in the same z order -> bit-identical; repeated 20 times under load from a
    second stream (the hand-off must not depend on timing).  LOTUS_SPLITK_FUSED=0 selects the two-launch path globally."""
    from robot_3dlotus_amd import ops

    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    nb = ops.query("lotus_linear_workspace", M, N, K)
    assert nb > 0, "shape was meant to take the split-K path"
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(ops.query("lotus_splitk_counters_bytes"), dtype=torch.uint8, device="cuda")

    def fwd(counters):
        y = torch.empty(M, N, device="cuda")
        ops.call("lotus_linear_fwd", x, w, b, res, y, None, M, N, K, ops.ACT_GELU, 0.0, 0, 0, ws, nb, counters)
        return y

    ref = fwd(None)
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device="cuda")
    for it in range(20):
        with torch.cuda.stream(side):   # uneven load on the CUs while the hand-off runs
            big = big @ big * 1e-3
        y = fwd(cnt)
        assert torch.equal(y, ref), f"fused split-K differs from the two-launch path (iteration {it})"
    assert int(cnt.view(torch.int32).abs().sum()) == 0, "arrival counters must be left at zero"
    torch.cuda.synchronize()

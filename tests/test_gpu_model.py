"""End-to-end GPU parity of the drop-in policy against (a) the golden fixtures captured from the
imported reference and (b) the oracle run live on the host CPU.

Tolerance (north star): fp32 action logits within 1e-4 of the fp32-ideal reference — applied as
1e-4 * max(1, max|logit|) because the scaled-weights fixtures have logits of magnitude 2..14."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import golden_util as gu  # noqa: E402

import ledger  # noqa: E402

LOGIT_TOL = 1e-4
# gradients: ||dg|| <= GRAD_TOL * (||g|| + GRAD_FLOOR * max_p ||g_p||) for every parameter, whole tensor (round 2 had 2e-3 on
# the norm and the first 48 entries only); measured errors are in profiles/r03_parity.json
GRAD_TOL = 1e-4
GRAD_FLOOR = 1e-3


def _build(cfg, sd, train):
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA

    m = SimplePolicyPTV3CA(cfg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    m.train(train)
    m.ptv3_model.proj_drop = m.ptv3_model.attn_drop = 0.0      # fixtures are dropout-free
    m.act_proj_head.dropout = 0.0
    return m


def _dev_batch(batch):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if k == "disc_pos_probs" else v))
            for k, v in batch.items()}


@pytest.mark.parametrize("case", gu.CASES)
def test_golden_fixture_parity(case):
    from robot_3dlotus_amd import config as lcfg

    fx, cfg, batch, sd = gu.load_case(case, gu.state_template)
    train = bool(fx["meta_train"])
    m = _build(cfg, sd, train)
    m.ptv3_model.order_perms = [p.tolist() for p in fx["perms"]]
    _, losses = m(_dev_batch(batch), compute_loss=True, compute_final_action=False)
    xt, xr, xo = m.last_pred
    errs = {}
    for name, got in (("xt", xt), ("xr", xr), ("xo", xo)):
        ref = fx[name]
        tol = LOGIT_TOL * max(1.0, float(np.abs(ref).max()))
        err = float(np.abs(got.detach().cpu().numpy() - ref).max())
        errs["logit_abs_err_" + name], errs["logit_max_" + name] = err, float(np.abs(ref).max())
        assert err <= tol, f"{case} {name}: max |diff| {err:.3e} > {tol:.3e}"
    for k in ("pos", "rot", "open", "total"):
        ref = float(fx["loss_" + k])
        errs["loss_abs_err_" + k] = abs(losses[k].item() - ref)
        assert abs(losses[k].item() - ref) <= 1e-4 * max(1.0, abs(ref)), (k, losses[k].item(), ref)
    losses["total"].backward()
    gmax = max(float(fx[k]) for k in fx if k.startswith("gnorm/"))
    worst, worst_head = (0.0, None), (0.0, None)
    for name, p in m.named_parameters():
        assert p.grad is not None, f"no gradient for {name}"
        ref = float(fx["gnorm/" + name])
        got = p.grad.double().norm().item()
        rel = abs(got - ref) / (ref + GRAD_FLOOR * gmax)
        worst = max(worst, (rel, name))
        head = fx["ghead/" + name]
        herr = float(np.abs(p.grad.flatten()[:48].cpu().numpy() - head).max()) / (float(np.abs(head).max()) + GRAD_FLOOR * gmax)
        worst_head = max(worst_head, (herr, name))
    # the fixture stores norms + leading entries of the reference's gradients; the WHOLE gradient of every parameter is
    # compared with the oracle run live on the fixture's inputs (the oracle itself is pinned to the reference's gradients
    # at 1e-4 by tests/test_oracle_vs_reference.py)
    from oracle.model import Oracle
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    out = Oracle(sdg, lcfg.plain(cfg), training=train).forward(batch, [p.tolist() for p in fx["perms"]])
    out["losses"]["total"].backward()
    og = max(float(v.grad.norm()) for v in sdg.values() if v.grad is not None)
    worst_full = (0.0, None)
    for name, p in m.named_parameters():
        r = sdg[name].grad
        rel = float((p.grad.cpu().double() - r.double()).norm()) / (float(r.norm()) + GRAD_FLOOR * og)
        worst_full = max(worst_full, (rel, name))
    ledger.record("golden_fixture/" + case, grad_norm_rel_err_max=worst[0], grad_norm_argmax=worst[1],
                  grad_head_rel_err_max=worst_head[0], grad_head_argmax=worst_head[1],
                  grad_full_vs_oracle_rel_err_max=worst_full[0], grad_full_argmax=worst_full[1], **errs)
    assert worst[0] < GRAD_TOL, f"gradient norm mismatch {worst}"
    # (the stored leading entries are the reference's own fp32 values: 2.8e-4 apart on one bias of v1_init_train, where the
    # whole tensor agrees with the oracle to 1.4e-5 — the strong check is the next line)
    assert worst_head[0] < 1e-3, f"gradient entries mismatch {worst_head}"
    assert worst_full[0] < GRAD_TOL, f"whole-gradient mismatch against the oracle {worst_full}"
    if train:
        sdn = m.state_dict()
        for k in fx:
            if k.startswith("buf/"):
                np.testing.assert_allclose(sdn[k[4:]].cpu().numpy(), fx[k], atol=2e-3, rtol=2e-3, err_msg=k)


def test_live_oracle_parity_and_determinism():
    """Fresh seeded inputs (not in any fixture): HIP model vs the oracle on the host CPU, and
    bit-identical repeat of the forward pass."""
    from oracle.model import Oracle
    from robot_3dlotus_amd import config as lcfg, synth
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("v1")
    sd = seeded_state_dict(gu.state_template(cfg), 77, "scaled")
    batch = synth.synth_batch(3, 900, ragged=True, seed=123)
    perms = [[1, 3, 0, 2], [0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1], [1, 0, 2, 3]]
    out = Oracle({k: v.clone() for k, v in sd.items()}, lcfg.plain(cfg), training=True).forward(batch, perms)
    m = _build(cfg, sd, True)
    m.ptv3_model.order_perms = perms
    _, losses = m(_dev_batch(batch), compute_loss=True, compute_final_action=False)
    xt = m.last_pred[0].detach().clone()
    ref = out["xt"].numpy()
    assert float(np.abs(xt.cpu().numpy() - ref).max()) <= LOGIT_TOL * max(1.0, float(np.abs(ref).max()))
    assert abs(losses["total"].item() - out["losses"]["total"].item()) < 1e-4 * max(1.0, abs(out["losses"]["total"].item()))
    m2 = _build(cfg, sd, True)
    m2.ptv3_model.order_perms = perms
    m2(_dev_batch(batch), compute_loss=True, compute_final_action=False)
    assert torch.equal(m2.last_pred[0], xt), "forward must be bit-reproducible"


def test_live_oracle_parity_with_duplicate_voxels_backward():
    """Augmented clouds (z rotation + jitter -> 1-7 % of the points share a voxel, SURVEY.md Trap 5): logits, losses and
    EVERY parameter gradient of the HIP model against the oracle under autograd."""
    from oracle.model import Oracle
    from robot_3dlotus_amd import config as lcfg, synth
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("tiny")
    sd = seeded_state_dict(gu.state_template(cfg), 21, "scaled")
    batch = synth.augment_clouds(synth.synth_batch(3, 700, ragged=True, seed=321), seed=5)
    perms = [[1, 3, 0, 2], [2, 0, 3, 1]]
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    out = Oracle(sdg, lcfg.plain(cfg), training=True).forward(batch, perms)
    out["losses"]["total"].backward()
    m = _build(cfg, sd, True)
    m.ptv3_model.order_perms = perms
    _, losses = m(_dev_batch(batch), compute_loss=True, compute_final_action=False)
    assert m.ptv3_model.frontend is not None
    ref = out["xt"].detach().numpy()
    assert float(np.abs(m.last_pred[0].detach().cpu().numpy() - ref).max()) <= LOGIT_TOL * max(1.0, float(np.abs(ref).max()))
    assert abs(losses["total"].item() - out["losses"]["total"].item()) < 1e-4 * max(1.0, abs(out["losses"]["total"].item()))
    losses["total"].backward()
    gmax = max(float(v.grad.norm()) for v in sdg.values() if v.grad is not None)
    worst = (0.0, None)
    for name, p in m.named_parameters():
        r = sdg[name].grad
        err = float((p.grad.cpu() - r).norm())
        worst = max(worst, (err / (float(r.norm()) + GRAD_FLOOR * gmax), name))
    ledger.record("live_oracle/tiny_duplicate_voxels", grad_full_vs_oracle_rel_err_max=worst[0], grad_full_argmax=worst[1],
                  n_dup=int(m.ptv3_model.last_n_dup))
    assert worst[0] <= GRAD_TOL, f"whole-gradient mismatch against the oracle {worst}"


@pytest.mark.parametrize("mode,tol", [("bf16x3", 1e-4), ("bf16", 3e-2)])
def test_gemm_precision_modes_against_golden(mode, tol):
    """Opt-in operand precisions of the dense fwd/dgrad products (ops.set_gemm_precision).  'bf16x3' must still meet
    the north-star bar (fp32 logits within 1e-4 of the fp32-ideal reference); 'bf16' is the bf16 compute mode of
    BASELINE configs[4] (GEMMs only so far) and is held to a bf16-sized bound.  The default mode is restored."""
    from robot_3dlotus_amd import config as lcfg, ops

    fx, cfg, batch, sd = gu.load_case("v1_scaled_train", gu.state_template(lcfg.preset("v1")))
    assert ops.get_gemm_precision() == "fp32"
    ops.set_gemm_precision(mode)
    try:
        m = _build(cfg, sd, True)
        m.ptv3_model.order_perms = [p.tolist() for p in fx["perms"]]
        _, losses = m(_dev_batch(batch), compute_loss=True, compute_final_action=False)
        losses["total"].backward()
        xt, xr, xo = m.last_pred
    finally:
        ops.set_gemm_precision("fp32")
    errs = {}
    for name, got in (("xt", xt), ("xr", xr), ("xo", xo)):
        ref = fx[name]
        errs[name] = float(np.abs(got.detach().cpu().numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
        assert errs[name] <= tol, (mode, name, errs[name])
    assert abs(losses["total"].item() - float(fx["loss_total"])) <= 10 * tol * abs(float(fx["loss_total"]))
    gmax = max(float(fx[k]) for k in fx if k.startswith("gnorm/"))
    worst = max(abs(p.grad.double().norm().item() - float(fx["gnorm/" + n])) / (float(fx["gnorm/" + n]) + 1e-3 * gmax)
                for n, p in m.named_parameters())
    assert worst < (2e-2 if mode == "bf16x3" else 0.5), (mode, worst)
    print(f"{mode}: logit errors {errs}, worst gradient-norm deviation {worst:.2e}")


def test_peract_config_bf16_mode_dense_clouds():
    """BASELINE configs[4] as a parity case: PerAct preset (the v1 network), dense 4096-point clouds, bf16 operand mode
    of the dense / sparse-convolution / attention products, against the fp32 oracle at a bf16-sized bound; the exact
    mode on the same inputs stays inside the 1e-4 bar."""
    from oracle.model import Oracle
    from robot_3dlotus_amd import config as lcfg, ops, synth
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("peract")
    sd = seeded_state_dict(gu.state_template(cfg), 91, "scaled")
    batch = synth.synth_batch(2, 4096, ragged=False, seed=321)
    assert batch["npoints_in_batch"] == [4096, 4096]
    perms = [[3, 1, 0, 2], [0, 2, 1, 3], [1, 0, 3, 2], [2, 3, 1, 0], [0, 1, 2, 3]]
    with torch.no_grad():
        ref = Oracle({k: v.clone() for k, v in sd.items()}, lcfg.plain(cfg), training=False).forward(batch, perms)["xt"].numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    errs = {}
    for mode in ("fp32", "bf16"):
        ops.set_gemm_precision(mode)
        try:
            m = _build(cfg, sd, False)
            m.ptv3_model.order_perms = perms
            with torch.no_grad():
                m(_dev_batch(batch), compute_loss=True, compute_final_action=False)
            errs[mode] = float(np.abs(m.last_pred[0].cpu().numpy() - ref).max()) / scale
        finally:
            ops.set_gemm_precision("fp32")
    assert errs["fp32"] <= LOGIT_TOL and errs["bf16"] <= 3e-2, errs


def test_peract_config_bf16_backward_with_augmented_dense_clouds():
    """BASELINE configs[4] including BACKWARD: PerAct preset, two dense 4096-point clouds after the PerAct
    augmentation (aug_max_rot 45 + jitter -> duplicate voxels), train mode, reference-initialised weights, against the
    fp32 oracle under autograd.  The exact mode must reproduce the gradient to fp32 accuracy; the bf16 mode (operands
    rounded to 8 bits of mantissa, fp32 accumulate) is held to what was measured for it on MI355X: logits within 3e-2,
    the gradient as a whole within 25 % of the fp32 gradient in norm and within 0.97 in direction (measured: 2e-3 on the
    logits, 15 % / 0.989; the noise sits in the small gradients — bf16x3 on the same inputs: 4e-6, 0.003 %)."""
    from oracle.model import Oracle
    from robot_3dlotus_amd import config as lcfg, synth
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("peract")
    sd = seeded_state_dict(gu.state_template(cfg), 92, "init")
    batch = synth.augment_clouds(synth.synth_batch(2, 4096, ragged=False, seed=322), seed=9, max_rot_deg=45.0)
    perms = [[3, 1, 0, 2], [0, 2, 1, 3], [1, 0, 3, 2], [2, 3, 1, 0], [0, 1, 2, 3]]
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    out = Oracle(sdg, lcfg.plain(cfg), training=True).forward(batch, perms)
    out["losses"]["total"].backward()
    ref = out["xt"].detach().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    gref = torch.cat([sdg[n].grad.flatten() for n, _ in _build(cfg, sd, True).named_parameters()])
    for mode, logit_tol, gtol, cos_min in (("fp32", LOGIT_TOL, 2e-3, 0.99999), ("bf16", 3e-2, 0.25, 0.97)):
        m = _build(cfg, sd, True)
        m.gemm_precision = mode
        m.ptv3_model.order_perms = perms
        _, losses = m(_dev_batch(batch), compute_loss=True, compute_final_action=False)
        losses["total"].backward()
        assert m.ptv3_model.frontend is not None and float(np.abs(m.last_pred[0].detach().cpu().numpy() - ref).max()) <= logit_tol * scale
        assert abs(losses["total"].item() - out["losses"]["total"].item()) <= max(logit_tol, 1e-4) * max(1.0, abs(out["losses"]["total"].item()))
        g = torch.cat([p.grad.flatten() for p in m.parameters()]).cpu()
        assert torch.isfinite(g).all()
        rel = float((g - gref).norm() / gref.norm())
        cos = float(torch.dot(g.double(), gref.double()) / (g.double().norm() * gref.double().norm()))
        assert rel < gtol and cos > cos_min, (mode, rel, cos)


def test_peract_config_bf16_storage_forward_backward():
    """BASELINE configs[4] as SURVEY 8(d) reads it: bf16 ACTIVATION STORAGE (every [N, C] tensor in HBM is bf16: the
    lotus_b16_* twins of the C-ABI), fp32 master weights, fp32 parameter gradients, fp32 accumulation.  PerAct preset, two
    dense 4096-point clouds after the PerAct augmentation (duplicate voxels), train mode, against the fp32 oracle under
    autograd.  Bounds are bf16-sized and were measured on MI355X (see profiles/r03_parity.json): logits relative to the
    largest logit, the whole gradient in norm and direction.  The fp32 model on the same inputs stays at fp32 accuracy, and
    a forward pass of an fp32 model after the bf16 one is bit-identical to one before it (no state leaks between the two
    storage modes)."""
    from oracle.model import Oracle
    from robot_3dlotus_amd import config as lcfg, synth
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("peract")
    sd = seeded_state_dict(gu.state_template(cfg), 93, "init")
    batch = synth.augment_clouds(synth.synth_batch(2, 4096, ragged=False, seed=323), seed=10, max_rot_deg=45.0)
    perms = [[3, 1, 0, 2], [0, 2, 1, 3], [1, 0, 3, 2], [2, 3, 1, 0], [0, 1, 2, 3]]
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    out = Oracle(sdg, lcfg.plain(cfg), training=True).forward(batch, perms)
    out["losses"]["total"].backward()
    ref = out["xt"].detach().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    names = [n for n, _ in _build(cfg, sd, True).named_parameters()]
    gref = torch.cat([sdg[n].grad.flatten() for n in names])

    def run(storage):
        m = _build(cfg, sd, True)
        m.act_storage = storage
        m.ptv3_model.order_perms = perms
        _, losses = m(_dev_batch(batch), compute_loss=True, compute_final_action=False)
        losses["total"].backward()
        return m, losses

    m32, l32 = run(None)
    xt32 = m32.last_pred[0].detach().clone()
    mb, lb = run("bf16")
    assert mb.last_pred[0].dtype == torch.bfloat16 and all(p.grad.dtype == torch.float32 for p in mb.parameters())
    err = float(np.abs(mb.last_pred[0].detach().float().cpu().numpy() - ref).max()) / scale
    g = torch.cat([p.grad.flatten() for p in mb.parameters()]).cpu()
    assert torch.isfinite(g).all()
    rel = float((g - gref).norm() / gref.norm())
    cos = float(torch.dot(g.double(), gref.double()) / (g.double().norm() * gref.double().norm()))
    lerr = abs(lb["total"].item() - out["losses"]["total"].item()) / max(1.0, abs(out["losses"]["total"].item()))
    ledger.record("peract_bf16_storage/2x4096_augmented", logit_rel_err=err, loss_rel_err=lerr, grad_rel_err=rel, grad_cosine=cos,
                  logit_max=scale)
    assert err <= 3e-2 and lerr <= 3e-2 and rel < 0.35 and cos > 0.95, (err, lerr, rel, cos)
    m32b, _ = run(None)
    assert torch.equal(m32b.last_pred[0], xt32), "fp32 forward changed after a bf16-storage pass"
    assert all(torch.equal(a.grad, b.grad) for a, b in zip(m32.parameters(), m32b.parameters()))
    assert float(np.abs(xt32.cpu().numpy() - ref).max()) <= LOGIT_TOL * scale


def test_drop_path_rows_and_backward():
    """DropPath (timm semantics, model.py:655-657): a stage of depth 2 with drop_path > 0 in train mode.  (1) the op: every row
    of x + DropPath(branch) is either x (dropped) or x + branch / (1 - p), the keep rate is 1 - p, the backward map uses the
    same rows; (2) the model: train-mode forward / backward run, differ from the drop_path = 0 run, repeat bit-identically for
    the same step seed, and eval mode ignores drop_path (bit-identical to a drop_path = 0 model)."""
    from robot_3dlotus_amd import config as lcfg, ops, synth
    from weights_util import seeded_state_dict

    g = torch.Generator(device="cuda").manual_seed(0)
    x, br = torch.randn(20000, 64, device="cuda", generator=g), torch.randn(20000, 64, device="cuda", generator=g)
    p = 0.3
    y = ops.drop_path(br, x, p, 1234)
    pq = int(p * 65536) / 65536   # the probability the 16-bit threshold applies; kept rows are scaled by 1 / (1 - pq) (round 6)
    kept = (y - x).abs().sum(1) > 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.02
    assert torch.allclose(y[kept], x[kept] + br[kept] / (1 - pq), rtol=1e-6, atol=1e-6) and torch.equal(y[~kept], x[~kept])
    dy = ops.drop_path(br, None, p, 1234)
    assert torch.equal(dy[~kept], torch.zeros_like(dy[~kept])) and torch.allclose(dy[kept], br[kept] / (1 - pq), rtol=1e-6, atol=1e-6)

    cfg = lcfg.preset("tinydeep")
    sd = seeded_state_dict(gu.state_template(cfg), 3, "scaled")
    batch = _dev_batch(synth.synth_batch(2, 600, ragged=True, seed=8))
    perms = [[1, 3, 0, 2], [2, 0, 3, 1]]

    def run(dp, train, seed_step=None):
        c = lcfg.preset("tinydeep")
        c.ptv3_config.drop_path = dp
        m = _build(c, sd, train)
        m.ptv3_model.order_perms = perms
        out = m(batch, compute_loss=True, compute_final_action=False)
        losses = out[1]
        if train:
            losses["total"].backward()
        return m, losses["total"].detach().clone()

    torch.manual_seed(5)
    m0, l0 = run(0.0, True)
    torch.manual_seed(5)
    m1, l1 = run(0.4, True)
    torch.manual_seed(5)
    m1b, l1b = run(0.4, True)
    assert torch.isfinite(l1) and float((l1 - l0).abs()) > 1e-4, "drop_path had no effect in train mode"
    assert torch.equal(l1, l1b) and all(torch.equal(a.grad, b.grad) for a, b in zip(m1.parameters(), m1b.parameters()))
    assert all(p_.grad is not None and torch.isfinite(p_.grad).all() for p_ in m1.parameters())
    _, e0 = run(0.0, False)
    _, e1 = run(0.4, False)
    assert torch.equal(e0, e1), "drop_path must be the identity in eval mode"


def test_full_size_train_step_properties():
    """BASELINE configs[1] size (16 x 4096, v1): forward+backward runs, everything finite, every
    parameter receives a gradient, eval-mode API returns f64[B, 8] like the reference."""
    from robot_3dlotus_amd import config as lcfg, synth
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA

    torch.manual_seed(0)
    m = SimplePolicyPTV3CA(lcfg.preset("v1")).cuda().train()
    batch = _dev_batch(synth.synth_batch(16, 4096, seed=0))
    _, losses = m(batch, compute_loss=True, compute_final_action=False)
    losses["total"].backward()
    assert all(torch.isfinite(v).all() for v in losses.values())
    for n_, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n_
    m.eval()
    with torch.no_grad():
        acts = m(batch, compute_loss=False)
    assert acts.shape == (16, 8) and acts.dtype == torch.float64


def test_prefetch_is_bit_identical():
    """policy.prefetch() (front-end of the next batch on a side stream) only moves work in time: losses,
    logits and gradients must equal the synchronous path bit for bit, over several pipelined steps."""
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import config as lcfg, synth
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("tiny")
    sd = seeded_state_dict(gu.state_template(cfg), 3, "scaled")
    perms = [[2, 0, 3, 1], [1, 3, 0, 2]]
    batches = [_dev_batch(synth.synth_batch(3, 700, ragged=True, seed=s)) for s in (5, 6, 7)]

    def run(prefetch):
        m = _build(cfg, sd, True)
        m.ptv3_model.order_perms = perms
        out = []
        if prefetch:
            m.prefetch(batches[0])
        for i, b in enumerate(batches):
            m.zero_grad(set_to_none=True)
            _, losses = m(b, compute_loss=True, compute_final_action=False)
            if prefetch and i + 1 < len(batches):
                m.prefetch(batches[i + 1])
            losses["total"].backward()
            out.append((losses["total"].detach().clone(), m.last_pred[0].detach().clone(),
                        [p.grad.detach().clone() for p in m.parameters()]))
        torch.cuda.synchronize()
        return out

    a, b = run(False), run(True)
    for (la, xa, ga), (lb, xb, gb) in zip(a, b):
        assert torch.equal(la, lb) and torch.equal(xa, xb)
        for u, v in zip(ga, gb):
            assert torch.equal(u, v)


def test_deferred_wgrad_join_is_bit_identical():
    """ops.set_wgrad_join("end") (one join of the weight-gradient stream per backward pass instead of one per
    autograd node) and running without the weight-gradient stream at all must not change a single bit of the
    gradients."""
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import config as lcfg, ops, synth
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("tiny")
    sd = seeded_state_dict(gu.state_template(cfg), 4, "scaled")
    batches = [_dev_batch(synth.synth_batch(4, 900, ragged=True, seed=s)) for s in (11, 12, 13)]

    def run(mode):
        ops.set_wgrad_join("node" if mode == "single" else mode)
        ops.enable_side_stream(mode != "single")
        try:
            m = _build(cfg, sd, True)
            m.ptv3_model.order_perms = [[0, 1, 2, 3], [3, 2, 1, 0]]
            out = []
            for b in batches:
                for p in m.parameters():
                    p.grad = None
                _, losses = m(b, compute_loss=True, compute_final_action=False)
                losses["total"].backward()
                out.append([p.grad.detach().clone() for p in m.parameters()])
            torch.cuda.synchronize()
            return out
        finally:
            ops.set_wgrad_join("node")
            ops.enable_side_stream(True)

    a, b, c = run("node"), run("end"), run("single")  # "single": LOTUS_SIDE_STREAM=0, everything on one stream
    for ga, gb, gc in zip(a, b, c):
        for u, v, w in zip(ga, gb, gc):
            assert torch.equal(u, v) and torch.equal(u, w)


def test_full_size_stream_modes_bit_identical():
    """The bench configuration (v1, 16 x 4096, high-priority step stream, prefetched front-end) over several steps with
    the weight gradients (a) on their own stream joined once per backward pass — ordered by completion events bound
    to the producing launches, created without the system-scope fence — and (b) on the step stream itself: a missing
    ordering or a stale cache line shows up as a differing gradient bit.  All 460 gradients of every step must be
    equal."""
    from robot_3dlotus_amd import config as lcfg, ops, synth
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA

    batches = [_dev_batch(synth.synth_batch(16, 4096, seed=s)) for s in (0, 1)]
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in SimplePolicyPTV3CA(lcfg.preset("v1")).state_dict().items()}

    def run(side):
        ops.enable_side_stream(side)
        ops.set_wgrad_join("end" if side else "node")
        hi = torch.cuda.Stream(priority=-1)
        try:
            torch.manual_seed(1)
            m = SimplePolicyPTV3CA(lcfg.preset("v1"))
            m.load_state_dict(sd)
            m = m.cuda().train()
            sums = []
            with torch.cuda.stream(hi):
                m.prefetch(batches[0])
                for i in range(6):
                    for p in m.parameters():
                        p.grad = None
                    _, losses = m(batches[i % 2], compute_loss=True, compute_final_action=False)
                    m.prefetch(batches[(i + 1) % 2])
                    losses["total"].backward()
                    hi.synchronize()
                    sums.append([p.grad.clone() for p in m.parameters()])
            return sums
        finally:
            ops.set_wgrad_join("node")
            ops.enable_side_stream(True)

    a, b = run(True), run(False)
    for step, (ga, gb) in enumerate(zip(a, b)):
        for k, (u, v) in enumerate(zip(ga, gb)):
            assert torch.equal(u, v), (step, k)


def test_models_of_different_precisions_coexist():
    """Operand precision is per call (captured per autograd node): an fp32 model and a bf16 model interleaved in one
    process — forward of one, forward of the other, then both backward passes — give bit-identically the results of
    running each alone (the old process-wide knob made this impossible)."""
    from robot_3dlotus_amd import config as lcfg, synth
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("tiny")
    sd = seeded_state_dict(gu.state_template(cfg), 8, "scaled")
    batch = synth.synth_batch(2, 600, ragged=True, seed=77)
    perms = [[0, 2, 1, 3], [3, 1, 0, 2]]

    def alone(mode):
        m = _build(cfg, sd, True)
        m.gemm_precision, m.ptv3_model.order_perms = mode, perms
        _, l = m(_dev_batch(batch), compute_loss=True, compute_final_action=False)
        l["total"].backward()
        return m.last_pred[0].detach().clone(), torch.cat([p.grad.flatten() for p in m.parameters()]).clone()

    xa, ga = alone("fp32")
    xb, gb = alone("bf16")
    assert not torch.equal(xa, xb)
    ma, mb = _build(cfg, sd, True), _build(cfg, sd, True)
    ma.gemm_precision, mb.gemm_precision = "fp32", "bf16"
    ma.ptv3_model.order_perms = mb.ptv3_model.order_perms = perms
    _, la = ma(_dev_batch(batch), compute_loss=True, compute_final_action=False)
    _, lb = mb(_dev_batch(batch), compute_loss=True, compute_final_action=False)
    lb["total"].backward()
    la["total"].backward()
    assert torch.equal(ma.last_pred[0], xa) and torch.equal(mb.last_pred[0], xb)
    assert torch.equal(torch.cat([p.grad.flatten() for p in ma.parameters()]), ga)
    assert torch.equal(torch.cat([p.grad.flatten() for p in mb.parameters()]), gb)

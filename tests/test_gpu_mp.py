"""GPU parity of the motion-planner drop-in (BASELINE configs[3], 3D-LOTUS++) against the fixtures captured from the
imported MotionPlannerPTV3CA and against the oracle run live.  Same tolerance rule as test_gpu_model.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import golden_util as gu  # noqa: E402

LOGIT_TOL = 1e-4


def _build(cfg, sd, train):
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd.policy import MODEL_FACTORY

    m = MODEL_FACTORY["MotionPlannerPTV3CA"](cfg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    m.train(train)
    m.ptv3_model.proj_drop = m.ptv3_model.attn_drop = 0.0      # fixtures are dropout-free
    m.act_proj_head.dropout = 0.0
    return m


def _dev_batch(batch):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else
                ([t.cuda() for t in v] if k == "gt_trajs_disc_pos_probs" else v)) for k, v in batch.items()}


@pytest.mark.parametrize("case", gu.MP_CASES)
def test_mp_golden_fixture_parity(case):
    fx, cfg, batch, sd = gu.load_case_mp(case)
    train = bool(fx["meta_train"])
    m = _build(cfg, sd, train)
    m.ptv3_model.order_perms = [p.tolist() for p in fx["perms"]]
    _, losses = m(_dev_batch(batch), compute_loss=True, compute_final_action=False, decode_actions=True)
    _, xr, xo, xs = m.last_pred
    for name, got in (("xt", m.pred_pos()), ("xr", xr), ("xo", xo), ("xstop", xs)):
        ref = fx[name]
        tol = LOGIT_TOL * max(1.0, float(np.abs(ref).max()))
        err = float(np.abs(got.detach().cpu().numpy() - ref).max())
        assert err <= tol, f"{case} {name}: max |diff| {err:.3e} > {tol:.3e}"
    assert set(losses) == {"pos", "rot", "open", "stop", "total"}
    for k in losses:
        ref = float(fx["loss_" + k])
        assert abs(losses[k].item() - ref) <= 1e-4 * max(1.0, abs(ref)), (k, losses[k].item(), ref)
    losses["total"].backward()
    gmax = max(float(fx[k]) for k in fx if k.startswith("gnorm/"))
    nograd = set(fx["nograd"].tolist())
    worst = (0.0, None)
    for name, p in m.named_parameters():
        if name in nograd:                       # txt_attn_fc: the reference leaves it without gradient too
            assert p.grad is None, name
            continue
        assert p.grad is not None, f"no gradient for {name}"
        ref = float(fx["gnorm/" + name])
        got = p.grad.double().norm().item()
        worst = max(worst, (abs(got - ref) / (ref + 1e-3 * gmax), name))
        head = fx["ghead/" + name]
        np.testing.assert_allclose(p.grad.flatten()[:48].cpu().numpy(), head,
                                   atol=2e-3 * float(np.abs(head).max()) + 2e-5 * gmax, rtol=0, err_msg=name)
    assert worst[0] < 2e-3, f"gradient norm mismatch {worst}"
    if train:
        sdn = m.state_dict()
        for k in fx:
            if k.startswith("buf/"):
                np.testing.assert_allclose(sdn[k[4:]].cpu().numpy(), fx[k], atol=2e-3, rtol=2e-3, err_msg=k)


def test_mp_live_oracle_parity_decode_and_determinism():
    """Fresh seeded inputs: HIP motion planner vs the oracle on the host (logits, losses), the decoded trajectory
    against the oracle's get_best_pos_from_disc_pos restatement, and a bit-identical repeat."""
    from oracle.labels import best_pos_max
    from oracle.model import Oracle
    from robot_3dlotus_amd import config as lcfg, synth
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("mp_tiny")
    sd = seeded_state_dict(gu.state_template(cfg), 41, "scaled")
    batch = synth.synth_batch_mp(3, 640, ragged=True, seed=77)
    perms = [[2, 0, 3, 1], [1, 3, 0, 2]]
    with torch.no_grad():
        ref = Oracle(sd, lcfg.plain(cfg), training=False).forward_mp(batch, perms)
    m = _build(cfg, sd, False)
    m.ptv3_model.order_perms = perms
    with torch.no_grad():
        final, losses = m(_dev_batch(batch), compute_loss=True)
    for name, got in (("xt", m.pred_pos()), ("xr", m.last_pred[1]), ("xo", m.last_pred[2]), ("xstop", m.last_pred[3])):
        r = ref[name].numpy()
        assert float(np.abs(got.cpu().numpy() - r).max()) <= LOGIT_TOL * max(1.0, float(np.abs(r).max())), name
    for k, v in ref["losses"].items():
        assert abs(losses[k].item() - v.item()) <= 1e-4 * max(1.0, abs(v.item())), k
    B, T = 3, cfg.action_config.max_traj_len
    assert final.shape == (B, T, 3 + 4 + 2)
    # decode: arg-max bin per axis of the HIP logits, through the oracle's restatement (float64 -> .float())
    xt = m.pred_pos().cpu()
    counts = batch["npoints_in_batch"]
    pcs = torch.split(batch["pc_fts"], counts)
    for b, lg in enumerate(torch.split(xt, counts, dim=2)):
        for t in range(T):
            prob = torch.softmax(lg[t].reshape(3, -1), -1).numpy()
            want = best_pos_max(prob, pcs[b][:, :3].numpy(), cfg.action_config.pos_bin_size, cfg.action_config.pos_bins)
            np.testing.assert_array_equal(final[b, t, :3].cpu().numpy(), want.astype(np.float32))
    with torch.no_grad():
        final2, losses2 = m(_dev_batch(batch), compute_loss=True)
    assert torch.equal(final, final2) and all(torch.equal(losses[k], losses2[k]) for k in losses)


def test_mp_bf16_activation_storage_forward_backward():
    """bf16 ACTIVATION STORAGE for the motion planner (model.act_storage = 'bf16': the lotus_b16_* twins, fp32 master weights,
    fp32 parameter gradients), as tests/test_gpu_model.py checks it for the policy: against the fp32 oracle under autograd on
    seeded inputs with scaled weights (train mode, dropout off) — logits relative to the largest logit of each head, the
    whole gradient in norm and direction, bf16-sized bars — and an fp32 pass after the bf16 one is bit-identical to one
    before it."""
    from oracle.model import Oracle
    from robot_3dlotus_amd import config as lcfg, synth
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("mp_tiny")
    sd = seeded_state_dict(gu.state_template(cfg), 43, "scaled")
    batch = synth.synth_batch_mp(3, 640, ragged=True, seed=79)
    perms = [[2, 0, 3, 1], [1, 3, 0, 2]]
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    ref = Oracle(sdg, lcfg.plain(cfg), training=True).forward_mp(batch, perms)
    ref["losses"]["total"].backward()

    def run(storage):
        m = _build(cfg, sd, True)
        m.act_storage = storage
        m.ptv3_model.order_perms = perms
        _, losses = m(_dev_batch(batch), compute_loss=True, compute_final_action=False)
        losses["total"].backward()
        return m, losses

    m32, l32 = run(None)
    x32 = m32.pred_pos().detach().clone()
    mb, lb = run("bf16")
    assert mb.last_pred[0][0].dtype == torch.bfloat16 and all(p.grad is None or p.grad.dtype == torch.float32 for p in mb.parameters())
    for name, got in (("xt", mb.pred_pos()), ("xr", mb.last_pred[1]), ("xo", mb.last_pred[2]), ("xstop", mb.last_pred[3])):
        r = ref[name].detach().numpy()
        err = float(np.abs(got.detach().float().cpu().numpy() - r).max()) / max(1.0, float(np.abs(r).max()))
        assert err <= 3e-2, (name, err)
    names = [n for n, p in mb.named_parameters() if p.grad is not None]
    g = torch.cat([dict(mb.named_parameters())[n].grad.flatten() for n in names]).cpu()
    gref = torch.cat([sdg[n].grad.flatten() for n in names])
    assert torch.isfinite(g).all()
    rel = float((g - gref).norm() / gref.norm())
    cos = float(torch.dot(g.double(), gref.double()) / (g.double().norm() * gref.double().norm()))
    lerr = abs(lb["total"].item() - ref["losses"]["total"].item()) / max(1.0, abs(ref["losses"]["total"].item()))
    assert lerr <= 3e-2 and rel < 0.35 and cos > 0.95, (lerr, rel, cos)
    m32b, _ = run(None)
    assert torch.equal(m32b.pred_pos(), x32), "fp32 forward changed after a bf16-storage pass"
    assert all((a.grad is None and b.grad is None) or torch.equal(a.grad, b.grad) for a, b in zip(m32.parameters(), m32b.parameters()))


def test_mp_train_step_is_deterministic():
    """Two identical training steps from the same state give bit-identical gradients (no atomics on the gradient path:
    the label-embedding gradient is a GEMM reduction)."""
    from robot_3dlotus_amd import config as lcfg, synth
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("mp_tiny")
    sd = seeded_state_dict(gu.state_template(cfg), 43, "scaled")
    batch = _dev_batch(synth.synth_batch_mp(2, 512, ragged=True, seed=78))
    grads = []
    for _ in range(2):
        m = _build(cfg, sd, True)
        m.ptv3_model.order_perms = [[0, 1, 2, 3], [3, 2, 1, 0]]
        _, losses = m(dict(batch), compute_loss=True, compute_final_action=False)
        losses["total"].backward()
        torch.cuda.synchronize()
        grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys()
    for n in grads[0]:
        assert torch.equal(grads[0][n], grads[1][n]), n


def test_mp_prefetch_is_bit_identical():
    """MotionPlannerPTV3CA.prefetch() (network input + integer front-end of the next batch on the side stream) only moves
    work in time: losses and gradients equal the synchronous path bit for bit over pipelined steps, and the prefetched
    front-end is actually consumed (no second build)."""
    from robot_3dlotus_amd import config as lcfg, synth
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("mp_tiny")
    sd = seeded_state_dict(gu.state_template(cfg), 44, "scaled")
    batches = [_dev_batch(synth.synth_batch_mp(2, 512, ragged=True, seed=s)) for s in (81, 82, 83)]

    def run(prefetch):
        m = _build(cfg, sd, True)
        m.ptv3_model.order_perms = [[0, 1, 2, 3], [3, 2, 1, 0]]
        out, consumed = [], 0
        if prefetch:
            m.prefetch(batches[0])
        for i, b in enumerate(batches):
            m.zero_grad(set_to_none=True)
            consumed += m.ptv3_model._pending is not None
            _, losses = m(b, compute_loss=True, compute_final_action=False)
            assert m.ptv3_model._pending is None
            if prefetch and i + 1 < len(batches):
                m.prefetch(batches[i + 1])
            losses["total"].backward()
            out.append((losses["total"].detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
        torch.cuda.synchronize()
        return out, consumed

    (a, ca), (b, cb) = run(False), run(True)
    assert ca == 0 and cb == len(batches)
    for (la, ga), (lb, gb) in zip(a, b):
        assert torch.equal(la, lb) and ga.keys() == gb.keys()
        for n in ga:
            assert torch.equal(ga[n], gb[n]), n


def test_traj_loss_fn_matches_torch_expression():
    """ops.TrajLossFn (one launch) against the reference's expression on the [B, T]-sized tensors
    (motion_planner_ptv3.py:327-397): the five losses and the gradients w.r.t. the logits and the heatmap cross
    entropies, ragged trajectory masks, non-unit loss weights."""
    import torch.nn.functional as F
    from robot_3dlotus_amd import ops

    torch.manual_seed(3)
    B, T, eb = 7, 5, 72
    dev = torch.device("cuda")
    ae = torch.randn(B, T, eb * 3 + 2, device=dev, requires_grad=True)
    ce = (torch.rand(B, T, 3, device=dev) * 5).requires_grad_(True)
    gt = torch.zeros(B, T, 7, device=dev)
    gt[..., :3] = torch.randn(B, T, 3, device=dev)
    gt[..., 3:6] = torch.randint(0, eb, (B, T, 3), device=dev).float()
    gt[..., 6] = torch.randint(0, 2, (B, T), device=dev).float()
    stop = torch.randint(0, 2, (B, T), device=dev).float()
    lens = torch.tensor([5, 1, 3, 2, 5, 4, 1], device=dev)
    m = (torch.arange(T, device=dev)[None] < lens[:, None]).float()
    pos_w, rot_w = 1.5, 0.7

    def ref(ae, ce):
        pred_rot = ae[..., :eb * 3].reshape(B, T, eb, 3)
        pred_open, pred_stop = ae[..., -2], ae[..., -1]
        msum = m.sum()
        pos = ((ce.sum(-1) * m).sum(1) / (3.0 * m.sum(1))).sum() / B
        rl = F.cross_entropy(pred_rot.permute(0, 1, 3, 2).reshape(-1, eb), gt[..., 3:-1].long().reshape(-1),
                             reduction="none").view(B, T, 3)
        rot = (rl * m.unsqueeze(-1)).sum() / msum / 3
        opn = (F.binary_cross_entropy_with_logits(pred_open, gt[..., -1], reduction="none") * m).sum() / msum
        stp = (F.binary_cross_entropy_with_logits(pred_stop, stop, reduction="none") * m).sum() / msum
        return torch.stack([pos, rot, opn, stp, pos_w * pos + rot_w * rot + opn + stp])

    w = torch.tensor([0.3, -1.1, 0.9, 2.0, 1.0], device=dev)   # upstream gradient of all five outputs
    Lr = ref(ae, ce)
    gr = torch.autograd.grad((Lr * w).sum(), [ae, ce])
    Lh = ops.TrajLossFn.apply(ae.reshape(B * T, -1), ce.reshape(B * T, 3), gt.reshape(B * T, -1).contiguous(),
                              stop.reshape(-1).contiguous(), m.contiguous(), eb, pos_w, rot_w)
    gh = torch.autograd.grad((Lh * w).sum(), [ae, ce])
    assert torch.allclose(Lh, Lr, rtol=2e-6, atol=1e-6), (Lh, Lr)
    for a, b in zip(gh, gr):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), (a - b).abs().max()

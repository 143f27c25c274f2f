"""Full-size parity of BASELINE configs[1] against the oracle: v1, 16 clouds x 4096 points, train mode (batch statistics in
every BatchNorm, dropout 0 because the reference's RNG streams cannot be reproduced), injected order permutations.

The oracle (oracle/model.py, torch-CPU under autograd) needs ~5-10 s for forward + backward at this size on the GPU
box's host cores.  It is evaluated in float64 (the yardstick) and in float32 (the reference's arithmetic, whose own distance to
the yardstick is recorded next to the HIP model's).  Compared: the action logits (absolute 1e-4 — the north-star bar — where |logit|max < 1, else relative to
the largest logit), the four losses, and EVERY parameter gradient as a whole tensor, ||dg|| / (||g|| + floor) (bars below).  This reaches
the shapes no committed fixture reaches: 65 536 points, 512 patches per order at level 0, the split-K dense products and the
tap-split convolutions of the deep levels.  A second case uses augmented clouds (rotation + jitter: 1-7 % duplicate voxels).

The measured errors go to the ledger (tests/ledger.py -> profiles/rNN_parity.json)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import golden_util as gu  # noqa: E402
import ledger  # noqa: E402

PERMS = [[1, 3, 0, 2], [0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1], [1, 0, 2, 3]]
# Gradients, per parameter, whole tensor: rel = ||g_hip - g_f64|| / (||g_f64|| + GRAD_FLOOR * max_p ||g_p||).
#
# What fp32 can deliver at this size is bounded by the max-pool arg-max: SerializedPooling routes the gradient of every
# (voxel, channel) to ONE child row, and a forward difference of one fp32 ulp between two near-tied children re-routes it —
# a discrete change that every parameter upstream in backward inherits coherently (with ~10^7 (voxel, channel) pairs a few
# near-ties per batch are certain).  The test measures that regime instead of assuming it away:
#   * `oracle32`: the same oracle evaluated in float32 (the reference's own arithmetic) against the float64 yardstick;
#   * `tie`: the float64 oracle re-run with its pooled projections perturbed by 1e-7 relative (a float32 rounding), against
#     itself.
# Observed over three rounds of runs (profiles/r03_parity.json has the last): whichever arithmetic happens to cross a tie
# shows 80-120 of the 421 gradients between 1e-4 and 3e-3 (HIP 6.5e-4 / float32 oracle 1.3e-3 / tie probe 1.3e-3 in one run;
# HIP 2.8e-3 / 1.7e-4 / 3e-6 in another after an unrelated kernel change moved the roundings), all others — and the MEDIAN,
# always — at 1e-6 ... 4e-6, while the logits agree to 4e-6 absolute.  Bars: median <= 2e-5 (what every gradient shows when no
# tie is crossed: 10x margin), maximum <= 5e-3 (the re-routing regime), and the number of gradients above 1e-4 is recorded
# next to the float32 oracle's.  Defects of the size these bars could hide (a dropped 0.1 % term) are caught where ties are
# rare: the fixtures and the live-oracle tests compare every whole gradient at 1e-4 (measured 5e-6 ... 1.4e-5) at 1-2 k points.
GRAD_TOL = 1e-4
GRAD_TOL_ROUTED = 2e-5  # every gradient once ALL discrete routing decisions are the yardstick's (measured: 5.0e-6 ... 7.2e-6)
GRAD_FLOOR = 1e-3
GRAD_MAX_TOL = 5e-3
GRAD_MEDIAN_TOL = 2e-5
LOGIT_TOL = 1e-4


def _dev_batch(batch):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if k == "disc_pos_probs" else v))
            for k, v in batch.items()}


def _run(variant, augment, seed, tag):
    import robot_3dlotus_amd  # noqa: F401
    from oracle.model import Oracle
    from robot_3dlotus_amd import config as lcfg, synth
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("v1")
    sd = seeded_state_dict(gu.state_template(cfg), seed, variant)
    batch = synth.synth_batch(16, 4096, seed=seed)
    if augment:
        batch = synth.augment_clouds(batch, seed=seed + 1)
    torch.set_num_threads(min(32, torch.get_num_threads() if torch.get_num_threads() > 1 else 16))

    def oracle(dt, tie_noise=0.0, record_arg=False):
        sdg_ = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for k, v in sdg_.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
        o = Oracle(sdg_, lcfg.plain(cfg), training=True, dtype=dt)
        o.record_arg = record_arg
        if tie_noise:  # perturb what the max-pool compares by a float32-rounding-sized relative amount
            gen, lin0 = torch.Generator().manual_seed(5), o.lin

            def lin(x, name):
                y = lin0(x, name)
                return y * (1 + tie_noise * torch.randn(y.shape, generator=gen, dtype=dt)) if name.endswith("down.proj") else y

            o.lin = lin
        out_ = o.forward(batch, PERMS)
        out_["losses"]["total"].backward()
        return out_, sdg_

    # the yardstick is the oracle evaluated in float64; the same oracle in float32 (the reference's arithmetic) is
    # measured against it too, which shows how much of a difference is fp32 summation order rather than a defect
    out, sdg = oracle(torch.float64, record_arg=True)
    out32, sdg32 = oracle(torch.float32)
    _, sdg_tie = oracle(torch.float64, tie_noise=1e-7)

    m = SimplePolicyPTV3CA(cfg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    m.ptv3_model.proj_drop = m.ptv3_model.attn_drop = 0.0
    m.act_proj_head.dropout = 0.0
    m.ptv3_model.order_perms = PERMS
    from robot_3dlotus_amd import ops
    ops.ARG_TAP = []
    try:
        _, losses = m(_dev_batch(batch), compute_loss=True, compute_final_action=False)
    finally:
        tap, ops.ARG_TAP = ops.ARG_TAP, None
    # the arg-max tables: the four SerializedPooling levels (model.py:760-765) in encoder order, then the head's cloud max
    # ... and, between them, the sign pattern of the head's LeakyReLU pre-activation: a third discrete routing decision
    ref_args = [("pool", a) for a in out["pool_arg"]] + [("leaky", out["leaky_pre"])] + [("cloud", out["cloud_arg"])]
    assert [k for k, _ in tap] == [k for k, _ in ref_args]
    argdiff = []
    for (kind, a), (_, r) in zip(tap, ref_args):
        if kind == "leaky":
            argdiff.append(dict(kind=kind, pairs=int(a.numel()), differ=int(((a.cpu() > 0) != (r > 0)).sum())))
            continue
        a = a.cpu().long()
        argdiff.append(dict(kind=kind, pairs=int(a.numel()), differ=int((a != r).sum())))
    if augment:
        assert m.ptv3_model.last_n_dup > 0, "the augmented clouds were meant to contain duplicate voxels"
    rec = {"n_dup": int(m.ptv3_model.last_n_dup), "points": int(sum(batch["npoints_in_batch"])), "weights": variant, "augmented": bool(augment),
           "argmax_tables_vs_f64_oracle": argdiff}
    fails = []
    for name, got, ref in (("xt", m.last_pred[0], out["xt"]), ("xr", m.last_pred[1], out["xr"]), ("xo", m.last_pred[2], out["xo"])):
        o32 = float((out32[name].detach().double() - ref.detach()).abs().max())
        ref = ref.detach().numpy()
        err = float(np.abs(got.detach().cpu().double().numpy() - ref).max())
        mag = float(np.abs(ref).max())
        rec["logit_abs_err_" + name], rec["logit_max_" + name], rec["oracle32_logit_abs_err_" + name] = err, mag, o32
        if err > LOGIT_TOL * max(1.0, mag):
            fails.append(f"{name}: max |diff| {err:.3e} (|logit|max {mag:.3g})")
    for k in ("pos", "rot", "open", "total"):
        ref = float(out["losses"][k].detach())
        err = abs(losses[k].item() - ref)
        rec["loss_abs_err_" + k] = err
        if err > 1e-4 * max(1.0, abs(ref)):
            fails.append(f"loss {k}: {losses[k].item()} vs {ref}")
    losses["total"].backward()
    gmax = max(float(v.grad.norm()) for v in sdg.values() if v.grad is not None)
    worst, worst32, n, rels, table = (0.0, None), (0.0, None), 0, [], []
    for name, p in m.named_parameters():
        r = sdg[name].grad
        assert p.grad is not None and r is not None, name
        err = float((p.grad.cpu().double() - r).norm())
        den = float(r.norm()) + GRAD_FLOOR * gmax
        rel, rel32 = err / den, float((sdg32[name].grad.double() - r).norm()) / den
        rel_tie = float((sdg_tie[name].grad - r).norm()) / den
        worst, worst32 = max(worst, (rel, name)), max(worst32, (rel32, name))
        rels.append(rel)
        n += 1
        table.append((rel, rel32, rel_tie, name, err, float(r.norm())))
    for rel, rel32, rel_tie, name, err, gn in table:
        if rel > GRAD_MAX_TOL:
            fails.append(f"grad {name}: ||dg|| {err:.3e} vs ||g|| {gn:.3e} (rel {rel:.2e}; oracle fp32 {rel32:.2e}, tie {rel_tie:.2e})")
    table.sort(reverse=True)
    rec["grad_worst5"] = [dict(name=t[3], hip=float("%.3g" % t[0]), oracle_fp32=float("%.3g" % t[1]), tie_1e_7=float("%.3g" % t[2]))
                          for t in table[:5]]
    rec["n_gradients_above_1e-4"] = dict(hip=sum(1 for t in table if t[0] > GRAD_TOL), oracle_fp32=sum(1 for t in table if t[1] > GRAD_TOL),
                                         tie_1e_7=sum(1 for t in table if t[2] > GRAD_TOL))
    rec["tie_grad_rel_err_max"] = max(t[2] for t in table)
    if float(np.median(rels)) > GRAD_MEDIAN_TOL:
        fails.append(f"median gradient error {float(np.median(rels)):.2e}")
    rec.update(n_gradients=n, grad_rel_err_max=worst[0], grad_rel_err_argmax=worst[1], grad_rel_err_median=float(np.median(rels)),
               oracle32_grad_rel_err_max=worst32[0], oracle32_grad_rel_err_argmax=worst32[1], grad_norm_max=gmax,
               grad_tol=GRAD_TOL, grad_floor=GRAD_FLOOR, grad_max_tol=GRAD_MAX_TOL, yardstick="oracle/model.py evaluated in float64")
    # ---- the proof: the same model, the float64 oracle's arg-max tables injected into backward (forward values stay the
    # kernels' own).  Every discrete routing decision is now the yardstick's, so what is left is summation noise and EVERY
    # gradient has to meet the 1e-4 bar — at the shapes only this test reaches (65 536 rows, split-K, tap-split).
    for p_ in m.parameters():
        p_.grad = None
    ops.ARG_INJECT = [(k, r.to(torch.float32 if k == "leaky" else torch.int32)) for k, r in ref_args]
    try:
        _, losses2 = m(_dev_batch(batch), compute_loss=True, compute_final_action=False)
        assert not ops.ARG_INJECT, "not every injected table was consumed"
    finally:
        ops.ARG_INJECT = None
    losses2["total"].backward()
    inj = []
    for name, p_ in m.named_parameters():
        r = sdg[name].grad
        inj.append((float((p_.grad.cpu().double() - r).norm()) / (float(r.norm()) + GRAD_FLOOR * gmax), name))
    inj.sort(reverse=True)
    rec["injected_argmax"] = dict(grad_rel_err_max=inj[0][0], grad_rel_err_argmax=inj[0][1], grad_rel_err_median=float(np.median([t[0] for t in inj])),
                                  n_gradients_above_1e_4=sum(1 for t in inj if t[0] > GRAD_TOL), bar=GRAD_TOL_ROUTED, worst5=[dict(name=t[1], rel=float("%.3g" % t[0])) for t in inj[:5]])
    # Round 6: with the arg-max tables alone the worst gradient was 7.6e-5 (act_proj_head.heatmap_mlp.0.weight, init weights; the
    # float32 oracle: 1.7e-4) — three of the 8.4 M LeakyReLU pre-activations of the head have the other sign in fp32, i.e. slope
    # 1 instead of 0.02.  With that sign pattern injected as well, every gradient of every case is within 7.2e-6.
    for rel, name in inj:
        if rel > GRAD_TOL_ROUTED:
            fails.append(f"grad {name} with the oracle's routing (arg-max tables + LeakyReLU signs) injected: rel {rel:.2e} > {GRAD_TOL_ROUTED}")
    ledger.record("fullsize_oracle/" + tag, **rec)
    assert not fails, "; ".join(fails[:8])


def test_fullsize_v1_against_oracle_init_weights():
    """Reference-initialisation weights: |logit|max < 1, so the 1e-4 bar is absolute."""
    _run("init", False, 11, "v1_16x4096_init")


def test_fullsize_v1_against_oracle_scaled_weights():
    """x3 weights, non-trivial affine / running statistics (SURVEY Trap 2): softmax, qk-norm, GELU and BN leave their
    linear regime."""
    _run("scaled", False, 12, "v1_16x4096_scaled")


def test_fullsize_v1_against_oracle_duplicate_voxels():
    _run("scaled", True, 13, "v1_16x4096_scaled_augmented")

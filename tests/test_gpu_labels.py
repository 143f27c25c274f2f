"""GPU parity of the device-side soft position targets / arg-max decode against the known-answer
cases captured from the imported reference and, on ragged multi-cloud batches, against the numpy oracle.
'plain' targets and decoded positions must be bit-equal; 'dist' targets (float64 normalisation in the reference,
different summation order here) within one float32 ulp (1.2e-7 relative)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "labels_cases.npz")


def _ops():
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import ops
    return ops


def _run(ops, xyzs, gts, bins, kind, robots=None):
    B = len(xyzs)
    counts = [len(x) for x in xyzs]
    pc = torch.from_numpy(np.concatenate([np.concatenate([x, np.zeros((len(x), 4), np.float32)], 1) for x in xyzs], 0)).cuda()
    off = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32).cuda()
    bidx = torch.repeat_interleave(torch.arange(B, dtype=torch.int32), torch.tensor(counts)).cuda()
    gt = torch.from_numpy(np.concatenate([np.stack(gts, 0), np.zeros((B, 4), np.float32)], 1)).cuda()
    rm = None
    if robots is not None:
        m = np.zeros(sum(counts), np.uint8)
        for b, r in enumerate(robots):
            m[off[b].item() + np.asarray(r, dtype=np.int64)] = 1
        rm = torch.from_numpy(m).cuda()
    tgt = ops.pos_targets(pc, off, bidx, gt, 2 * bins, 0.01, kind, rm).cpu().numpy()
    out, o = [], 0
    for n in counts:
        out.append(tgt[o:o + 3 * n * 2 * bins].reshape(3, -1))
        o += 3 * n * 2 * bins
    return pc, off, out


def test_reference_known_answer_cases():
    ops = _ops()
    fx = np.load(GOLD)
    for k in range(int(fx["ncases"])):
        kind, bins = str(fx[f"kind{k}"]), int(fx[f"bins{k}"])
        xyz, gt, robot = fx[f"xyz{k}"], fx[f"gt{k}"], fx[f"robot{k}"]
        pc, off, got = _run(ops, [xyz], [gt], bins, kind, [robot] if len(robot) else None)
        ref = fx[f"prob{k}"]
        if kind == "plain":
            assert np.array_equal(got[0], ref), (k, np.abs(got[0] - ref).max())
        else:
            assert np.abs(got[0] - ref).max() <= 1.2e-7 * ref.max(), (k, np.abs(got[0] - ref).max())
        lg = fx[f"logits{k}"]  # (3, n * nb) -> xt layout [n][3][nb]
        n = len(xyz)
        xt = torch.from_numpy(np.ascontiguousarray(lg.reshape(3, n, 2 * bins).transpose(1, 0, 2)).reshape(n, -1)).cuda()
        best = ops.pos_decode_max(xt, pc, off, 1, 2 * bins, 0.01).cpu().numpy()[0]
        assert np.array_equal(best, fx[f"best{k}"]), (k, best, fx[f"best{k}"])


@pytest.mark.parametrize("kind", ["plain", "dist"])
def test_ragged_batch_matches_oracle(kind):
    ops = _ops()
    from oracle import labels as ol
    from robot_3dlotus_amd import synth

    batch = synth.synth_batch(5, 900, ragged=True, seed=9)
    pcs = np.split(batch["pc_fts"].numpy()[:, :3], np.cumsum(batch["npoints_in_batch"])[:-1])
    rng = np.random.default_rng(1)
    gts = [(p[rng.integers(len(p))] + rng.uniform(-0.02, 0.02, 3) + (3.0 if b == 2 else 0.0)).astype(np.float32) for b, p in enumerate(pcs)]
    robots = [np.sort(rng.choice(len(p), len(p) // 7, replace=False)) for p in pcs]
    pc, off, got = _run(ops, pcs, gts, 15, kind, robots)
    for b, p in enumerate(pcs):
        ref = ol.disc_gt_pos_prob(p, gts[b], 0.01, 15, kind, robots[b])
        if kind == "plain":
            assert np.array_equal(got[b], ref), b
        else:
            assert np.abs(got[b] - ref).max() <= 1.2e-7 * ref.max(), b


def test_policy_uses_device_labels_and_decode():
    """End to end through the drop-in policy: (i) a batch WITHOUT disc_pos_probs gives the same losses as the same batch
    with host-made labels from the oracle; (ii) the decoded positions of forward(compute_final_action=True) equal the
    oracle's arg-max decode of the returned logits."""
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import config as lcfg, synth
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
    from oracle import labels as ol

    torch.manual_seed(0)
    cfg = lcfg.preset("tiny")
    m = SimplePolicyPTV3CA(cfg).cuda().eval()
    m.ptv3_model.order_perms = [[0, 1, 2, 3], [1, 0, 3, 2]]
    bins, bs = int(cfg.action_config.pos_bins), float(cfg.action_config.pos_bin_size)
    batch = synth.synth_batch(3, 500, ragged=True, seed=4, pos_bins=bins)
    pcs = np.split(batch["pc_fts"].numpy()[:, :3], np.cumsum(batch["npoints_in_batch"])[:-1])
    gts = batch["gt_actions"].numpy()[:, :3]
    host = [torch.from_numpy(ol.disc_gt_pos_prob(p, gts[b], bs, bins, "plain")) for b, p in enumerate(pcs)]

    def dev(d):
        return {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if k == "disc_pos_probs" else v)) for k, v in d.items()}

    with torch.no_grad():
        b1 = dev({**batch, "disc_pos_probs": host})
        _, l1 = m(b1, compute_loss=True, compute_final_action=False)
        b2 = dev({k: v for k, v in batch.items() if k != "disc_pos_probs"})
        _, l2 = m(b2, compute_loss=True, compute_final_action=False)
        assert torch.equal(l1["pos"], l2["pos"]) and torch.equal(l1["total"], l2["total"])
        final = m(dev(batch), compute_loss=False)
    assert final.shape == (3, 8) and final.dtype == torch.float64
    xt = m.last_pred[0].cpu().numpy()  # (3, N, nb)
    o = 0
    for b, p in enumerate(pcs):
        lg = xt[:, o:o + len(p)].reshape(3, -1)
        assert np.array_equal(final[b, :3].cpu().numpy(), ol.best_pos_max(lg, p, bs, bins)), b
        o += len(p)


def test_policy_decodes_with_best_disc_pos_ens1():
    """forward(compute_final_action=True) with action_config.best_disc_pos = 'ens1' (the evaluation scripts' option,
    eval_simple_policy.py:63,83): the decoded positions equal the oracle's loop-for-loop restatement of
    get_best_pos_from_disc_pos(best='ens1') applied to the softmax of the model's own logits."""
    import golden_util as gu
    from oracle import labels as ol
    from robot_3dlotus_amd import config as lcfg, synth
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
    from weights_util import seeded_state_dict

    cfg = lcfg.preset("tiny")
    cfg.action_config.best_disc_pos = "ens1"
    sd = seeded_state_dict(gu.state_template(cfg), 4, "scaled")
    m = SimplePolicyPTV3CA(cfg)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    batch = synth.synth_batch(3, 300, ragged=True, seed=9)
    dev = {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if k == "disc_pos_probs" else v))
           for k, v in batch.items()}
    with torch.no_grad():
        final = m(dev, compute_loss=False, compute_final_action=True)
    bins, bs = cfg.action_config.pos_bins, cfg.action_config.pos_bin_size
    xt = m.last_pred[0].float().cpu()                     # (3, N, 2 * bins)
    counts = batch["npoints_in_batch"]
    pcs = torch.split(batch["pc_fts"], counts)
    for b, lg in enumerate(torch.split(xt, counts, dim=1)):
        prob = torch.softmax(lg.reshape(3, -1), -1).numpy()
        want = ol.best_pos_ens1(prob, pcs[b][:, :3].numpy(), bs, bins)
        assert np.array_equal(final[b, :3].cpu().numpy(), want), (b, final[b, :3], want)

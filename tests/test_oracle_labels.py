"""oracle/labels.py against the known-answer cases captured from the imported reference
(genrobo3d/utils/action_position_utils.py; tests/golden/make_golden_labels.py)."""
import os

import numpy as np

from oracle import labels as ol

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "labels_cases.npz")


def test_soft_targets_and_argmax_decode_match_reference_cases():
    fx = np.load(GOLD)
    for k in range(int(fx["ncases"])):
        kind, bins = str(fx[f"kind{k}"]), int(fx[f"bins{k}"])
        xyz, gt, robot = fx[f"xyz{k}"], fx[f"gt{k}"], fx[f"robot{k}"]
        prob = ol.disc_gt_pos_prob(xyz, gt, 0.01, bins, kind, robot if len(robot) else None)
        ref = fx[f"prob{k}"]
        assert prob.dtype == ref.dtype and prob.shape == ref.shape
        assert np.array_equal(prob, ref), (k, kind, np.abs(prob - ref).max())
        assert np.allclose(prob.sum(-1), 1.0, atol=1e-6)
        best = ol.best_pos_max(fx[f"logits{k}"], xyz, 0.01, bins)
        assert np.array_equal(best, fx[f"best{k}"]), k
        ens = ol.best_pos_ens1(fx[f"softmax{k}"], xyz, 0.01, bins)
        assert np.array_equal(ens, fx[f"ens1_{k}"]), (k, ens, fx[f"ens1_{k}"])


def test_product_ens1_decode_matches_reference_cases():
    """ops.pos_decode_ens1 (the product's vectorised host evaluation of best='ens1', an evaluation-time option:
    eval_simple_policy.py:63,83) on the logits of the reference cases — softmax inside, like simple_policy_ptv3.py:259-266 —
    against the answers captured from the imported reference."""
    import torch
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import ops

    fx = np.load(GOLD)
    xts, pcs, counts, want = [], [], [], []
    for k in range(int(fx["ncases"])):
        if int(fx[f"bins{k}"]) != 15:
            continue
        xyz, lg = fx[f"xyz{k}"], fx[f"logits{k}"]
        n = len(xyz)
        xts.append(torch.from_numpy(lg.reshape(3, n, 30).transpose(1, 0, 2).reshape(n, 90).copy()))
        pcs.append(torch.from_numpy(np.concatenate([xyz, np.zeros((n, 4), np.float32)], 1)))
        counts.append(n)
        want.append(fx[f"ens1_{k}"])
    got = ops.pos_decode_ens1(torch.cat(xts), torch.cat(pcs), counts, 30, 0.01).numpy()   # two clouds in one batch
    assert np.array_equal(got, np.stack(want))
    for k in range(int(fx["ncases"])):   # every case on its own (other bin counts)
        bins, xyz, lg = int(fx[f"bins{k}"]), fx[f"xyz{k}"], fx[f"logits{k}"]
        n = len(xyz)
        xt = torch.from_numpy(lg.reshape(3, n, 2 * bins).transpose(1, 0, 2).reshape(n, 6 * bins).copy())
        pc = torch.from_numpy(np.concatenate([xyz, np.zeros((n, 4), np.float32)], 1))
        assert np.array_equal(ops.pos_decode_ens1(xt, pc, [n], 2 * bins, 0.01).numpy()[0], fx[f"ens1_{k}"]), k

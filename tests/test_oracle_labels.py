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

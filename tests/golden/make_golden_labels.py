"""Known-answer cases for the soft position targets / arg-max decode from the IMPORTED reference (build container
only):  python tests/golden/make_golden_labels.py  ->  tests/golden/labels_cases.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


def main():
    from genrobo3d.utils.action_position_utils import get_disc_gt_pos_prob, get_best_pos_from_disc_pos

    rng = np.random.default_rng(0)
    out = {}
    cases = [("plain", 37, 15, False, 0.0), ("dist", 37, 15, False, 0.0), ("plain", 64, 4, True, 0.0),
             ("dist", 64, 4, True, 0.0), ("plain", 20, 3, False, 5.0), ("dist", 20, 3, True, 5.0)]
    for k, (kind, n, bins, robot, far) in enumerate(cases):
        xyz = (np.round(rng.uniform(-0.3, 0.3, size=(n, 3)) / 0.01) * 0.01 + rng.uniform(0.002, 0.008, size=(n, 3))).astype(np.float32)
        gt = (xyz[rng.integers(n)] + rng.uniform(-0.02, 0.02, size=3) + far).astype(np.float32)  # far: nothing within reach
        ridx = np.sort(rng.choice(n, size=n // 5, replace=False)) if robot else np.zeros(0, dtype=np.int64)
        prob = get_disc_gt_pos_prob(xyz, gt, pos_bin_size=0.01, pos_bins=bins, heatmap_type=kind,
                                    robot_point_idxs=ridx if robot else None)
        logits = rng.standard_normal((3, n * 2 * bins)).astype(np.float32)
        logits[0, 5] = logits[0, 11] = logits[0].max() + 1.0   # tie: the first index wins
        best = get_best_pos_from_disc_pos(logits, xyz, pos_bin_size=0.01, pos_bins=bins, best="max")
        # best='ens1' on the softmax of the logits (what the model hands over, simple_policy_ptv3.py:259-266)
        e = np.exp(logits - logits.max(-1, keepdims=True))
        sm = (e / e.sum(-1, keepdims=True)).astype(np.float32)
        ens = get_best_pos_from_disc_pos(sm, xyz, pos_bin_size=0.01, pos_bins=bins, best="ens1")
        out.update({f"kind{k}": np.array(kind), f"bins{k}": np.int64(bins), f"xyz{k}": xyz, f"gt{k}": gt, f"robot{k}": ridx,
                    f"prob{k}": prob, f"logits{k}": logits, f"best{k}": best, f"softmax{k}": sm, f"ens1_{k}": np.asarray(ens)})
    out["ncases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "labels_cases.npz"), **out)
    print("wrote labels_cases.npz", [out[f"prob{k}"].dtype for k in range(len(cases))])


if __name__ == "__main__":
    main()

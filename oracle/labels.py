"""ORACLE (test infrastructure, never shipped as product): soft position targets and arg-max position decoding.

numpy restatement of genrobo3d/utils/action_position_utils.py:
  * get_disc_gt_pos_prob (:7-46): per axis c and (point n, bin b) the candidate coordinate xyz[n,c] + (b - pos_bins) *
    pos_bin_size; 'plain' gives equal mass to every candidate closer than 0.01 to gt_pos[c], 'dist' gives mass
    1 / max(dist, 1e-4) to every candidate with dist <= 0.01; robot points are zeroed; an axis with no mass falls
    back to the single nearest candidate (first index on ties); each axis is normalised to sum 1.  Layout (3, n * 2 *
    pos_bins), index n * 2 * pos_bins + b.
  * get_best_pos_from_disc_pos(best='max') (:48-64): per axis the candidate coordinate of the first arg-max.
  * get_best_pos_from_disc_pos(best='ens1') (:66-85): per axis the probabilities of all candidates are summed per 5 mm cell
    (round(candidate / 0.005)), visiting the candidates in order of decreasing probability; the first cell (in that visiting
    order) with the strictly largest sum wins and the answer is cell * 0.005.
Arithmetic follows the reference: float64 candidates / distances (np.arange * float), 'plain' output float32, 'dist'
output float64.

Only `tests/` may import this.  Pinned by tests/golden/labels_cases.npz (made by importing the reference:
tests/golden/make_golden_labels.py)."""
import numpy as np


def candidates(xyz, pos_bin_size, pos_bins):
    shift = np.arange(-pos_bins, pos_bins) * pos_bin_size             # (2 * pos_bins,)  float64
    return (np.stack([shift] * 3, 0)[None, :, :] + xyz[:, :, None])   # (n, 3, 2 * pos_bins)


def disc_gt_pos_prob(xyz, gt_pos, pos_bin_size=0.01, pos_bins=50, heatmap_type="plain", robot_point_idxs=None):
    n = xyz.shape[0]
    cands = candidates(xyz, pos_bin_size, pos_bins)
    dists = np.abs(gt_pos[None, :, None] - cands).transpose(1, 0, 2).reshape(3, -1)  # 'n c b -> c (n b)'
    if heatmap_type == "plain":
        prob = np.zeros((3, n * pos_bins * 2), dtype=np.float32)
        prob[dists < 0.01] = 1
    else:
        prob = 1 / np.maximum(dists, 1e-4)
        prob[dists > 0.01] = 0
    if robot_point_idxs is not None and len(robot_point_idxs) > 0:
        prob = prob.reshape(3, n, -1)
        prob[:, robot_point_idxs] = 0
        prob = prob.reshape(3, -1)
    for i in range(3):
        if np.sum(prob[i]) == 0:
            prob[i, np.argmin(dists[i])] = 1
    return prob / np.sum(prob, -1, keepdims=True)


def best_pos_max(disc_pos_prob, xyz, pos_bin_size=0.01, pos_bins=50):
    cands = candidates(xyz, pos_bin_size, pos_bins).transpose(1, 0, 2).reshape(3, -1)
    idxs = np.argmax(disc_pos_prob, -1)
    return cands[np.arange(3), idxs]


def best_pos_ens1(disc_pos_prob, xyz, pos_bin_size=0.01, pos_bins=50):
    """get_best_pos_from_disc_pos(best='ens1'), utils/action_position_utils.py:66-85, loop for loop (dict insertion order,
    strict '>' when scanning the sums, accumulation in the probabilities' own dtype starting from int 0)."""
    import collections

    cands = candidates(xyz, pos_bin_size, pos_bins).transpose(1, 0, 2).reshape(3, -1)
    vox = np.round(cands / 0.005).astype(np.int32)
    idxs = np.argsort(-disc_pos_prob, -1)
    best = []
    for i in range(3):
        sums = collections.defaultdict(int)
        for k in idxs[i]:
            sums[vox[i, k].item()] += disc_pos_prob[i, k]
        bi, bv = None, -np.inf
        for k, v in sums.items():
            if v > bv:
                bv, bi = v, k
        best.append(bi * 0.005)
    return np.array(best)

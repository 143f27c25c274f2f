"""Host-side batch assembly for the hot path (the part that is testable offline).

`ptv3_collate_fn` / `ptv3_collate_fn_partial` produce the batch dictionaries the policies consume, from the per-item
dictionaries a dataset yields (every value a list with one entry per key step, as in the reference).  They mirror
genrobo3d/train/datasets/simple_policy_dataset.py:391-415 and motion_planner_dataset.py:360-410 key for key, with two
MI355X-minded additions that do not change any value:

* `pack=True` concatenates the soft position targets into ONE pinned buffer in the layout the cross-entropy kernel reads
  (`disc_pos_probs`: [sum_b 3*n_b*nb]; trajectories: [T, sum_b 3*n_b*nb]), so the 24 MB / step upload is a single
  asynchronous H2D copy instead of one per cloud (the list form is what the reference collate returns).
* `pin=True` puts the large tensors in pinned host memory (non_blocking uploads overlap with the previous step).

The LMDB / msgpack episode reader, augmentation and robot-box removal of the reference datasets are not built: no
episode data exists offline to pin them against.
"""
import numpy as np
import torch


def gen_seq_masks(seq_lens, max_len=None):
    """[N, L] bool, True inside the sequence (genrobo3d/train/datasets/common.py:23-40)."""
    seq_lens = np.array(seq_lens)
    if max_len is None:
        max_len = max(seq_lens)
    if max_len == 0:
        return np.zeros((len(seq_lens), 0), dtype=bool)
    batch_size = len(seq_lens)
    masks = np.arange(max_len).reshape(-1, max_len).repeat(batch_size, 0)
    return masks < seq_lens.reshape(-1, 1)


def _flatten(data):
    batch = {}
    for key in data[0].keys():
        batch[key] = sum([x[key] for x in data], [])
    return batch


def _pin(t, pin):
    return t.pin_memory() if pin and torch.cuda.is_available() else t


def _common(batch, pin):
    npts = [x.size(0) for x in batch["pc_fts"]]
    batch["npoints_in_batch"] = npts
    batch["offset"] = torch.cumsum(torch.LongTensor(npts), dim=0)
    batch["pc_fts"] = _pin(torch.cat(batch["pc_fts"], 0), pin)
    batch["txt_lens"] = [x.size(0) for x in batch["txt_embeds"]]
    batch["txt_embeds"] = _pin(torch.cat(batch["txt_embeds"], 0), pin)
    if len(batch.get("pc_centroids", [])) > 0:
        batch["pc_centroids"] = np.stack(batch["pc_centroids"], 0)


def ptv3_collate_fn(data, pack=False, pin=False):
    """3D-LOTUS policy batch (simple_policy_dataset.py:391-415)."""
    batch = _flatten(data)
    _common(batch, pin)
    for key in ("ee_poses", "gt_actions"):
        batch[key] = torch.stack(batch[key], 0)
    batch["step_ids"] = torch.LongTensor(batch["step_ids"])
    if pack and "disc_pos_probs" in batch:
        batch["disc_pos_probs"] = _pin(torch.cat([t.reshape(-1) for t in batch["disc_pos_probs"]]), pin)
    return batch


def ptv3_collate_fn_partial(max_traj_len, data, pack=False, pin=False):
    """3D-LOTUS++ motion-planner batch (motion_planner_dataset.py:360-410): trajectories shorter than `max_traj_len`
    repeat their last action / target and are masked out by `traj_masks`."""
    batch = _flatten(data)
    _common(batch, pin)
    batch["pc_labels"] = torch.cat(batch["pc_labels"], 0)
    for key in ("ee_poses", "gt_trajs_stop"):
        if key in batch:
            batch[key] = torch.stack(batch[key], 0)
    gt_trajs, traj_lens = [], []
    for traj in batch["gt_trajs"]:
        traj_lens.append(traj.size(0))
        if traj.size(0) < max_traj_len:
            gt_trajs.append(torch.cat([traj, traj[-1].repeat(max_traj_len - traj.size(0), 1)]))
        else:
            assert len(traj) == max_traj_len, len(traj)
            gt_trajs.append(traj)
    batch["gt_trajs"] = torch.stack(gt_trajs, 0)
    batch["traj_lens"] = traj_lens
    batch["traj_masks"] = torch.from_numpy(gen_seq_masks(traj_lens, max_len=max_traj_len)).bool()
    probs = []
    for traj in batch["gt_trajs_disc_pos_probs"]:
        if traj.size(0) < max_traj_len:
            probs.append(torch.cat([traj, traj[-1].repeat(max_traj_len - traj.size(0), 1, 1)]))
        else:
            assert len(traj) == max_traj_len, len(traj)
            probs.append(traj)
    batch["gt_trajs_disc_pos_probs"] = probs
    if pack:
        batch["gt_trajs_disc_pos_probs"] = _pin(torch.cat([t.reshape(max_traj_len, -1) for t in probs], 1), pin)
    return batch

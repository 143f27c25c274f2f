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
import itertools

import numpy as np
import torch


def gen_seq_masks(seq_lens, max_len=None):
    """Validity mask of padded sequences: bool [len(seq_lens), max_len], entry (i, j) is True iff j < seq_lens[i]
    (what genrobo3d/train/datasets/common.py:23-40 returns; a single broadcast comparison here)."""
    lens = np.asarray(seq_lens, dtype=np.int64).reshape(-1, 1)
    width = int(lens.max()) if max_len is None else int(max_len)
    return np.arange(width, dtype=np.int64)[None, :] < lens


def _pin(t, pin):
    return t.pin_memory() if pin and torch.cuda.is_available() else t


def _concat_items(items, pin):
    """Per-item lists (one entry per key step) -> one flat list per key, then the ragged point / token tensors are
    concatenated and described by counts + inclusive offsets (the Pointcept convention the backbone reads)."""
    batch = {key: list(itertools.chain.from_iterable(it[key] for it in items)) for key in items[0]}
    for key, counts_key in (("pc_fts", "npoints_in_batch"), ("txt_embeds", "txt_lens")):
        batch[counts_key] = [int(t.shape[0]) for t in batch[key]]
        batch[key] = _pin(torch.cat(batch[key], 0), pin)
    batch["offset"] = torch.tensor(np.cumsum(batch["npoints_in_batch"]), dtype=torch.long)
    if batch.get("pc_centroids"):
        batch["pc_centroids"] = np.stack(batch["pc_centroids"], 0)
    return batch


def ptv3_collate_fn(data, pack=False, pin=False):
    """3D-LOTUS policy batch (simple_policy_dataset.py:391-415)."""
    batch = _concat_items(data, pin)
    batch["ee_poses"] = torch.stack(batch["ee_poses"], 0)
    batch["gt_actions"] = torch.stack(batch["gt_actions"], 0)
    batch["step_ids"] = torch.tensor(batch["step_ids"], dtype=torch.long)
    if pack and "disc_pos_probs" in batch:
        batch["disc_pos_probs"] = _pin(torch.cat([t.reshape(-1) for t in batch["disc_pos_probs"]]), pin)
    return batch


def _pad_last(seq, length):
    """Extend a [t, ...] tensor to [length, ...] by repeating its last entry (index clamp, no copies in a loop)."""
    t = int(seq.shape[0])
    if t > length:
        raise ValueError(f"trajectory of {t} steps exceeds max_traj_len={length}")
    return seq if t == length else seq[torch.clamp(torch.arange(length), max=t - 1)]


def ptv3_collate_fn_partial(max_traj_len, data, pack=False, pin=False):
    """3D-LOTUS++ motion-planner batch (motion_planner_dataset.py:360-410): trajectories shorter than `max_traj_len`
    repeat their last action / target and are masked out by `traj_masks`."""
    batch = _concat_items(data, pin)
    batch["pc_labels"] = torch.cat(batch["pc_labels"], 0)
    for key in ("ee_poses", "gt_trajs_stop"):
        if key in batch:
            batch[key] = torch.stack(batch[key], 0)
    batch["traj_lens"] = [int(t.shape[0]) for t in batch["gt_trajs"]]
    batch["gt_trajs"] = torch.stack([_pad_last(t, max_traj_len) for t in batch["gt_trajs"]], 0)
    batch["traj_masks"] = torch.from_numpy(gen_seq_masks(batch["traj_lens"], max_len=max_traj_len))
    probs = [_pad_last(t, max_traj_len) for t in batch["gt_trajs_disc_pos_probs"]]
    batch["gt_trajs_disc_pos_probs"] = (_pin(torch.cat([t.reshape(max_traj_len, -1) for t in probs], 1), pin)
                                        if pack else probs)
    return batch

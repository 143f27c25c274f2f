"""Canonical synthetic "GemBench-shape" key-step batch.

The real input contract is the output of the reference's `ptv3_collate_fn`
(genrobo3d/train/datasets/simple_policy_dataset.py:391-415): a ragged concat of per-key-step
point clouds that are already 1 cm voxel-unique (preprocess/gen_simple_policy_data.py:89).  No
dataset is available offline, so the bench and the tests use table-top scenes made of
axis-aligned boxes plus an arm-stub column, surface-sampled on unique 1 cm voxels.

Voxel uniqueness survives `trunc((coord - batch_min) / 0.01)` by construction: every cloud is
shifted by a whole number of cells (its centroid rounded to the grid), sub-cell jitter is in
[3, 8] mm, and the points of each cloud's minimal cell per axis sit at exactly 2 mm, so the
batch-global minimum has the smallest sub-cell phase (>= 1 mm margin to every cell face).
"""
import numpy as np
import torch


def _box_surface(lo, hi):
    """All integer voxels on the surface of the box [lo, hi) (inclusive-exclusive)."""
    xs, ys, zs = (np.arange(lo[i], hi[i]) for i in range(3))
    g = np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), -1).reshape(-1, 3)
    on = ((g == lo) | (g == hi - 1)).any(1)
    return g[on]


def synth_cloud(rng, n):
    vox = [_box_surface(np.array([27, 27, 0]), np.array([33, 33, 50]))]  # 6x6x50 cm arm stub
    nbox = int(rng.integers(3, 7))
    total = len(vox[0])
    k = 0
    while k < nbox or total < n * 1.05:
        edge = rng.integers(5, 26, size=3)
        c = rng.integers(0, 60, size=2)
        lo = np.array([c[0] - edge[0] // 2, c[1] - edge[1] // 2, 0])
        v = _box_surface(lo, lo + edge)
        vox.append(v)
        total += len(v)
        k += 1
    vox = np.unique(np.concatenate(vox, 0), axis=0)
    if len(vox) < n:  # overlaps removed too much: recurse with a fresh scene
        return synth_cloud(rng, n)
    vox = vox[rng.permutation(len(vox))[:n]]
    jit = rng.uniform(0.003, 0.008, size=vox.shape)
    for a in range(3):
        jit[vox[:, a] == vox[:, a].min(), a] = 0.002
    cen = np.round(vox.mean(0)).astype(np.int64)
    xyz = ((vox - cen) * 0.01 + jit).astype(np.float32)
    rgb = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    height = (xyz[:, 2:3] - xyz[:, 2].min()).astype(np.float32)
    return np.concatenate([xyz, rgb, height], 1)


def synth_batch(batch_size=16, npoints=4096, ragged=False, seed=0, pos_bins=15, txt_dim=512,
                real_soft_labels=False):
    """Returns a dict with the reference batch schema: pc_fts f32[N,7],
    npoints_in_batch list, offset i64[B], txt_embeds f32[sumL,512], txt_lens list, gt_actions
    f32[B,7], disc_pos_probs list of f32[3, n*2*pos_bins], ee_poses, step_ids."""
    rng = np.random.default_rng(seed)
    pcs, txt, probs = [], [], []
    for b in range(batch_size):
        n = int(rng.integers(npoints // 2, npoints + 1)) if ragged else npoints
        pcs.append(synth_cloud(rng, n))
        L = int(rng.integers(6, 20))
        txt.append(rng.standard_normal((L, txt_dim)).astype(np.float32))
        z = rng.standard_normal((3, n * 2 * pos_bins)).astype(np.float32)
        z = np.exp(z - z.max(1, keepdims=True))
        probs.append((z / z.sum(1, keepdims=True)).astype(np.float32))
    gt = np.concatenate([rng.normal(0, 0.1, size=(batch_size, 3)),
                         rng.integers(0, 72, size=(batch_size, 3)).astype(np.float64),
                         rng.integers(0, 2, size=(batch_size, 1)).astype(np.float64)], 1).astype(np.float32)
    npts = [len(p) for p in pcs]
    # current end-effector pose (xyz, unit quaternion xyzw, open) and key-step index per cloud, from a generator of their own:
    # the draws above (which the committed fixtures rebuild bit for bit) are not disturbed
    rng2 = np.random.default_rng([seed, 0xEE])
    quat = rng2.standard_normal((batch_size, 4))
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    ee = np.concatenate([rng2.normal(0, 0.3, size=(batch_size, 3)), quat, rng2.integers(0, 2, size=(batch_size, 1))], 1).astype(np.float32)
    return {
        "pc_fts": torch.from_numpy(np.concatenate(pcs, 0)),
        "npoints_in_batch": npts,
        "offset": torch.from_numpy(np.cumsum(npts)).long(),
        "txt_embeds": torch.from_numpy(np.concatenate(txt, 0)),
        "txt_lens": [len(t) for t in txt],
        "gt_actions": torch.from_numpy(gt),
        "disc_pos_probs": [torch.from_numpy(p) for p in probs],
        "ee_poses": torch.from_numpy(ee),
        "step_ids": torch.from_numpy(rng2.integers(0, 30, size=batch_size)).long(),
    }


def synth_batch_mp(batch_size=8, npoints=4096, ragged=False, seed=0, pos_bins=15, txt_dim=512, max_traj_len=5):
    """Motion-planner batch (schema of genrobo3d/train/datasets/motion_planner_dataset.py:360-415, use_color False):
    pc_fts f32[N,4] (xyz, height), pc_labels i64[N] in 0..3, gt_trajs f32[B,T,7] (xyz, 3 euler bins, open; short
    trajectories repeat their last action), gt_trajs_stop f32[B,T], traj_lens list, traj_masks bool[B,T],
    gt_trajs_disc_pos_probs list of f32[T,3,n*2*pos_bins], plus the text fields of synth_batch."""
    rng = np.random.default_rng(seed)
    T = max_traj_len
    pcs, labels, txt, probs, trajs, stops, lens = [], [], [], [], [], [], []
    for b in range(batch_size):
        n = int(rng.integers(npoints // 2, npoints + 1)) if ragged else npoints
        pc = synth_cloud(rng, n)
        pcs.append(np.concatenate([pc[:, :3], pc[:, 6:7]], 1))
        labels.append(rng.integers(0, 4, size=n).astype(np.int64))
        L = int(rng.integers(6, 20))
        txt.append(rng.standard_normal((L, txt_dim)).astype(np.float32))
        z = rng.standard_normal((T, 3, n * 2 * pos_bins)).astype(np.float32)
        z = np.exp(z - z.max(2, keepdims=True))
        probs.append((z / z.sum(2, keepdims=True)).astype(np.float32))
        tl = int(rng.integers(1, T + 1))
        tr = np.concatenate([rng.normal(0, 0.1, size=(tl, 3)), rng.integers(0, 72, size=(tl, 3)).astype(np.float64),
                             rng.integers(0, 2, size=(tl, 1)).astype(np.float64)], 1)
        tr = np.concatenate([tr, np.repeat(tr[-1:], T - tl, 0)], 0).astype(np.float32)
        st = np.zeros(T, np.float32)
        st[tl - 1:] = 1.0
        trajs.append(tr); stops.append(st); lens.append(tl)
    npts = [len(p) for p in pcs]
    masks = np.arange(T)[None, :] < np.array(lens)[:, None]
    # current end-effector pose per cloud (xyz, unit quaternion xyzw, open) for action_config.use_ee_pose, from a generator of
    # its own: the draws above (which the committed fixtures rebuild bit for bit) are not disturbed
    rng2 = np.random.default_rng([seed, 0xEE])
    quat = rng2.standard_normal((batch_size, 4))
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    ee = np.concatenate([rng2.normal(0, 0.3, size=(batch_size, 3)), quat, rng2.integers(0, 2, size=(batch_size, 1))], 1).astype(np.float32)
    return {
        "ee_poses": torch.from_numpy(ee),
        "pc_fts": torch.from_numpy(np.concatenate(pcs, 0)),
        "pc_labels": torch.from_numpy(np.concatenate(labels, 0)),
        "npoints_in_batch": npts,
        "offset": torch.from_numpy(np.cumsum(npts)).long(),
        "txt_embeds": torch.from_numpy(np.concatenate(txt, 0)),
        "txt_lens": [len(t) for t in txt],
        "gt_trajs": torch.from_numpy(np.stack(trajs, 0)),
        "gt_trajs_stop": torch.from_numpy(np.stack(stops, 0)),
        "traj_lens": lens,
        "traj_masks": torch.from_numpy(masks),
        "gt_trajs_disc_pos_probs": [torch.from_numpy(p) for p in probs],
    }


def augment_clouds(batch, seed=0, max_rot_deg=45.0, noise=0.002):
    """The geometry part of the training-time augmentation (rotation of every cloud about z by U(-max_rot, max_rot) and
    U(0, noise) metres of per-coordinate jitter; simple_policy_dataset.py:158-181) applied to a synthetic batch in place.
    After re-gridding from the batch-global minimum this puts 1-7 % of the points into a voxel that already holds another
    point (SURVEY.md Trap 5) — the duplicate-voxel case real batches present to the sparse convolutions."""
    rng = np.random.default_rng(seed)
    pc = batch["pc_fts"].numpy().copy()
    start = 0
    for n in batch["npoints_in_batch"]:
        a = np.deg2rad(rng.uniform(-1.0, 1.0) * max_rot_deg)
        c, s = np.cos(a), np.sin(a)
        xy = pc[start:start + n, :2].astype(np.float64)
        pc[start:start + n, 0] = (c * xy[:, 0] - s * xy[:, 1]).astype(np.float32)
        pc[start:start + n, 1] = (s * xy[:, 0] + c * xy[:, 1]).astype(np.float32)
        pc[start:start + n, :3] += rng.uniform(0.0, noise, size=(n, 3)).astype(np.float32)
        start += n
    batch["pc_fts"] = torch.from_numpy(pc)
    return batch


def take_clouds(batch, idx):
    """The sub-batch holding clouds `idx` (in that order) of a synth_batch() dict — what one rank of a data-parallel run gets
    (parallel.shard_clouds)."""
    npts, tl = batch["npoints_in_batch"], batch["txt_lens"]
    po, to = np.concatenate([[0], np.cumsum(npts)]), np.concatenate([[0], np.cumsum(tl)])
    n2 = [npts[i] for i in idx]
    out = {
        "pc_fts": torch.cat([batch["pc_fts"][po[i]:po[i + 1]] for i in idx], 0),
        "npoints_in_batch": n2,
        "offset": torch.from_numpy(np.cumsum(n2)).long(),
        "txt_embeds": torch.cat([batch["txt_embeds"][to[i]:to[i + 1]] for i in idx], 0),
        "txt_lens": [tl[i] for i in idx],
        "gt_actions": batch["gt_actions"][list(idx)],
        "disc_pos_probs": [batch["disc_pos_probs"][i] for i in idx],
        "ee_poses": batch["ee_poses"][list(idx)],
        "step_ids": batch["step_ids"][list(idx)],
    }
    return out


def align_extents(batch):
    """Give every cloud of a synth_batch() the SAME bounding box by moving its first two points to two common corners (one
    cell outside the batch's extent).  The reference voxelises relative to the minimum over the rank-local batch and derives
    the serialisation depth from its maximum (PointTransformerV3/model.py:96-110), so in general a sharded batch is not the
    full batch split up; with a common box every shard sees the lattice origin and depth of the full batch, which makes
    "data-parallel over shards == one process over the whole batch" a testable identity."""
    pc = batch["pc_fts"].clone()
    lo = pc[:, :3].min(0)[0] - 0.01
    hi = pc[:, :3].max(0)[0] + 0.01
    off = np.concatenate([[0], np.cumsum(batch["npoints_in_batch"])])
    for b in range(len(batch["npoints_in_batch"])):
        pc[off[b], :3] = lo
        pc[off[b] + 1, :3] = hi
    out = dict(batch)
    out["pc_fts"] = pc
    return out

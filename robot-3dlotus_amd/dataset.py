"""Episode reader + training-time augmentation for the policy (the remainder of the input path).

What the reference does per key step on CPU workers (genrobo3d/train/datasets/simple_policy_dataset.py:205-363):
read an episode record (LMDB value = msgpack with msgpack-numpy arrays: xyz / rgb per step, bbox_info / pose_info of the
arm links, action = gripper pose per step; written by preprocess/gen_simple_policy_data.py:69-115), drop the table
(z <= TABLE_HEIGHT) and the robot arm (oriented link boxes, genrobo3d/utils/robot_box.py:52-69), cap / thin the cloud,
rotate about z and jitter (:158-181), centre, and emit the item dictionary `data.ptv3_collate_fn` consumes.

This module provides the same item schema for the option family of the published job scripts (euler_disc rotations,
heatmap_disc positions, xyz_shift none | center | gripper).  Differences that are deliberate:

  * the soft position labels are NOT built on the host by default: the policy builds them on the device from
    `gt_actions` (ops.pos_targets; 24 MB per 16 clouds stay off PCIe).  `host_labels=True` restores the reference's
    `disc_pos_probs` entry (same arithmetic, own numpy formulation);
  * the record store is pluggable: an LMDB environment when the `lmdb` package is importable, else a directory of
    `<taskvar>/<key>.msgpack` files with identical values (no LMDB wheel exists in the build image; `pack_episode` /
    `DirStore.write` produce such files, e.g. to convert or to synthesise data);
  * random draws come from the module-level generators in the reference's ORDER (python `random.choice` for the
    instruction, then numpy for sampling, rotation, jitter), so a run seeded like the reference sees the same clouds —
    that is what the parity test against the imported reference relies on.

The oriented-box test is plain geometry (|R^T (p - c)| <= extent / 2); the reference delegates it to open3d, which is
not installed here — parity for that single predicate is therefore by construction, not by comparison.
"""
import json
import os
import random

import msgpack
import numpy as np
import torch
from scipy.spatial.transform import Rotation

TABLE_HEIGHT_RLBENCH = 0.7505  # genrobo3d/configs/rlbench/constants.py:18
_ARM_LINKS = tuple(f"Panda_link{i}" for i in range(8))
_GRIPPER_LINKS = ("Panda_rightfinger", "Panda_leftfinger", "Panda_gripper")
_VISUAL_LINKS = {"Panda_link0", *_GRIPPER_LINKS}  # these carry *_visual_* entries, the others *_respondable_*


# --------------------------------------------------------------------------------------------- record format
def _decode_nd(obj):
    """object_hook for msgpack-numpy's ndarray / numpy-scalar encoding ({nd, type, shape, data})."""
    if b"nd" in obj:
        if obj[b"nd"]:
            descr = obj[b"type"]
            if isinstance(descr, bytes):
                descr = descr.decode()
            dt = np.dtype(descr if isinstance(descr, str) else [tuple(d) for d in descr])
            return np.frombuffer(obj[b"data"], dtype=dt).reshape(obj[b"shape"]).copy()
        t = obj[b"type"]
        return np.frombuffer(obj[b"data"], dtype=np.dtype(t.decode() if isinstance(t, bytes) else t))[0]
    return {(k.decode() if isinstance(k, bytes) else k): v for k, v in obj.items()}


def _encode_nd(obj):
    if isinstance(obj, np.ndarray):
        return {b"nd": True, b"type": obj.dtype.str, b"kind": b"", b"shape": list(obj.shape),
                b"data": np.ascontiguousarray(obj).tobytes()}
    if isinstance(obj, np.generic):
        return {b"nd": False, b"type": obj.dtype.str, b"data": obj.tobytes()}
    raise TypeError(f"cannot pack {type(obj)}")


def unpack_episode(buf):
    """bytes of one LMDB value -> dict (xyz: list[T] of f[n_t, 3], rgb, bbox_info / pose_info: {link: [T, ...]}, action [T, 8])."""
    return msgpack.unpackb(buf, object_hook=_decode_nd, raw=True, strict_map_key=False)


def pack_episode(episode):
    return msgpack.packb(episode, default=_encode_nd, use_bin_type=True)


class DirStore:
    """<root>/<taskvar>/<key>.msgpack — the same values an LMDB environment per task variation would hold."""

    def __init__(self, root):
        self.root = root

    def taskvars(self):
        return sorted(d for d in os.listdir(self.root) if os.path.isdir(os.path.join(self.root, d)))

    def keys(self, taskvar):
        d = os.path.join(self.root, taskvar)
        return sorted(f[:-8].encode() for f in os.listdir(d) if f.endswith(".msgpack"))

    def get(self, taskvar, key):
        with open(os.path.join(self.root, taskvar, key.decode() + ".msgpack"), "rb") as f:
            return f.read()

    def write(self, taskvar, key, episode):
        os.makedirs(os.path.join(self.root, taskvar), exist_ok=True)
        with open(os.path.join(self.root, taskvar, (key.decode() if isinstance(key, bytes) else key) + ".msgpack"), "wb") as f:
            f.write(pack_episode(episode))


class LmdbStore:
    """One read-only LMDB environment per task variation (simple_policy_dataset.py:60-71)."""

    def __init__(self, root):
        import lmdb  # noqa: F401  (absent from the build image; present wherever the reference's data is)

        self.root, self._lmdb, self._txn = root, lmdb, {}

    def taskvars(self):
        return sorted(d for d in os.listdir(self.root) if os.path.isdir(os.path.join(self.root, d)))

    def _t(self, taskvar):
        if taskvar not in self._txn:
            env = self._lmdb.open(os.path.join(self.root, taskvar), readonly=True, lock=False)
            self._txn[taskvar] = env.begin()
        return self._txn[taskvar]

    def keys(self, taskvar):
        return list(self._t(taskvar).cursor().iternext(values=False))

    def get(self, taskvar, key):
        return self._t(taskvar).get(key)


def open_store(root):
    has_mdb = any(os.path.exists(os.path.join(root, d, "data.mdb")) for d in os.listdir(root))
    if has_mdb:
        try:
            return LmdbStore(root)
        except ImportError as e:
            raise ImportError(f"{root} holds LMDB environments but the `lmdb` package is not installed") from e
    return DirStore(root)


# --------------------------------------------------------------------------------------------- geometry
class RobotBox:
    """Oriented boxes of the arm links of one key step (robot_box.py:6-50, RLBench naming): centre = link position,
    axes = link rotation (quaternion xyzw), extent = (max - min) of the link's local bounding box."""

    def __init__(self, arm_links_info, keep_gripper=False):
        bbox_info, pose_info = arm_links_info
        links = _ARM_LINKS if keep_gripper else _ARM_LINKS + _GRIPPER_LINKS
        self.centres, self.rots, self.half = [], [], []
        for link in links:
            kind = "visual" if link in _VISUAL_LINKS else "respondable"
            bbox = np.asarray(bbox_info[f"{link}_{kind}_bbox"], dtype=np.float64)
            pose = np.asarray(pose_info[f"{link}_{kind}_pose"], dtype=np.float64)
            self.centres.append(pose[:3])
            self.rots.append(Rotation.from_quat(pose[3:7]).as_matrix())
            self.half.append(0.5 * (bbox[1::2] - bbox[0::2]))

    def inside(self, xyz):
        """bool [n]: the point lies in at least one link box (faces included)."""
        xyz = np.asarray(xyz, dtype=np.float64)
        hit = np.zeros(len(xyz), dtype=bool)
        for c, r, h in zip(self.centres, self.rots, self.half):
            local = (xyz - c) @ r  # coordinates along the box axes (columns of r)
            hit |= np.all(np.abs(local) <= h, axis=1)
        return hit


def rotate_z(points, angle):
    """Rotation of row vectors about +z (common.py:81-87)."""
    c, s = np.cos(angle), np.sin(angle)
    return np.dot(points, np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]]))


def quaternion_to_discrete_euler(quat, resolution):
    """xyz Euler angles in bins of `resolution` degrees, 0 == -180 deg (utils/rotation_transform.py:151-190), with the
    pitch snap near +-90 deg that removes the gimbal ambiguity."""
    e = Rotation.from_quat(quat).as_euler("xyz", degrees=True)
    near = np.abs(np.abs(e[..., 1]) - 90.0) < 1.0
    if np.any(near):
        e = np.array(e, copy=True)
        e[..., 1] = np.where(near, np.sign(e[..., 1]) * 90.0, e[..., 1])
        e = Rotation.from_euler("xyz", e, degrees=True).as_euler("xyz", degrees=True)
    d = np.around((e + 180.0) / resolution).astype(int)
    d[d == int(360 / resolution)] = 0
    return d


def soft_position_labels(xyz, gt_pos, pos_bins, pos_bin_size, kind="plain", robot_idx=None):
    """get_disc_gt_pos_prob (utils/action_position_utils.py:7-46): per axis a distribution over (point, bin) candidates
    xyz[n, c] + (b - pos_bins) * pos_bin_size that lie within 1 cm of the target coordinate -> f32/f64 [3, n * 2 * pos_bins]."""
    shift = np.arange(-pos_bins, pos_bins) * pos_bin_size
    cand = xyz.T[:, :, None] + shift[None, None, :]                       # [3, n, 2 * pos_bins]
    dist = np.abs(np.asarray(gt_pos)[:3, None, None] - cand)
    if kind == "plain":
        w = np.zeros(dist.shape, dtype=np.float32)
        w[dist < 0.01] = 1
    else:
        w = 1 / np.maximum(dist, 1e-4)
        w[dist > 0.01] = 0
    if robot_idx is not None and len(robot_idx) > 0:
        w[:, robot_idx] = 0
    w, dist = w.reshape(3, -1), dist.reshape(3, -1)
    for c in range(3):
        if np.sum(w[c]) == 0:
            w[c, np.argmin(dist[c])] = 1
    return w / np.sum(w, -1, keepdims=True)


# --------------------------------------------------------------------------------------------- dataset
class KeystepDataset(torch.utils.data.Dataset):
    """Items of the reference's `SimplePolicyDataset` (constructor arguments of the same names and meaning) for
    rot_type 'euler_disc'.  One item = one episode (all key steps but the last) when `all_step_in_batch`, else one step."""

    def __init__(self, data_dir, instr_embed_file, taskvar_instr_file, taskvar_file=None, num_points=10000,
                 xyz_shift="center", xyz_norm=True, use_height=False, rot_type="euler_disc", instr_embed_type="last",
                 all_step_in_batch=True, rm_table=True, rm_robot="none", include_last_step=False, augment_pc=False,
                 sample_points_by_distance=False, same_npoints_per_example=False, rm_pc_outliers=False,
                 euler_resolution=5, pos_type="disc", pos_bins=50, pos_bin_size=0.01, pos_heatmap_type="plain",
                 pos_heatmap_no_robot=False, aug_max_rot=45, real_robot=False, host_labels=False, store=None, **_unused):
        if rot_type != "euler_disc" or real_robot or rm_pc_outliers:
            raise NotImplementedError("lotus-hip reads the published configuration family: rot_type euler_disc, simulated "
                                      "robot, no outlier filter")
        if xyz_shift not in ("none", "center", "gripper") or rm_robot not in ("none", "box", "box_keep_gripper"):
            raise ValueError(f"xyz_shift={xyz_shift!r} / rm_robot={rm_robot!r}")
        self.taskvar_instrs = json.load(open(taskvar_instr_file))
        embeds = np.load(instr_embed_file, allow_pickle=True).item()
        self.instr_embeds = {k: (v[-1:] if instr_embed_type == "last" else v) for k, v in embeds.items()}
        self.store = store if store is not None else open_store(data_dir)
        taskvars = json.load(open(taskvar_file)) if taskvar_file is not None else self.store.taskvars()
        have = set(self.store.taskvars())
        self.ids = []
        for tv in taskvars:
            if tv not in have:
                continue
            for key in self.store.keys(tv):
                if all_step_in_batch:
                    self.ids.append((tv, key, None))
                else:
                    T = len(unpack_episode(self.store.get(tv, key))["xyz"])
                    self.ids.extend((tv, key, t) for t in range(T if include_last_step else T - 1))
        self.opt = dict(num_points=num_points, xyz_shift=xyz_shift, xyz_norm=xyz_norm, use_height=use_height,
                        rm_table=rm_table, rm_robot=rm_robot, include_last_step=include_last_step, augment_pc=augment_pc,
                        by_distance=sample_points_by_distance, same_npoints=same_npoints_per_example,
                        euler_resolution=euler_resolution, pos_type=pos_type, pos_bins=pos_bins, pos_bin_size=pos_bin_size,
                        heatmap=pos_heatmap_type, no_robot=pos_heatmap_no_robot, max_rot=np.deg2rad(aug_max_rot),
                        host_labels=host_labels)

    def __len__(self):
        return len(self.ids)

    # -- per-step pieces ----------------------------------------------------------------------------------------
    def _select_points(self, xyz, ee_xyz):
        o, n = self.opt, len(xyz)
        if n > o["num_points"]:
            if o["by_distance"]:
                from scipy.special import softmax
                w = np.maximum(softmax(1 / np.maximum(np.sqrt(np.sum((xyz - ee_xyz) ** 2, 1)), 0.1)), 1e-30)
                return np.random.choice(n, o["num_points"], replace=False, p=w / sum(w))
            return np.random.choice(n, o["num_points"], replace=False)
        if o["same_npoints"]:
            return np.random.choice(n, o["num_points"], replace=True)
        keep = int(n * np.random.uniform(0.95, 1))
        return np.random.permutation(n)[:keep]

    def _augment(self, xyz, ee_pose, gt_action):
        """z rotation of the scene (cloud, both poses, both orientations) + U(0, 2 mm) jitter (:158-181); the discrete
        rotation target is recomputed from the rotated target orientation."""
        angle = np.random.uniform(-1, 1) * self.opt["max_rot"]
        turn = Rotation.from_euler("z", angle)
        xyz = rotate_z(xyz, angle)
        for pose in (ee_pose, gt_action):
            pose[:3] = rotate_z(pose[:3], angle)
        for pose in (ee_pose, gt_action):
            pose[3:7] = (turn * Rotation.from_quat(pose[3:7])).as_quat()
        rot = quaternion_to_discrete_euler(gt_action[3:7], self.opt["euler_resolution"])
        return xyz + np.random.uniform(0, 0.002, size=xyz.shape), rot

    def __getitem__(self, idx):
        o = self.opt
        taskvar, key, only_t = self.ids[idx]
        ep = unpack_episode(self.store.get(taskvar, key))
        actions = np.asarray(ep["action"])
        T = len(ep["xyz"])
        # rotation target of step t = orientation of the NEXT key pose (the last one repeats) (:188-190)
        nxt = [quaternion_to_discrete_euler(q, o["euler_resolution"]) for q in actions[1:, 3:7]]
        rot_targets = np.stack(nxt + nxt[-1:])
        out = {k: [] for k in ("data_ids", "pc_fts", "step_ids", "pc_centroids", "pc_radius", "ee_poses", "txt_embeds", "gt_actions")}
        if o["pos_type"] == "disc" and o["host_labels"]:
            out["disc_pos_probs"] = []
        if o["no_robot"] and not o["host_labels"]:
            out["robot_point_mask"] = []
        for t in range(T):
            if (only_t is not None and t != only_t) or (not o["include_last_step"] and t == T - 1):
                continue
            xyz, rgb = np.asarray(ep["xyz"][t]), np.asarray(ep["rgb"][t])
            links = ({k: v[t] for k, v in ep["bbox_info"].items()}, {k: v[t] for k, v in ep["pose_info"].items()})
            gt_action = np.array(actions[min(t + 1, T - 1)], dtype=np.float64, copy=True)
            ee_pose = np.array(actions[t], dtype=np.float64, copy=True)
            gt_rot = rot_targets[t]
            embed = self.instr_embeds[random.choice(self.taskvar_instrs[taskvar])]
            if o["rm_table"]:
                keep = xyz[:, 2] > TABLE_HEIGHT_RLBENCH
                xyz, rgb = xyz[keep], rgb[keep]
            if o["rm_robot"] != "none":
                keep = ~RobotBox(links, keep_gripper=o["rm_robot"] == "box_keep_gripper").inside(xyz)
                xyz, rgb = xyz[keep], rgb[keep]
            sel = self._select_points(xyz, ee_pose[:3])
            xyz, rgb = xyz[sel], rgb[sel]
            height = xyz[:, -1] - TABLE_HEIGHT_RLBENCH
            robot_idx = np.nonzero(RobotBox(links).inside(xyz))[0] if o["no_robot"] else None
            if o["augment_pc"]:
                xyz, gt_rot = self._augment(xyz, ee_pose, gt_action)
            centroid = {"none": np.zeros(3), "center": np.mean(xyz, 0) if len(xyz) else np.zeros(3),
                        "gripper": ee_pose[:3].copy()}[o["xyz_shift"]]
            radius = np.max(np.sqrt(np.sum((xyz - centroid) ** 2, axis=1))) if (o["xyz_norm"] and len(xyz)) else 1
            xyz, height = (xyz - centroid) / radius, height / radius
            gt_action[:3] = (gt_action[:3] - centroid) / radius
            ee_pose[:3] = (ee_pose[:3] - centroid) / radius
            if len(xyz) == 0:
                continue
            feats = [xyz, rgb / 255.0 * 2 - 1] + ([height[:, None]] if o["use_height"] else [])
            target = np.concatenate([gt_action[:3], gt_rot, gt_action[-1:]], 0)
            out["pc_centroids"].append(centroid)
            out["pc_radius"].append(radius)
            if "disc_pos_probs" in out:
                out["disc_pos_probs"].append(torch.from_numpy(soft_position_labels(
                    xyz, target[:3], o["pos_bins"], o["pos_bin_size"], o["heatmap"], robot_idx)))
            if "robot_point_mask" in out:
                m = np.zeros(len(xyz), dtype=bool)
                m[robot_idx] = True
                out["robot_point_mask"].append(torch.from_numpy(m))
            out["data_ids"].append(f"{taskvar}-{key.decode('ascii')}-t{t}")
            out["pc_fts"].append(torch.from_numpy(np.concatenate(feats, 1)).float())
            out["txt_embeds"].append(torch.from_numpy(np.asarray(embed)).float())
            out["ee_poses"].append(torch.from_numpy(ee_pose).float())
            out["gt_actions"].append(torch.from_numpy(target).float())
            out["step_ids"].append(t)
        return out


def synth_episode(rng, steps=4, points=3000):
    """A synthetic episode record in the reference's format (for tests, benchmarks and format conversion checks):
    1 cm-unique surface points above and below the table height, an arm made of link boxes, key poses."""
    from .synth import synth_cloud

    xyz, rgb = [], []
    for _ in range(steps):
        pc = synth_cloud(rng, points)[:, :3].astype(np.float64)
        pc[:, 2] += TABLE_HEIGHT_RLBENCH + 0.2 + rng.uniform(-0.25, 0.0)  # part of the cloud dips below the table plane
        xyz.append(pc)
        rgb.append(rng.integers(0, 256, size=(len(pc), 3)).astype(np.float64))
    bbox, pose = {}, {}
    for link in _ARM_LINKS + _GRIPPER_LINKS:
        kind = "visual" if link in _VISUAL_LINKS else "respondable"
        half = rng.uniform(0.03, 0.12, size=(steps, 3))
        bbox[f"{link}_{kind}_bbox"] = np.stack([-half[:, 0], half[:, 0], -half[:, 1], half[:, 1], -half[:, 2], half[:, 2]], 1)
        centre = np.stack([rng.uniform(-0.3, 0.3, steps), rng.uniform(-0.3, 0.3, steps),
                           TABLE_HEIGHT_RLBENCH + rng.uniform(0.0, 0.5, steps)], 1)
        pose[f"{link}_{kind}_pose"] = np.concatenate([centre, Rotation.random(steps, random_state=int(rng.integers(1 << 30))).as_quat()], 1)
    action = np.concatenate([rng.uniform(-0.3, 0.3, (steps, 2)), TABLE_HEIGHT_RLBENCH + rng.uniform(0.05, 0.5, (steps, 1)),
                             Rotation.random(steps, random_state=int(rng.integers(1 << 30))).as_quat(),
                             rng.integers(0, 2, (steps, 1)).astype(np.float64)], 1)
    return {"xyz": xyz, "rgb": rgb, "bbox_info": bbox, "pose_info": pose, "action": action,
            "key_frameids": np.arange(steps) * 10}

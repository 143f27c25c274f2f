"""Pin the un-vendored third-party semantics against a REAL reference checkpoint (the moment one is reachable).

    python tools/validate_checkpoint.py <model_step_N.pt> <episode_store_dir> --instr-embeds E.npy --taskvar-instrs I.json
    python tools/validate_checkpoint.py --self-test          # synthetic checkpoint + synthetic episodes (no downloads)

What cannot be verified offline: spconv 2.3.6 stores SubMConv3d.weight as (Cout, kx, ky, kz, Cin)
and enumerates taps x-major over indices[:, 1:4]; both are assumptions shared by the oracle and the HIP kernels.  A trained
checkpoint disambiguates them: with the right layout the policy reproduces the demonstrated actions of its training episodes
(position error of a few millimetres, the demonstrated gripper state), with a wrong one it predicts noise.  This script
  1. loads the checkpoint with load_state_dict(strict=True) into robot_3dlotus_amd.policy.SimplePolicyPTV3CA (the reference's
     key grammar) and reports the 5-D convolution weight shapes it found;
  2. runs eval-mode inference (the reference's `forward(batch, compute_loss=False)`, simple_policy_ptv3.py:225-306) on every
     key step of the given episodes through the episode reader (dataset.KeystepDataset: table / robot-box removal, voxel
     subsampling, centring — no augmentation);
  3. compares the predicted action with the recorded next-key-step action: position error (m), rotation error (deg),
     open/close agreement — and repeats it with the two alternative weight interpretations (tap axes reversed: z-major;
     Cin / Cout transposed where the layer is square) so that the interpretation the checkpoint was trained with stands out.
Exit code 0 iff the default interpretation has the lowest mean position error."""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import config as lcfg, data as ld, dataset as ds  # noqa: E402
from robot_3dlotus_amd.policy import SimplePolicyPTV3CA  # noqa: E402


def reinterpret(sd, how):
    """State dict as it would have to be re-laid-out if spconv's convention differed from the one assumed."""
    out = {}
    for k, v in sd.items():
        if isinstance(v, torch.Tensor) and v.ndim == 5:
            if how == "taps_z_major":          # taps enumerated z-major instead of x-major
                v = v.permute(0, 3, 2, 1, 4).contiguous()
            elif how == "cin_cout_swapped" and v.shape[0] == v.shape[-1]:   # (Cin, k, k, k, Cout)
                v = v.permute(4, 1, 2, 3, 0).contiguous()
        out[k] = v
    return out


def rot_err_deg(q_pred, q_gt):
    d = abs(float(np.dot(q_pred / np.linalg.norm(q_pred), q_gt / np.linalg.norm(q_gt))))
    return float(np.degrees(2 * np.arccos(min(1.0, d))))


@torch.no_grad()
def evaluate(model, loader, max_steps):
    pos, rot, opn, n = [], [], [], 0
    for batch in loader:
        gt = batch["gt_actions_raw"] if "gt_actions_raw" in batch else None
        acts = model(batch, compute_loss=False).cpu().numpy()      # f64 [B, 8]: xyz, quaternion, open
        centre = np.stack(batch["pc_centroids"]) if len(batch.get("pc_centroids", [])) else 0.0
        tgt = np.stack([np.asarray(a, dtype=np.float64) for a in batch["gt_actions_world"]])
        for a, t in zip(acts, tgt):
            pos.append(float(np.linalg.norm(a[:3] + (centre if np.ndim(centre) == 0 else 0.0) - t[:3])))
            rot.append(rot_err_deg(a[3:7], t[3:7]))
            opn.append(float((a[7] > 0) == (t[7] > 0.5)))
        n += len(acts)
        if n >= max_steps:
            break
    return dict(n=n, pos_err_m=float(np.mean(pos)), pos_err_median_m=float(np.median(pos)), rot_err_deg=float(np.mean(rot)),
                open_acc=float(np.mean(opn)))


def collate(items):
    b = ld.ptv3_collate_fn(items)
    # world-frame targets for the report: gt_actions are (centred xyz, discrete euler bins, open); keep the raw record too
    b["gt_actions_world"] = [np.concatenate([np.asarray(p, dtype=np.float64), np.asarray(q, dtype=np.float64), [float(o)]])
                             for it in items for p, q, o in zip(it["gt_pos_world"], it["gt_quat"], it["gt_open"])]
    return b


class _WithWorldTargets(ds.KeystepDataset):
    """KeystepDataset item + the un-centred target pose of every key step (for the report only)."""

    def __getitem__(self, idx):
        item = super().__getitem__(idx)
        n = len(item["pc_fts"])
        cent = item["pc_centroids"] if len(item["pc_centroids"]) else [np.zeros(3)] * n
        item["gt_pos_world"] = [np.asarray(g[:3], dtype=np.float64) + np.asarray(c, dtype=np.float64)[:3] for g, c in zip(item["gt_actions"], cent)]
        item["gt_quat"] = [ds_quat_from_bins(np.asarray(g[3:6])) for g in item["gt_actions"]]
        item["gt_open"] = [float(g[6]) for g in item["gt_actions"]]
        return item


def ds_quat_from_bins(bins, resolution=5):
    from scipy.spatial.transform import Rotation as R
    return R.from_euler("xyz", np.asarray(bins, dtype=np.float64) * resolution - 180, degrees=True).as_quat()


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkpoint", nargs="?")
    ap.add_argument("episodes", nargs="?", help="episode store (directory or LMDB root, dataset.open_store)")
    ap.add_argument("--instr-embeds"); ap.add_argument("--taskvar-instrs"); ap.add_argument("--taskvars", default=None)
    ap.add_argument("--preset", default="v1", help="model preset (config.preset): v1 | peract | tiny")
    ap.add_argument("--max-steps", type=int, default=200)
    ap.add_argument("--self-test", action="store_true")
    args = ap.parse_args()
    cfg = lcfg.preset("tiny" if args.self_test else args.preset)
    tmp = None
    if args.self_test:
        tmp = tempfile.mkdtemp()
        rng = np.random.default_rng(0)
        store = ds.DirStore(os.path.join(tmp, "eps"))
        for e in range(3):
            store.write("close_jar+0", f"episode{e}".encode(), ds.synth_episode(rng, steps=5, points=3000))
        json.dump({"close_jar+0": ["close the jar"]}, open(os.path.join(tmp, "i.json"), "w"))
        np.save(os.path.join(tmp, "e.npy"), {"close the jar": rng.standard_normal((9, 512)).astype(np.float32)}, allow_pickle=True)
        torch.manual_seed(0)
        torch.save(SimplePolicyPTV3CA(cfg).state_dict(), os.path.join(tmp, "model_step_0.pt"))
        args.checkpoint, args.episodes = os.path.join(tmp, "model_step_0.pt"), os.path.join(tmp, "eps")
        args.instr_embeds, args.taskvar_instrs = os.path.join(tmp, "e.npy"), os.path.join(tmp, "i.json")
    if not (args.checkpoint and args.episodes and args.instr_embeds and args.taskvar_instrs):
        ap.error("checkpoint, episodes, --instr-embeds and --taskvar-instrs are required (or --self-test)")
    if not torch.cuda.is_available():
        raise SystemExit("validate_checkpoint.py runs the HIP model: a GPU is required (no CPU fallback)")
    sd = torch.load(args.checkpoint, map_location="cpu")
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    conv = {k: tuple(v.shape) for k, v in sd.items() if isinstance(v, torch.Tensor) and v.ndim == 5}
    dset = _WithWorldTargets(None, args.instr_embeds, args.taskvar_instrs, taskvar_file=args.taskvars, store=ds.open_store(args.episodes),
                             num_points=4096, xyz_shift="center", xyz_norm=False, use_height=True, instr_embed_type="all",
                             rm_robot="box_keep_gripper", augment_pc=False, pos_bins=cfg.action_config.pos_bins, pos_bin_size=0.01)
    loader = torch.utils.data.DataLoader(dset, batch_size=2, shuffle=False, num_workers=0, collate_fn=collate)
    report = {"checkpoint": args.checkpoint, "conv_weight_shapes": conv, "interpretations": {}}
    for how in ("as_assumed", "taps_z_major", "cin_cout_swapped"):
        model = SimplePolicyPTV3CA(cfg)
        missing = model.load_state_dict(reinterpret(sd, how), strict=True)   # raises on any key / shape mismatch
        assert not missing.missing_keys and not missing.unexpected_keys
        model = model.cuda().eval()
        report["interpretations"][how] = evaluate(model, loader, args.max_steps)
    best = min(report["interpretations"], key=lambda k: report["interpretations"][k]["pos_err_m"])
    report["lowest_position_error"] = best
    report["verdict"] = ("self-test: pipeline ran on a randomly initialised checkpoint (errors are meaningless)" if args.self_test else
                         ("assumed spconv layout CONFIRMED" if best == "as_assumed" else f"assumed layout REFUTED: `{best}` fits better"))
    print(json.dumps(report, indent=1))
    return 0 if (best == "as_assumed" or args.self_test) else 1


if __name__ == "__main__":
    sys.exit(main())

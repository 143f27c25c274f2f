"""Golden items of the imported reference `SimplePolicyDataset` (build container only: needs /root/reference).

    python tests/golden/make_golden_dataset.py      -> tests/golden/dataset_items.npz

The fixture holds DATA only: two small synthetic episode records (msgpack bytes in the reference's record format),
the instruction table, and for a fixed option set / seeds the item dictionaries the reference dataset returned."""
import json
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]
import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import dataset as ds  # noqa: E402
import test_host_dataset as th  # noqa: E402

OPTS = dict(rot_type="euler_disc", pos_type="disc", pos_bins=15, pos_bin_size=0.01, euler_resolution=5, rm_robot="box_keep_gripper",
            augment_pc=True, aug_max_rot=180, xyz_shift="center", xyz_norm=False, use_height=True, instr_embed_type="all",
            num_points=200, pos_heatmap_type="plain")


def main():
    rng = np.random.default_rng(2024)
    tmp = tempfile.mkdtemp()
    store = ds.DirStore(os.path.join(tmp, "eps"))
    taskvar = "push_button+0"
    recs = {}
    for e in range(2):
        ep = ds.synth_episode(rng, steps=3, points=260)
        ep["xyz"] = [x.astype(np.float32) for x in ep["xyz"]]
        ep["rgb"] = [x.astype(np.uint8) for x in ep["rgb"]]
        store.write(taskvar, f"episode{e}".encode(), ep)
        recs[f"episode{e}"] = np.frombuffer(store.get(taskvar, f"episode{e}".encode()), dtype=np.uint8)
    instrs = {taskvar: ["push the button", "press it"]}
    embeds = {s: rng.standard_normal((5, 8)).astype(np.float32) for s in instrs[taskvar]}
    json.dump(instrs, open(os.path.join(tmp, "instr.json"), "w"))
    np.save(os.path.join(tmp, "embeds.npy"), embeds, allow_pickle=True)
    th._install_reference_standins()
    from genrobo3d.train.datasets.simple_policy_dataset import SimplePolicyDataset

    ref = SimplePolicyDataset(store.root, os.path.join(tmp, "embeds.npy"), os.path.join(tmp, "instr.json"), **OPTS)
    out = {"opts": json.dumps(OPTS), "taskvar": taskvar, "instrs": json.dumps(instrs)}
    for k, v in recs.items():
        out["rec/" + k] = v
    for s, v in embeds.items():
        out["embed/" + s] = v
    for idx in range(len(ref)):
        random.seed(7 + idx); np.random.seed(7 + idx)
        item = ref[idx]
        for k, vals in item.items():
            for j, v in enumerate(vals):
                out[f"item{idx}/{k}/{j}"] = np.asarray(v.numpy() if hasattr(v, "numpy") else v)
    np.savez_compressed(os.path.join(HERE, "dataset_items.npz"), **out)
    print("wrote", os.path.join(HERE, "dataset_items.npz"), os.path.getsize(os.path.join(HERE, "dataset_items.npz")), "bytes")


if __name__ == "__main__":
    main()

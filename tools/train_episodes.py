"""The trainer loop fed by the episode reader: synthetic episode records in the reference's format -> KeystepDataset
(table / robot-box removal, 4096-point sampling, z rotation + jitter, centring; soft labels built on the device) ->
DataLoader workers with the pinned collate function -> prefetch() / forward / backward / fused AdamW on one MI355X.
Prints the end-to-end keystep-samples/s and what the loader alone delivers.

    python tools/train_episodes.py [--workers 16 --episodes 96 --steps 60]
"""
import argparse
import json
import os
import sys
import tempfile
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import config as lcfg, data as ld, dataset as ds, optim as loptim  # noqa: E402
from robot_3dlotus_amd.policy import SimplePolicyPTV3CA  # noqa: E402


def collate(items):
    return ld.ptv3_collate_fn(items, pin=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--episodes", type=int, default=96)
    ap.add_argument("--episodes-per-batch", type=int, default=3)   # x 5-6 key steps = 15-18 clouds per step
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup-steps", type=int, default=5000, help="lr warm-up of the schedule (reference: 5000)")
    ap.add_argument("--curve", default=None, help="write the loss every 20 steps to this JSON file")
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    tmp = tempfile.mkdtemp()
    store = ds.DirStore(os.path.join(tmp, "eps"))
    tv = "close_jar+0"
    for e in range(args.episodes):
        store.write(tv, f"episode{e}".encode(), ds.synth_episode(rng, steps=int(rng.integers(6, 8)), points=7000))
    json.dump({tv: ["close the jar", "screw the lid on"]}, open(os.path.join(tmp, "i.json"), "w"))
    np.save(os.path.join(tmp, "e.npy"), {s: rng.standard_normal((int(rng.integers(6, 20)), 512)).astype(np.float32)
                                         for s in ("close the jar", "screw the lid on")}, allow_pickle=True)
    dset = ds.KeystepDataset(None, os.path.join(tmp, "e.npy"), os.path.join(tmp, "i.json"), store=store, num_points=4096,
                             xyz_shift="center", xyz_norm=False, use_height=True, instr_embed_type="all", rm_robot="box_keep_gripper",
                             augment_pc=True, aug_max_rot=180, pos_bins=15, pos_bin_size=0.01)
    loader = torch.utils.data.DataLoader(dset, batch_size=args.episodes_per_batch, shuffle=True, num_workers=args.workers,
                                         collate_fn=collate, drop_last=True, persistent_workers=True, prefetch_factor=4)
    # loader alone
    n, t0 = 0, None
    for epoch in range(3):
        for b in loader:
            if t0 is None:
                t0 = time.perf_counter()      # (first batch: worker start-up)
                continue
            n += len(b["npoints_in_batch"])
    loader_rate = n / (time.perf_counter() - t0)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = SimplePolicyPTV3CA(lcfg.preset("v1")).to(dev).train()
    topts = SimpleNamespace(learning_rate=1e-4, weight_decay=0.05, optim="adamw", betas=[0.9, 0.98], lr_sched="cosine",
                            warmup_steps=args.warmup_steps, num_train_steps=150000, grad_norm=10.0)
    opt, init_lrs = loptim.build_optimizer(model, topts)
    torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    step, seen, t1 = 0, 0, None
    curve = []
    it = iter(loader)
    nxt = next(it)
    model.prefetch(nxt)
    while step < args.steps:
        batch = nxt
        try:
            nxt = next(it)
        except StopIteration:
            it = iter(loader)
            nxt = next(it)
        opt.zero_grad(set_to_none=True)
        _, losses = model(batch, compute_loss=True, compute_final_action=False)
        model.prefetch(nxt)
        losses["total"].backward()
        loptim.set_lr(opt, init_lrs, step + 1, topts)
        opt.clip_grad_norm_(topts.grad_norm)
        opt.step()
        step += 1
        if args.curve and step % 20 == 0:
            curve.append((step, torch.stack([losses[k].detach() for k in ("total", "pos", "rot", "open")])))  # no host sync
        if step == 10:
            torch.cuda.synchronize()
            t1, seen = time.perf_counter(), 0
        seen += len(batch["npoints_in_batch"])
    torch.cuda.synchronize()
    rate = (seen - len(batch["npoints_in_batch"]) * 0) / (time.perf_counter() - t1)
    print(json.dumps({"loader_only_keysteps_per_s": round(loader_rate, 1), "train_keysteps_per_s": round(rate, 1),
                      "workers": args.workers, "clouds_per_step": round(seen / (args.steps - 10), 1), "loss": round(losses["total"].item(), 4),
                      "note": "episode records -> KeystepDataset -> DataLoader(pin) -> prefetch/forward/backward/AdamW, v1 model, 4096-point clouds"}))
    if args.curve:
        write_curve(args.curve, curve, rate)


def write_curve(path, curve, rate):
    rows = [[st] + [round(float(v), 4) for v in t.cpu()] for st, t in curve]
    json.dump({"columns": ["step", "total", "pos", "rot", "open"], "train_keysteps_per_s": round(rate, 1), "rows": rows},
              open(path, "w"))


if __name__ == "__main__":
    main()

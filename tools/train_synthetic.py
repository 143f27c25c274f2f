"""Minimal trainer on synthetic key-step batches — the loop of genrobo3d/train/train_simple_policy.py:196-262 with the
lotus-hip drop-ins (model, optimiser, gradient reducer, checkpoints).  One process per GPU:

    python tools/train_synthetic.py --steps 50                        # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_synthetic.py --steps 50
"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import checkpoint, config as lcfg, data, ops, optim as loptim, parallel, synth  # noqa: E402
from robot_3dlotus_amd.policy import MODEL_FACTORY  # noqa: E402


def keysteps_as_items(batch):
    """Split a synthetic batch back into per-key-step items (what a dataset would yield) for the collate function."""
    items, o = [], 0
    txt_o = 0
    for i, n in enumerate(batch["npoints_in_batch"]):
        L = batch["txt_lens"][i]
        items.append({"pc_fts": [batch["pc_fts"][o:o + n]], "txt_embeds": [batch["txt_embeds"][txt_o:txt_o + L]],
                      "gt_actions": [batch["gt_actions"][i]], "disc_pos_probs": [batch["disc_pos_probs"][i]],
                      "ee_poses": [batch["ee_poses"][i]], "step_ids": [int(batch["step_ids"][i])], "pc_centroids": []})
        o += n
        txt_o += L
    return items


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--npoints", type=int, default=4096)
    ap.add_argument("--preset", default="v1")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16"])
    ap.add_argument("--output-dir", default=None)
    args = ap.parse_args()
    rank, local, world = parallel.init_distributed()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.manual_seed(2024)
    cfg = lcfg.preset(args.preset)
    model = MODEL_FACTORY[cfg.model_class](cfg).to(dev).train()
    topts = SimpleNamespace(learning_rate=1e-4, weight_decay=0.05, optim="adamw", betas=[0.9, 0.98], lr_sched="cosine",
                            warmup_steps=5000, num_train_steps=150000, grad_norm=10.0)      # simple_policy_ptv3.yaml TRAIN
    opt, init_lrs = loptim.build_optimizer(model, topts)
    reducer = parallel.GradReducer(model) if world > 1 else None
    parallel.enable_sync_batchnorm()
    ops.set_gemm_precision(args.precision)
    saver = checkpoint.ModelSaver(args.output_dir) if (args.output_dir and rank == 0) else None

    def load(step):  # "dataset": a fresh synthetic batch per step and rank, through the collate function, pinned + packed
        b = synth.synth_batch(args.batch, args.npoints, ragged=True, seed=1000 * rank + step)
        return data.ptv3_collate_fn(keysteps_as_items(b), pack=True, pin=True)

    nxt = load(0)
    t0 = time.perf_counter()
    for step in range(args.steps):
        batch, nxt = nxt, (load(step + 1) if step + 1 < args.steps else None)
        if reducer is not None:
            reducer.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        _, losses = model(batch, compute_loss=True, compute_final_action=False)
        if nxt is not None:
            model.prefetch(nxt)                       # next batch's integer front-end runs under this backward
        losses["total"].backward()
        if reducer is not None:
            reducer.finish()
        loptim.set_lr(opt, init_lrs, step + 1, topts)  # global_step is incremented before scheduling (train_simple_policy.py:225-229)
        gn = opt.clip_grad_norm_(topts.grad_norm)
        opt.step()
        if rank == 0 and (step % 10 == 0 or step == args.steps - 1):
            print(f"step {step:4d}  " + "  ".join(f"{k} {v.item():.4f}" for k, v in losses.items()) + f"  |g| {float(gn):.3f}", flush=True)
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.perf_counter() - t0
        print(f"{args.steps * args.batch * world / dt:.1f} keystep-samples/s incl. host-side batch synthesis + collate ({world} GPU)")
        if saver is not None:
            print("saved", saver.save(model, args.steps, optimizer=opt, rewrite_optimizer=True))


if __name__ == "__main__":
    main()

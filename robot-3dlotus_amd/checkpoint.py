"""Checkpoint writer / reader with the reference's file formats.

`ModelSaver` mirrors genrobo3d/train/utils/save.py:26-55: `model_step_<N>.pt` is a plain state_dict on the CPU
(`module.` prefixes stripped) and `train_state_<N>.pt` / `train_state_latest.pt` is `{'step', 'optimizer'}`.
`load_model_checkpoint` / `find_resume_state` mirror the resume logic of train_simple_policy.py:130-183, so a run can
be resumed from — or hand its checkpoints to — the reference trainer unchanged (the drop-in modules keep the
reference's state_dict keys; robot_3dlotus_amd.optim.AdamW keeps torch's optimizer state layout).
"""
import os

import torch


class ModelSaver(object):
    def __init__(self, output_dir, prefix="model_step", suffix="pt"):
        self.output_dir, self.prefix, self.suffix = output_dir, prefix, suffix

    def save(self, model, step, optimizer=None, rewrite_optimizer=False):
        os.makedirs(self.output_dir, exist_ok=True)
        path = os.path.join(self.output_dir, f"{self.prefix}_{step}.{self.suffix}")
        state = {}
        for k, v in model.state_dict().items():
            if k.startswith("module."):
                k = k[7:]
            state[k] = v.cpu() if isinstance(v, torch.Tensor) else v
        torch.save(state, path)
        if optimizer is not None:
            dump = {"step": step, "optimizer": optimizer.state_dict()}
            name = "train_state_latest.pt" if rewrite_optimizer else f"train_state_{step}.pt"
            torch.save(dump, os.path.join(self.output_dir, name))
        return path


def find_resume_state(ckpt_dir, resume_training=True, checkpoint=None):
    """-> (model_checkpoint_file | None, optimizer_checkpoint | None, global_step), train_simple_policy.py:130-152:
    `train_state_latest.pt` wins over the configured `checkpoint` when resuming."""
    opt_file = os.path.join(ckpt_dir, "train_state_latest.pt")
    if os.path.exists(opt_file) and resume_training:
        opt_ckpt = torch.load(opt_file, map_location="cpu")
        latest = os.path.join(ckpt_dir, "model_step_%d.pt" % opt_ckpt["step"])
        return (latest if os.path.exists(latest) else checkpoint), opt_ckpt, int(opt_ckpt["step"])
    return checkpoint, None, 0


def load_model_checkpoint(model, path, strict=False):
    """train_simple_policy.py:154-173: keep the entries whose name and shape match, then load_state_dict(strict)."""
    checkpoint = torch.load(path, map_location="cpu")
    own = model.state_dict()
    kept = {k: v for k, v in checkpoint.items() if k in own and v.size() == own[k].size()}
    missing = model.load_state_dict(kept, strict=strict)
    return len(kept), missing

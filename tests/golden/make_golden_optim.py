"""Golden optimiser trajectory from the IMPORTED reference (run in the build container only):

    python tests/golden/make_golden_optim.py   ->  tests/golden/optim_traj.npz

A 4-tensor toy model (two weights, two biases, named so that the reference's name-based grouping puts them in
different weight-decay groups) is driven for 6 steps with seeded gradients through the reference's
build_optimizer -> AdamW, clip_grad_norm_(10) and the cosine schedule of train_simple_policy.py:225-241."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = torch.nn.Linear(5, 7)
        self.LayerNorm = torch.nn.LayerNorm(7)
        self.fc2 = torch.nn.Linear(7, 3)


def main():
    from types import SimpleNamespace
    from genrobo3d.train.optim.misc import build_optimizer
    from genrobo3d.train.optim.sched import get_lr_sched_decay_rate

    torch.manual_seed(0)
    net = Toy()
    opts = SimpleNamespace(learning_rate=1e-4, weight_decay=0.05, optim="adamw", betas=[0.9, 0.98], lr_sched="cosine",
                           warmup_steps=3, num_train_steps=20, grad_norm=10.0)
    optim, init_lrs = build_optimizer(net, opts)
    names = [n for n, _ in net.named_parameters()]
    out = {"names": np.array(names), "p0": np.concatenate([p.detach().numpy().ravel() for p in net.parameters()])}
    out["sizes"] = np.array([p.numel() for p in net.parameters()])
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        rate = get_lr_sched_decay_rate(step, opts)
        for kp, grp in enumerate(optim.param_groups):
            grp["lr"] = max(init_lrs[kp] * rate, 1e-8)
        scale = 40.0 if step == 2 else 1.0  # one step exceeds the clip threshold
        grads = [torch.randn(p.shape, generator=g) * scale for p in net.parameters()]
        for p, gr in zip(net.parameters(), grads):
            p.grad = gr.clone()
        norm = torch.nn.utils.clip_grad_norm_(net.parameters(), opts.grad_norm)
        optim.step()
        out[f"g{step}"] = np.concatenate([x.numpy().ravel() for x in grads])
        out[f"p{step + 1}"] = np.concatenate([p.detach().numpy().ravel() for p in net.parameters()])
        out[f"norm{step}"] = np.float64(norm)
        out[f"lr{step}"] = np.float64(optim.param_groups[0]["lr"])
    out["hyper"] = np.array([opts.learning_rate, opts.weight_decay, opts.betas[0], opts.betas[1], opts.warmup_steps,
                             opts.num_train_steps, opts.grad_norm])
    np.savez_compressed(os.path.join(HERE, "optim_traj.npz"), **out)
    print("wrote optim_traj.npz", {k: v.shape for k, v in out.items() if k.startswith("p")})


if __name__ == "__main__":
    main()

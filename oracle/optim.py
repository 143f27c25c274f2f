"""ORACLE (test infrastructure, never shipped as product): the optimiser step of the 3D-LOTUS trainer.

numpy float32 restatement of
  * the HF-style AdamW the reference selects with `optim: 'adamw'` (genrobo3d/train/optim/adamw.py:53-112):
    eps is added to sqrt(v) OUTSIDE the bias correction, step_size = lr * sqrt(1 - b2^t) / (1 - b1^t), decoupled
    weight decay `p -= lr * wd * p` applied AFTER the Adam update (on the updated p);
  * torch.nn.utils.clip_grad_norm_(parameters, max_norm) as called at train_simple_policy.py:237-241
    (total L2 norm over all gradients, coefficient max_norm / (norm + 1e-6) clamped to 1);
  * the learning-rate schedule: get_lr_sched_decay_rate (optim/sched.py:95-113: warmup_cosine :44-55, floor 1e-5) and
    `lr = max(init_lr * rate, 1e-8)` (train_simple_policy.py:229);
  * the name-based weight-decay grouping of build_optimizer (optim/misc.py:13-55).

Only `tests/` may import this.  Pinned against the imported reference by tests/test_oracle_vs_reference.py and the
golden trajectory tests/golden/optim_traj.npz (made by tests/golden/make_golden_optim.py)."""
import math

import numpy as np

F32 = np.float32


def warmup_cosine(step, warmup_step, tot_step, num_cycles=0.5):
    """optim/sched.py:44-55"""
    if step < warmup_step:
        return step / warmup_step
    progress = float(step - warmup_step) / float(max(1, tot_step - warmup_step))
    return 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress))


def lr_at(step, init_lr, warmup_step, tot_step):
    """get_lr_sched_decay_rate (sched.py:95-113, 'cosine') + train_simple_policy.py:229"""
    rate = max(warmup_cosine(step, warmup_step, tot_step), 1e-5)
    return max(init_lr * rate, 1e-8)


def no_decay(name):
    """optim/misc.py:14,35-40: substring match on the parameter name"""
    return any(nd in name for nd in ("bias", "LayerNorm.bias", "LayerNorm.weight"))


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (norm_type 2): returns (total_norm, clipped grads)."""
    total = math.sqrt(sum(float(np.sum(g.astype(np.float64) ** 2)) for g in grads))
    coef = min(max_norm / (total + 1e-6), 1.0)
    return total, [(g * F32(coef)).astype(F32) for g in grads]


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0, correct_bias=True):
    """One AdamW.step() for one tensor (adamw.py:74-110), float32 arithmetic in the reference's order.
    `step` is the 1-based count AFTER the increment.  Returns (p, m, v)."""
    p, g, m, v = (np.asarray(a, dtype=F32) for a in (p, g, m, v))
    m = (m * F32(beta1) + g * F32(1.0 - beta1)).astype(F32)
    v = (v * F32(beta2) + (g * g) * F32(1.0 - beta2)).astype(F32)
    denom = (np.sqrt(v) + F32(eps)).astype(F32)
    step_size = lr
    if correct_bias:
        step_size = step_size * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p = (p + F32(-step_size) * (m / denom)).astype(F32)
    if weight_decay > 0.0:
        p = (p + F32(-lr * weight_decay) * p).astype(F32)
    return p, m, v

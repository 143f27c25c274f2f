"""The optimiser oracle (oracle/optim.py) against the golden trajectory captured from the imported reference
(build_optimizer -> AdamW, clip_grad_norm_(10), cosine schedule; tests/golden/make_golden_optim.py)."""
import os

import numpy as np

from oracle import optim as oo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim_traj.npz")


def replay(step_fn):
    fx = np.load(GOLD)
    lr0, wd, b1, b2, warm, total, max_norm = fx["hyper"].tolist()
    names, sizes = [str(n) for n in fx["names"]], fx["sizes"].tolist()
    off = np.concatenate([[0], np.cumsum(sizes)])
    p = [fx["p0"][off[i]:off[i + 1]].astype(np.float32) for i in range(len(sizes))]
    m = [np.zeros_like(x) for x in p]
    v = [np.zeros_like(x) for x in p]
    worst = 0.0
    for step in range(6):
        lr = oo.lr_at(step, lr0, int(warm), int(total))
        assert abs(lr - float(fx[f"lr{step}"])) <= 1e-12 * max(1.0, lr)
        g = [fx[f"g{step}"][off[i]:off[i + 1]].astype(np.float32) for i in range(len(sizes))]
        norm, gc = oo.clip_grad_norm(g, max_norm)
        assert abs(norm - float(fx[f"norm{step}"])) <= 2e-6 * norm
        p, m, v = step_fn(names, p, gc, m, v, step + 1, lr, b1, b2, wd)
        ref = fx[f"p{step + 1}"]
        got = np.concatenate(p)
        worst = max(worst, float(np.abs(got - ref).max()))
    return worst


def test_oracle_adamw_clip_schedule_match_reference_trajectory():
    def step_fn(names, p, g, m, v, t, lr, b1, b2, wd):
        out = [oo.adamw_step(p[i], g[i], m[i], v[i], t, lr, b1, b2, 1e-6, 0.0 if oo.no_decay(names[i]) else wd)
               for i in range(len(p))]
        return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]

    # float32 throughout; torch's CPU kernels and numpy may differ by an ulp of the update (1e-4 * 1e-3)
    assert replay(step_fn) <= 2e-9


def test_schedule_and_grouping_known_answers():
    assert oo.lr_at(0, 1e-4, 3, 20) == 1e-8                  # rate floor 1e-5 (sched.py:112), then the 1e-8 lr floor
    assert abs(oo.lr_at(3, 1e-4, 3, 20) - 1e-4) < 1e-18     # end of warm-up
    assert oo.lr_at(20, 1e-2, 3, 20) == 1e-2 * 1e-5          # cosine reaches 0 -> rate floor
    assert oo.no_decay("ptv3_model.enc.enc0.block0.attn.qkv.bias")
    assert not oo.no_decay("ptv3_model.enc.enc0.block0.attn.qkv.weight")
    assert not oo.no_decay("ptv3_model.enc.enc0.block0.norm1.0.weight")  # only literal 'LayerNorm.weight' matches

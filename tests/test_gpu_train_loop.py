"""End-to-end training loop on the device (tiny config): the loss falls under the reference's optimiser recipe in every
operand-precision mode, and a run resumed from a reference-format checkpoint continues bit-identically."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

TOPTS = SimpleNamespace(learning_rate=1e-3, weight_decay=0.05, optim="adamw", betas=[0.9, 0.98], lr_sched="cosine",
                        warmup_steps=5, num_train_steps=200, grad_norm=10.0)


def _setup(seed=0):
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import config as lcfg, optim as loptim, synth
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA

    torch.manual_seed(seed)
    m = SimplePolicyPTV3CA(lcfg.preset("tiny")).cuda().train()
    m.ptv3_model.order_perms = [[0, 1, 2, 3], [2, 3, 0, 1]]
    m.ptv3_model.proj_drop = m.ptv3_model.attn_drop = 0.0
    m.act_proj_head.dropout = 0.0
    opt, init_lrs = loptim.build_optimizer(m, TOPTS)
    b = synth.synth_batch(4, 512, ragged=True, seed=5)
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if k == "disc_pos_probs" else v))
             for k, v in b.items()}
    return m, opt, init_lrs, batch, loptim


def _train(m, opt, init_lrs, batch, loptim, start, n):
    out = []
    for step in range(start, start + n):
        opt.zero_grad(set_to_none=True)
        _, losses = m(dict(batch), compute_loss=True, compute_final_action=False)
        losses["total"].backward()
        loptim.set_lr(opt, init_lrs, step, TOPTS)
        opt.clip_grad_norm_(TOPTS.grad_norm)
        opt.step()
        out.append(losses["total"].item())
    return out


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16"])
def test_loss_falls_in_every_precision_mode(mode):
    from robot_3dlotus_amd import ops

    m, opt, init_lrs, batch, loptim = _setup()
    ops.set_gemm_precision(mode)
    try:
        hist = _train(m, opt, init_lrs, batch, loptim, 0, 40)
    finally:
        ops.set_gemm_precision("fp32")
    assert all(torch.isfinite(torch.tensor(hist))), hist
    assert hist[-1] < 0.7 * hist[0], (mode, hist[0], hist[-1])


def test_resume_from_reference_format_checkpoint(tmp_path):
    from robot_3dlotus_amd import checkpoint as ck

    m, opt, init_lrs, batch, loptim = _setup()
    _train(m, opt, init_lrs, batch, loptim, 0, 5)
    ck.ModelSaver(str(tmp_path)).save(m, 5, optimizer=opt, rewrite_optimizer=True)
    cont = _train(m, opt, init_lrs, batch, loptim, 5, 3)
    m2, opt2, init2, batch2, _ = _setup(seed=123)                     # different init: everything must come from the files
    mfile, ock, step = ck.find_resume_state(str(tmp_path), True)
    assert step == 5 and ck.load_model_checkpoint(m2, mfile, strict=True)[0] == len(m.state_dict())
    opt2.load_state_dict(ock["optimizer"])
    resumed = _train(m2, opt2, init2, batch2, loptim, step, 3)
    assert resumed == cont, (resumed, cont)

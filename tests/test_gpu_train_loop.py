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


def test_episode_reader_to_training_step(tmp_path):
    """The whole input path on real-format records: episode store -> KeystepDataset (table / robot-box removal, z rotation
    + jitter, centring) -> DataLoader with the collate function (pinned, workers) -> prefetch()/forward/backward with the
    optimiser.  The soft position labels built on the device from gt_actions give the same loss as the host-built
    `disc_pos_probs` of the reference dataset (1e-5), for both heatmap types."""
    import json
    import random
    import numpy as np
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import config as lcfg, data as ld, dataset as ds, optim as loptim
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA

    rng = np.random.default_rng(0)
    store = ds.DirStore(str(tmp_path / "eps"))
    tv = "open_drawer+1"
    for e in range(4):
        store.write(tv, f"episode{e}".encode(), ds.synth_episode(rng, steps=4, points=1500))
    instrs = {tv: ["open the drawer"]}
    (tmp_path / "i.json").write_text(json.dumps(instrs))
    np.save(tmp_path / "e.npy", {"open the drawer": rng.standard_normal((7, 512)).astype(np.float32)}, allow_pickle=True)
    kw = dict(num_points=800, xyz_shift="center", xyz_norm=False, use_height=True, instr_embed_type="all", rm_robot="box_keep_gripper",
              augment_pc=True, aug_max_rot=180, pos_bins=15, pos_bin_size=0.01, store=store)
    torch.manual_seed(0)
    m = SimplePolicyPTV3CA(lcfg.preset("tiny")).cuda().train()
    m.ptv3_model.proj_drop = m.ptv3_model.attn_drop = 0.0
    m.act_proj_head.dropout = 0.0
    for kind in ("plain", "dist"):
        host = ds.KeystepDataset(None, str(tmp_path / "e.npy"), str(tmp_path / "i.json"), host_labels=True, pos_heatmap_type=kind, **kw)
        devl = ds.KeystepDataset(None, str(tmp_path / "e.npy"), str(tmp_path / "i.json"), pos_heatmap_type=kind, **kw)
        random.seed(3); np.random.seed(3)
        a = ld.ptv3_collate_fn([host[0], host[1]])
        random.seed(3); np.random.seed(3)
        b = ld.ptv3_collate_fn([devl[0], devl[1]])
        b["pos_heatmap_type"] = kind
        assert "disc_pos_probs" in a and "disc_pos_probs" not in b and torch.equal(a["pc_fts"], b["pc_fts"])
        m.ptv3_model.order_perms = [[0, 1, 2, 3], [2, 3, 0, 1]]
        _, la = m(a, compute_loss=True, compute_final_action=False)
        _, lb = m(b, compute_loss=True, compute_final_action=False)
        assert abs(la["pos"].item() - lb["pos"].item()) < 1e-5 * max(1.0, abs(la["pos"].item())), (kind, la["pos"].item(), lb["pos"].item())
    # a short training run fed by a DataLoader
    m.ptv3_model.order_perms = None
    opt, init_lrs = loptim.build_optimizer(m, TOPTS)
    loader = torch.utils.data.DataLoader(devl, batch_size=2, shuffle=True, num_workers=2, collate_fn=lambda x: ld.ptv3_collate_fn(x, pin=False))
    losses = []
    for epoch in range(3):
        for step, batch in enumerate(loader):
            opt.zero_grad(set_to_none=True)
            _, l = m(batch, compute_loss=True, compute_final_action=False)
            l["total"].backward()
            loptim.set_lr(opt, init_lrs, len(losses) + 1, TOPTS)
            opt.clip_grad_norm_(TOPTS.grad_norm)
            opt.step()
            losses.append(l["total"].item())
    assert all(np.isfinite(losses)) and np.mean(losses[-2:]) < np.mean(losses[:2]), losses


def test_checkpoint_validator_self_test():
    """tools/validate_checkpoint.py --self-test: strict load of a (synthetic) checkpoint, eval-mode inference on stored key
    steps through the episode reader, action errors under the assumed spconv weight layout and the two alternatives."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "validate_checkpoint.py"), "--self-test"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout[r.stdout.index("{"):])
    assert set(rep["interpretations"]) == {"as_assumed", "taps_z_major", "cin_cout_swapped"}
    for v in rep["interpretations"].values():
        assert v["n"] > 0 and all(k in v for k in ("pos_err_m", "rot_err_deg", "open_acc"))
    assert any(len(s) == 5 for s in rep["conv_weight_shapes"].values())

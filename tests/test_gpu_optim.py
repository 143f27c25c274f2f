"""GPU parity of the fused optimiser step (robot_3dlotus_amd.optim.AdamW: multi-tensor HIP AdamW + gradient-norm clip)
against (a) the golden trajectory captured from the imported reference trainer pieces and (b) the numpy oracle on
ragged tensor sizes.  Tolerance: 2e-7 relative to max|p| (one float32 ulp of the accumulated update; the kernel may
contract a*b+c into fma where torch / numpy round twice)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim_traj.npz")


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = torch.nn.Linear(5, 7)
        self.LayerNorm = torch.nn.LayerNorm(7)
        self.fc2 = torch.nn.Linear(7, 3)


def test_golden_trajectory_of_the_reference_trainer():
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import optim as lo

    fx = np.load(GOLD)
    lr0, wd, b1, b2, warm, total, max_norm = fx["hyper"].tolist()
    net = Toy().cuda()
    assert [n for n, _ in net.named_parameters()] == [str(n) for n in fx["names"]]
    sizes = fx["sizes"].tolist()
    off = np.concatenate([[0], np.cumsum(sizes)])
    with torch.no_grad():
        for i, p in enumerate(net.parameters()):
            p.copy_(torch.from_numpy(fx["p0"][off[i]:off[i + 1]]).view_as(p))
    opts = SimpleNamespace(learning_rate=lr0, weight_decay=wd, optim="adamw", betas=[b1, b2], lr_sched="cosine",
                           warmup_steps=int(warm), num_train_steps=int(total))
    opt, init_lrs = lo.build_optimizer(net, opts)
    assert [len(g["params"]) for g in opt.param_groups] == [2, 4]  # fc weights decay; biases + LayerNorm.* do not
    for step in range(6):
        lr = lo.set_lr(opt, init_lrs, step, opts)
        assert abs(lr - float(fx[f"lr{step}"])) <= 1e-12 * max(lr, 1.0)
        for i, p in enumerate(net.parameters()):
            p.grad = torch.from_numpy(fx[f"g{step}"][off[i]:off[i + 1]]).view_as(p).cuda()
        norm = opt.clip_grad_norm_(max_norm)
        opt.step()
        assert abs(norm.item() - float(fx[f"norm{step}"])) <= 2e-6 * float(fx[f"norm{step}"])
        got = torch.cat([p.detach().flatten() for p in net.parameters()]).cpu().numpy()
        ref = fx[f"p{step + 1}"]
        assert np.abs(got - ref).max() <= 2e-7 * np.abs(ref).max(), (step, np.abs(got - ref).max())
    sd = opt.state_dict()["state"]
    assert set(sd[0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and sd[0]["step"] == 6


@pytest.mark.parametrize("clip", [None, 0.5])
def test_ragged_multi_tensor_step_matches_oracle(clip):
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import optim as lo
    from oracle import optim as oo

    g = torch.Generator().manual_seed(5)
    shapes = [(1,), (3,), (4097,), (70001,), (128, 128), (64, 5, 5, 5, 7), (5,)]
    wds = [0.0, 0.05, 0.05, 0.0, 0.05, 0.05, 0.05]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    opt = lo.AdamW([{"params": [p], "weight_decay": w} for p, w in zip(ps, wds)], lr=3e-4, betas=(0.9, 0.98))
    ref_p = [p.detach().cpu().numpy().ravel().copy() for p in ps]
    ref_m = [np.zeros_like(x) for x in ref_p]
    ref_v = [np.zeros_like(x) for x in ref_p]
    steps = [0] * len(ps)
    for it in range(3):
        grads = [torch.randn(s, generator=g) * (3.0 if it == 1 else 0.1) for s in shapes]
        for i, p in enumerate(ps):
            p.grad = None if (i == 6 and it == 0) else grads[i].cuda()  # a parameter without gradient is skipped
        live = [i for i, p in enumerate(ps) if p.grad is not None]
        gl = [grads[i].numpy().ravel() for i in live]
        if clip is not None:
            norm = opt.clip_grad_norm_(clip)
            rn, gl = oo.clip_grad_norm(gl, clip)
            assert abs(norm.item() - rn) <= 2e-6 * rn
        opt.step()
        for k, i in enumerate(live):
            steps[i] += 1
            ref_p[i], ref_m[i], ref_v[i] = oo.adamw_step(ref_p[i], gl[k], ref_m[i], ref_v[i], steps[i], 3e-4, 0.9, 0.98, 1e-6, wds[i])
        for i, p in enumerate(ps):
            got = p.detach().cpu().numpy().ravel()
            assert np.abs(got - ref_p[i]).max() <= 2e-7 * max(1.0, np.abs(ref_p[i]).max()), (it, i)
            assert np.abs(opt.state[p]["exp_avg_sq"].cpu().numpy().ravel() - ref_v[i]).max() <= 1e-6 * max(1e-6, np.abs(ref_v[i]).max())


def test_usage_mask_skips_what_no_rank_used_in_the_step_it_flips():
    """lotus_adamw_step(..., used, used_idx): a tensor whose usage flag is 0 is skipped ON THE DEVICE — parameter, moments and
    (host side, one step late) its step counter untouched — exactly like a parameter without a gradient on one GPU
    (adamw.py:67-68).  The reducer stand-in delivers the flags to the host one step late, as parallel.GradReducer does."""
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import optim as lo

    torch.manual_seed(0)
    dev = torch.device("cuda")

    def make():
        torch.manual_seed(1)
        return [torch.nn.Parameter(torch.randn(n, device=dev)) for n in (5000, 37, 4096, 12)]

    class FakeReducer:  # the protocol AdamW.step(reducer=...) uses
        def __init__(self, params):
            self.params, self.step_id, self.used_mask = params, 0, None
            self.flat = [torch.zeros_like(p) for p in params]
            self._late = {}

        def finish(self, grads, used):
            self.step_id += 1
            for v, g in zip(self.flat, grads):
                v.copy_(g)
            self.used_mask = torch.tensor(used, dtype=torch.int32, device=dev)
            self._late[self.step_id] = {i for i, u in enumerate(used) if not u}

        def unused_of(self, k):  # flags of step k are known to the host from step k + 1 on
            return self._late.get(k) if k < self.step_id else None

        def view_of(self, p):
            return self.flat[[id(q) for q in self.params].index(id(p))]

    pa, pb = make(), make()
    # the optimiser's parameter order differs from the reducer's (used_idx maps one onto the other)
    oa = lo.AdamW([{"params": [pa[2], pa[0]], "weight_decay": 0.05}, {"params": [pa[3], pa[1]], "weight_decay": 0.0}], lr=1e-2, betas=(0.9, 0.98))
    ob = lo.AdamW([{"params": [pb[2], pb[0]], "weight_decay": 0.05}, {"params": [pb[3], pb[1]], "weight_decay": 0.0}], lr=1e-2, betas=(0.9, 0.98))
    red = FakeReducer(pa)
    plans = [[1, 1, 1, 1], [1, 0, 1, 1], [1, 0, 0, 1], [1, 1, 1, 1], [1, 1, 0, 1]]
    for step, used in enumerate(plans):
        grads = [torch.randn_like(p) for p in pa]
        red.finish([g if u else torch.zeros_like(g) for g, u in zip(grads, used)], used)
        for p in pa:
            p.grad = None                      # the fused step must not depend on .grad in reducer mode
        oa.step(reducer=red)
        for p, g, u in zip(pb, grads, used):   # one-GPU semantics: no gradient -> skipped
            p.grad = g.clone() if u else None
        ob.step()
        for i, (x, y) in enumerate(zip(pa, pb)):
            assert torch.equal(x.detach(), y.detach()), (step, i)
            assert torch.equal(oa.state[x]["exp_avg_sq"], ob.state[y]["exp_avg_sq"]), (step, i)
    oa._usage(red)  # (what the next step would do first: retract the counters of the last step's skipped tensors)
    assert [oa.state[p]["step"] - (1 if k == 2 else 0) for k, p in enumerate(pa)] == [ob.state[p]["step"] for p in pb] == [5, 3, 3, 5]

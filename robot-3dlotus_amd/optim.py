"""Optimiser side of the training step, mirroring the reference's interface:

  AdamW(params, lr, betas, eps, weight_decay, correct_bias)   genrobo3d/train/optim/adamw.py:13-112
  build_optimizer(model, opts) -> (optimizer, init_lrs)        genrobo3d/train/optim/misc.py:13-55
  get_lr_sched_decay_rate(global_step, opts)                   genrobo3d/train/optim/sched.py:95-113
  AdamW.clip_grad_norm_(max_norm)                              torch.nn.utils.clip_grad_norm_ (train_simple_policy.py:237)

Same hyper-parameters, `param_groups` / `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter) and
numerics as the reference class, but `step()` is ONE multi-tensor HIP launch over all parameters (plus two for the
gradient norm) instead of ~8 ATen kernels per tensor from a Python loop.  The clip coefficient stays on the device and
is folded into the update, so clipping costs no extra pass over the gradients and no host synchronisation."""
import math

import numpy as np
import torch

from ._capi import call, lib


def warmup_cosine(step, warmup_step, tot_step, num_cycles=0.5):
    if step < warmup_step:
        return step / warmup_step
    progress = float(step - warmup_step) / float(max(1, tot_step - warmup_step))
    return 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress))


def warmup_linear(step, warmup_step, tot_step):
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def get_lr_sched_decay_rate(global_step, opts):
    """sched.py:95-113 for the schedules the published configs select ('cosine', 'linear')."""
    sched = getattr(opts, "lr_sched", "cosine")
    if sched == "cosine":
        rate = warmup_cosine(global_step, opts.warmup_steps, opts.num_train_steps)
    elif sched == "linear":
        rate = warmup_linear(global_step, opts.warmup_steps, opts.num_train_steps)
    else:
        raise NotImplementedError(f"lr schedule {sched!r} is not built")
    return max(rate, 1e-5)


def set_lr(optimizer, init_lrs, global_step, opts):
    """train_simple_policy.py:227-229"""
    rate = get_lr_sched_decay_rate(global_step, opts)
    lr = None
    for kp, group in enumerate(optimizer.param_groups):
        group["lr"] = lr = max(init_lrs[kp] * rate, 1e-8)
    return lr


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        # same admissible ranges as the reference class (adamw.py:40-47)
        b1, b2 = (float(b) for b in betas)
        for name, value, ok in (("learning rate", lr, lr >= 0.0), ("beta1", b1, 0.0 <= b1 < 1.0),
                                ("beta2", b2, 0.0 <= b2 < 1.0), ("epsilon", eps, eps >= 0.0)):
            if not ok:
                raise ValueError(f"AdamW: {name} = {value} is outside its admissible range")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        super().__init__(params, defaults)
        self._tables = None
        self._clip = None  # device [norm, coefficient] of the pending clip_grad_norm_
        self._pending_usage = []  # reducer step ids counted before their usage flags reached the host (_usage)

    # ---- device tables -------------------------------------------------------------------------------------------
    def _build_tables(self):
        plist = [(p, g) for g in self.param_groups for p in g["params"]]
        dev = plist[0][0].device
        if dev.type != "cuda":
            raise RuntimeError("lotus-hip optimiser runs on a HIP device only (no CPU fallback)")
        for p, _ in plist:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
            assert p.data.is_contiguous() and p.dtype == torch.float32, "fp32 contiguous parameters only"
        T = len(plist)
        chunk = lib().fn["lotus_mt_chunk"]()
        numel = np.array([p.numel() for p, _ in plist], dtype=np.int64)
        ch = [(t, c) for t in range(T) for c in range((int(numel[t]) + chunk - 1) // chunk)]
        ptr = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=dev)
        tb = dict(plist=plist, T=T, dev=dev, nchunks=len(ch),
                  numel=torch.from_numpy(numel).to(dev), chunks=torch.tensor(ch, dtype=torch.int32, device=dev).reshape(-1, 2),
                  p=ptr([p.data for p, _ in plist]), m=ptr([self.state[p]["exp_avg"] for p, _ in plist]),
                  v=ptr([self.state[p]["exp_avg_sq"] for p, _ in plist]),
                  ptrs=[(p.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p, _ in plist],
                  partial=torch.empty(max(len(ch), 1), dtype=torch.float64, device=dev))
        # bf16 weight shadows (ops.WeightShadows, bf16-storage mode): the step rewrites them together with their masters
        sh = [getattr(p, "_lotus_b16", None) for p, _ in plist]
        tb["shadow_ptrs"] = [0 if t is None else t.data_ptr() for t in sh]
        tb["shadow"] = torch.tensor(tb["shadow_ptrs"], dtype=torch.int64, device=dev) if any(tb["shadow_ptrs"]) else None
        self._tables = tb

    def _tables_ok(self):
        tb = self._tables
        if tb is None:
            return False
        plist = [(p, g) for g in self.param_groups for p in g["params"]]
        if len(plist) != tb["T"]:
            return False
        if not all(len(self.state[p]) and (p.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) == q
                   for (p, _), q in zip(plist, tb["ptrs"])):
            return False
        # a shadow attached (or moved) after the tables were built must be picked up: the step changes the master through a
        # raw pointer, i.e. without bumping the version counter WeightShadows.refresh() looks at
        for (p, _), q in zip(plist, tb["shadow_ptrs"]):
            t = getattr(p, "_lotus_b16", None)
            if (0 if t is None else t.data_ptr()) != q:
                return False
        return True

    def _usage(self, reducer):
        """Data-parallel step: which parameters are skipped is decided ON THE DEVICE by the reducer's exact usage mask; the host
        only keeps the per-parameter step counters (bias correction) in line with it.  The flags of a step reach the host one
        step late (parallel.GradReducer._sync_usage): a step is first counted for every parameter, and retracted for the
        parameters nobody used once their flags are known — always before the counter is read again, because a parameter
        skipped on the device does not read its step size.  -> (device mask, device index table, indices known unused NOW)"""
        tb = self._tables
        if tb.get("red") is not reducer:
            idx = {p: i for i, p in enumerate(reducer.params)}
            missing = [1 for p, _ in tb["plist"] if p not in idx]
            if missing:
                raise ValueError(f"AdamW.step(reducer=...): {len(missing)} optimiser parameters are not managed by the reducer")
            tb["red"], tb["red_index"] = reducer, [idx[p] for p, _ in tb["plist"]]
            tb["red_idx_dev"] = torch.tensor(tb["red_index"], dtype=torch.int32, device=tb["dev"])
            self._pending_usage = []
        k = reducer.step_id
        by_red = {r: p for r, (p, _) in zip(tb["red_index"], tb["plist"])}
        for j in list(self._pending_usage):   # steps whose flags had not arrived when they were counted
            un = reducer.unused_of(j)
            if un is not None:
                for r in un:
                    if r in by_red:
                        self.state[by_red[r]]["step"] -= 1
                self._pending_usage.remove(j)
        now = reducer.unused_of(k)
        if now is None:
            self._pending_usage.append(k)
        return reducer.used_mask, tb["red_idx_dev"], (now or set())

    def _grad_table(self, reducer=None, known_unused=()):
        """Per step: gradient pointers (fresh tensors every backward), step sizes and decays -> one pinned upload."""
        tb = self._tables
        T = tb["T"]
        host = torch.empty(3 * T, dtype=torch.int64, pin_memory=True)  # (caching host allocator: safe to reuse per step)
        hp = host.numpy()
        fl = hp[T:3 * T].view(np.float32)  # 4 T floats available; [0:T] step_size, [T:2T] decay
        for i, (p, group) in enumerate(tb["plist"]):
            if reducer is not None:
                # the slice of the flat, rank-averaged buffer — also for a parameter whose `.grad` the reducer has reset to
                # None on one-step-old knowledge: the device mask decides, from THIS step's flags
                g = None if tb["red_index"][i] in known_unused else reducer.view_of(p)
            else:
                g = p.grad
            if g is None:
                hp[i] = 0
                continue
            if g.is_sparse:
                raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
            if not g.is_contiguous():
                g = p.grad = g.contiguous()
            hp[i] = g.data_ptr()
            st = self.state[p]
            st["step"] += 1
            beta1, beta2 = group["betas"]
            ss = group["lr"]
            if group["correct_bias"]:
                ss = ss * math.sqrt(1.0 - beta2 ** st["step"]) / (1.0 - beta1 ** st["step"])
            fl[i] = ss
            fl[T + i] = group["lr"] * group["weight_decay"] if group["weight_decay"] > 0.0 else 0.0
        return host.to(tb["dev"], non_blocking=True)

    # ---- public ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def clip_grad_norm_(self, max_norm):
        """Global L2 norm of all gradients + clip coefficient, on the device.  Returns the norm (0-dim tensor, no host
        sync); the gradients themselves are left untouched — the following step() applies the coefficient."""
        if not self._tables_ok():
            self._build_tables()
        tb = self._tables
        gp = torch.tensor([0 if p.grad is None else p.grad.data_ptr() for p, _ in tb["plist"]], dtype=torch.int64).pin_memory()
        gp = gp.to(tb["dev"], non_blocking=True)
        out = torch.empty(2, dtype=torch.float32, device=tb["dev"])
        call("lotus_grad_norm", gp, tb["numel"], tb["chunks"], tb["nchunks"], tb["partial"], out, float(max_norm))
        self._clip = out
        return out[0]

    @torch.no_grad()
    def step(self, closure=None, reducer=None):
        """reducer: the parallel.GradReducer of a data-parallel step (after its finish()): gradients are read from its flat
        buffer and parameters NO rank used in this step are skipped on the device (exact in the step their usage changes)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._tables_ok():
            self._build_tables()
        tb = self._tables
        T = tb["T"]
        used = used_idx = None
        known = ()
        if reducer is not None and reducer.used_mask is not None:
            used, used_idx, known = self._usage(reducer)
        else:
            reducer = None
        dv = self._grad_table(reducer, known)
        fl = dv[T:3 * T].view(torch.float32)
        g0 = self.param_groups[0]
        for g in self.param_groups:
            assert tuple(g["betas"]) == tuple(g0["betas"]) and g["eps"] == g0["eps"], "per-group betas / eps are not built"
        clip = self._clip[1:2] if self._clip is not None else None
        call("lotus_adamw_step", tb["p"], dv[:T], tb["m"], tb["v"], tb["numel"], fl[:T], fl[T:2 * T], tb["chunks"], tb["nchunks"],
             float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]), clip, tb["shadow"], used, used_idx)
        self._clip = None
        return loss


def build_optimizer(model, opts):
    """Parameter groups of misc.py:13-55 for `optim: 'adamw'`: per family (`rgb_encoder` parameters with their learning-rate
    multiplier, everything else) one group with weight decay and one without.  A parameter is exempt from decay when its
    NAME contains 'bias', 'LayerNorm.bias' or 'LayerNorm.weight' — in this model that is every `*.bias` and nothing
    else, because the norm layers are called norm1.0 / cpe.2.  -> (optimizer, init_lrs)"""
    if getattr(opts, "optim", "adamw") != "adamw":
        raise NotImplementedError("only the optimiser the published configs select ('adamw') is built")
    exempt_tags = ("bias", "LayerNorm.bias", "LayerNorm.weight")
    families = {"rgb": [], "others": []}
    for name, p in model.named_parameters():
        if p.requires_grad:
            families["rgb" if "rgb_encoder" in name else "others"].append((name, p))
    groups, init_lrs = [], []
    for fam, members in families.items():
        if not members:
            continue
        lr = opts.learning_rate * (getattr(opts, "rgb_encoder_lr_multi", 1) if fam == "rgb" else 1)
        exempt = [any(tag in name for tag in exempt_tags) for name, _ in members]
        for want_exempt, wd in ((False, opts.weight_decay), (True, 0.0)):
            groups.append({"params": [p for (_, p), e in zip(members, exempt) if e == want_exempt], "weight_decay": wd, "lr": lr})
            init_lrs.append(lr)
    return AdamW(groups, lr=opts.learning_rate, betas=opts.betas), init_lrs

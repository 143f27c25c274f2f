"""CPU tests of the host-side logic: config presets, state_dict layout, synthetic batches,
cloud sharding and the data-parallel gradient reducer (world_size 2 over gloo)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as gu
import robot_3dlotus_amd  # noqa: F401
from oracle import front_end as fe
from robot_3dlotus_amd import config as lcfg, parallel, synth
from robot_3dlotus_amd.policy import MODEL_FACTORY, SimplePolicyPTV3CA


def test_v1_preset_matches_published_overrides():
    c = lcfg.preset("v1")
    assert c.model_class == "SimplePolicyPTV3CA" and MODEL_FACTORY[c.model_class] is SimplePolicyPTV3CA
    p, a = c.ptv3_config, c.action_config
    assert p.enc_channels == [64, 128, 256, 512, 768] and p.dec_channels == [128, 128, 256, 512]
    assert p.enc_depths == [1] * 5 and p.qk_norm is True and p.enable_flash is True and p.in_channels == 7
    assert a.pos_pred_type == "heatmap_disc" and a.rot_pred_type == "euler_disc" and a.pos_bins == 15
    assert a.dropout == 0.2 and p.proj_drop == 0.1 and p.drop_path == 0.0 and a.use_ee_pose is False


def test_override_parser_follows_yacs_list_semantics():
    c = lcfg.load_model_config(None, ["MODEL.ptv3_config.enc_depths", "[1, 1]", "action_config.pos_bins", "7",
                                      "ptv3_config.pdnorm_conditions", "null", "action_config.reduce", "max"])
    assert c.ptv3_config.enc_depths == [1, 1] and c.action_config.pos_bins == 7
    assert c.ptv3_config.pdnorm_conditions is None and c.action_config.reduce == "max"


@pytest.mark.parametrize("variant,nparams,nentries", [("v1", 68177587, 460), ("tiny", 947827, 157)])
def test_module_state_dict_layout(variant, nparams, nentries):
    cfg = lcfg.preset(variant)
    m = SimplePolicyPTV3CA(cfg)
    sd, t = m.state_dict(), gu.state_template(cfg)
    assert set(sd) == set(t) and len(sd) == nentries
    assert all(tuple(sd[k].shape) == tuple(t[k].shape) for k in t)
    assert m.num_parameters[0] == nparams == m.num_trainable_parameters[0]
    # weight-decay grouping of the reference keys on these substrings (optim/misc.py:15)
    nodecay = [n for n, _ in m.named_parameters() if any(s in n for s in ("bias", "LayerNorm.bias", "LayerNorm.weight"))]
    assert all(n.endswith("bias") for n in nodecay)


def test_yaml_default_depths_and_drop_path_construct():
    """The reference YAML's stage depths (enc [2, 2, 2, 6, 2], dec [2, 2, 2, 2]) and drop_path 0.1 on the v1 widths: block{i} /
    ca_block{i} containers with the reference's state_dict keys, the stochastic-depth schedule of model_ca.py:250-252,316-325
    (linear over the encoder / decoder, decoder slices reversed), patch tables for min(depth, 4) curve slots."""
    c = lcfg.load_model_config(None, lcfg.V1_OVERRIDES + ["ptv3_config.enc_depths", "[2, 2, 2, 6, 2]", "ptv3_config.dec_depths",
                                                          "[2, 2, 2, 2]", "ptv3_config.drop_path", "0.1"])
    m = SimplePolicyPTV3CA(c)
    sd, t = m.state_dict(), gu.state_template(c)
    assert set(sd) == set(t) and all(tuple(sd[k].shape) == tuple(t[k].shape) for k in t)
    assert "ptv3_model.enc.enc3.block5.attn.qkv.weight" in sd and "ptv3_model.dec.dec0.ca_block1.attn.kv.weight" in sd
    p = m.ptv3_model
    flat = [x for r in p.enc_drop_path for x in r]
    assert len(flat) == 14 and flat[0] == 0.0 and abs(flat[-1] - 0.1) < 1e-7 and flat == sorted(flat)
    assert abs(p.dec_drop_path[3][0] - 0.1) < 1e-7 and p.dec_drop_path[0][1] == 0.0 and p.dec_drop_path[0][0] > 0
    assert p.frontend.n_patch_orders == 4
    m2 = SimplePolicyPTV3CA(lcfg.preset("tinydeep"))
    assert len(m2.state_dict()) == 397 and m2.ptv3_model.frontend.n_patch_orders == 4


def test_unsupported_configurations_raise():
    with pytest.raises(NotImplementedError):
        SimplePolicyPTV3CA(lcfg.load_model_config(None, lcfg.V1_OVERRIDES + ["ptv3_config.enable_flash", "False"]))


def test_context_token_options_build_the_reference_parameters():
    """use_ee_pose / use_step_id (simple_policy_ptv3.py:386-389): parameter names and shapes of base.py:52-60."""
    m = SimplePolicyPTV3CA(lcfg.preset("tinyctx"))
    sd, t = m.state_dict(), gu.state_template(lcfg.preset("tinyctx"))
    assert set(sd) == set(t) and all(tuple(sd[k].shape) == tuple(t[k].shape) for k in t)
    cc = lcfg.preset("tinyctx").action_config.context_channels
    assert sd["pose_embedding.rot_embedding.weight"].shape == (cc, 6) and sd["stepid_embedding.weight"].shape == (30, cc)
    assert m.pose_embedding.layer_norm.eps == 1e-12


def test_synthetic_batch_schema_and_voxel_uniqueness():
    b = synth.synth_batch(5, 1000, ragged=True, seed=3)
    n = b["npoints_in_batch"]
    assert b["pc_fts"].shape == (sum(n), 7) and b["offset"].tolist() == np.cumsum(n).tolist()
    assert b["txt_embeds"].shape == (sum(b["txt_lens"]), 512) and b["gt_actions"].shape == (5, 7)
    assert all(p.shape == (3, k * 30) for p, k in zip(b["disc_pos_probs"], n))
    g = fe.grid_coord(b["pc_fts"][:, :3].numpy())
    key = np.concatenate([fe.offset2batch(n)[:, None], g], 1)
    assert len(np.unique(key, axis=0)) == len(key), "synthetic clouds must stay voxel-unique after re-gridding"


def test_shard_clouds_balances_points():
    counts = [4096, 100, 3000, 2900, 50, 4000, 3500, 1200]
    sh = parallel.shard_clouds(counts, 4)
    assert sorted(i for s in sh for i in s) == list(range(8))
    loads = [sum(counts[i] for i in s) for s in sh]
    assert max(loads) - min(loads) < 2000


def _reducer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, _, w = parallel.init_distributed(backend="gloo")
    torch.manual_seed(rank)  # different init per rank -> the broadcast must equalise
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2))
    red = parallel.GradReducer(net, bucket_mb=0.0002)  # tiny buckets -> several async all-reduces
    assert len(red.buckets) >= 2
    gsum = None
    for step in range(2):
        red.zero_grad()
        g = torch.Generator().manual_seed(100 + rank + 10 * step)
        x = torch.randn(5, 8, generator=g)
        net(x).square().sum().backward()
        red.finish()
        gsum = torch.cat([p.grad.flatten() for p in net.parameters()]).clone()
    # reference: average of the per-rank gradients computed serially
    torch.manual_seed(0)
    ref_net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2))
    acc = None
    for rr in range(world):
        ref_net.zero_grad()
        g = torch.Generator().manual_seed(100 + rr + 10)
        ref_net(torch.randn(5, 8, generator=g)).square().sum().backward()
        v = torch.cat([p.grad.flatten() for p in ref_net.parameters()])
        acc = v if acc is None else acc + v
    ok = torch.allclose(gsum, acc / world, atol=1e-6)
    # the bucket order was re-learnt from the first backward: gradients of the LAST layer arrive first
    ok = ok and (not red._learning) and red._bparams[0][0] in set(net[3].parameters())
    ok = ok and sorted(id(p) for ps in red._bparams for p in ps) == sorted(id(p) for p in net.parameters())
    # ... and with the order known, one hook per bucket (on its last-arriving parameter) replaces the per-parameter hooks
    ok = ok and 1 <= len(red._hook_handles) <= len(red.buckets) < len(red.params)
    # a module with a parameter that never receives a gradient (DDP's find_unused_parameters case): the used
    # parameters are still averaged — in the learning pass (partial flush at finish()) and afterwards
    class Net2(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.unused, self.b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 1), torch.nn.Linear(4, 2)

        def forward(self, x):
            return self.b(torch.relu(self.a(x)))

    torch.manual_seed(0)
    n2 = Net2()
    red2 = parallel.GradReducer(n2, bucket_mb=10.0)   # one bucket holding used and unused parameters
    for step in range(2):
        red2.zero_grad()
        g = torch.Generator().manual_seed(500 + rank + 7 * step)
        n2(torch.randn(3, 4, generator=g)).square().sum().backward()
        red2.finish()
        loc = torch.cat([p.grad.flatten() for p in (n2.a.weight, n2.a.bias, n2.b.weight, n2.b.bias)])
        gathered = [torch.zeros_like(loc) for _ in range(world)]
        dist.all_gather(gathered, loc)
        ok = ok and all(torch.allclose(gathered[0], t_, atol=1e-7) for t_ in gathered)
        # nobody used it: .grad stays None as on one GPU / under DDP(find_unused_parameters=True) — the usage bitmap is
        # MAX-reduced in finish(), so the optimiser skips it on every rank
        ok = ok and n2.unused.weight.grad is None and n2.unused.bias.grad is None
    torch.manual_seed(0)
    ref2, acc2 = Net2(), None
    for rr in range(world):
        ref2.zero_grad()
        g = torch.Generator().manual_seed(500 + rr + 7)
        ref2(torch.randn(3, 4, generator=g)).square().sum().backward()
        v = torch.cat([p.grad.flatten() for p in (ref2.a.weight, ref2.a.bias, ref2.b.weight, ref2.b.bias)])
        acc2 = v if acc2 is None else acc2 + v
    ok = ok and torch.allclose(loc, acc2 / world, atol=1e-6)
    checks = {"average_unused": bool(ok)}
    # gradient accumulation: two micro-batches per step, the first under no_sync(); the result is the
    # rank-average of the per-rank SUM over micro-batches, step after step
    torch.manual_seed(0)
    n3 = Net2()
    red3 = parallel.GradReducer(n3, bucket_mb=0.00005)
    used = (n3.a.weight, n3.a.bias, n3.b.weight, n3.b.bias)
    okacc = True
    for step in range(3):
        red3.zero_grad()
        xs = [torch.randn(3, 4, generator=torch.Generator().manual_seed(900 + rank + 7 * step + 31 * mb)) for mb in range(2)]
        with red3.no_sync():
            n3(xs[0]).square().sum().backward()
        n3(xs[1]).square().sum().backward()
        red3.finish()
        got = torch.cat([p.grad.flatten() for p in used]).clone()
        accr = None
        for rr in range(world):
            for mb in range(2):
                ref2.zero_grad()
                x = torch.randn(3, 4, generator=torch.Generator().manual_seed(900 + rr + 7 * step + 31 * mb))
                ref2(x).square().sum().backward()
                v = torch.cat([p.grad.flatten() for p in (ref2.a.weight, ref2.a.bias, ref2.b.weight, ref2.b.bias)])
                accr = v if accr is None else accr + v
        okacc = okacc and torch.allclose(got, accr / world, atol=1e-6)
    checks["accumulation"] = bool(okacc)
    # ... and a second backward without zero_grad() / no_sync() must raise instead of adding onto averaged gradients
    try:
        n3(xs[0]).square().sum().backward()
        checks["misuse_raises"] = False
    except RuntimeError as e:
        checks["misuse_raises"] = "no_sync" in str(e)
    red3.zero_grad()

    # a conditionally used module: unused in the learning pass, later used on ONE rank only -> no KeyError,
    # every rank ends with the same averaged gradient for it
    class Net3(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.opt, self.b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 2)

        def forward(self, x, use):
            h = torch.relu(self.a(x))
            if use:
                h = h + self.opt(h)
            return self.b(h)

    torch.manual_seed(0)
    n4 = Net3()
    red4 = parallel.GradReducer(n4, bucket_mb=0.00005)
    okc = True
    for step, use in enumerate([False, rank == 0, True]):
        red4.zero_grad()
        x = torch.randn(3, 4, generator=torch.Generator().manual_seed(40 + rank + step))
        n4(x, use).square().sum().backward()
        red4.finish()
        if step == 0:  # no rank used the optional module: no gradient for it anywhere
            okc = okc and n4.opt.weight.grad is None and n4.opt.bias.grad is None
        loc = torch.cat([p.grad.flatten() for p in n4.parameters() if p.grad is not None])
        gathered = [torch.zeros_like(loc) for _ in range(world)]
        dist.all_gather(gathered, loc)
        okc = okc and all(torch.equal(gathered[0], t_) for t_ in gathered)
        if step == 1:
            okc = okc and float(n4.opt.weight.grad.abs().max()) > 0.0  # rank 0's contribution / world
        # the usage mask the fused optimiser consumes (lotus_adamw_step `used`) is exact in the step the usage flips in, on
        # every rank: 0 for the optional module while nobody runs it, 1 from the step ONE rank does
        flags = red4.used_mask.tolist()
        opt_idx = [red4._index[q] for q in n4.opt.parameters()]
        okc = okc and [flags[i] for i in opt_idx] == [0 if step == 0 else 1] * 2 and sum(flags) == len(flags) - (2 if step == 0 else 0)
        okc = okc and red4.unused_of(red4.step_id) == (set(opt_idx) if step == 0 else set())
        okc = okc and all(torch.equal(red4.view_of(q), q.grad) for q in n4.parameters() if q.grad is not None)
    checks["conditional_usage"] = bool(okc)

    # gradient accumulation while the arrival order CHANGES: step 1 runs b(a(x)) and teaches the order, step 2 runs
    # a(b(x)) under no_sync() plus one syncing backward.  With one hook per bucket on its last-arriving parameter the bucket
    # would be packed when `a` arrives — every `.grad` exists after the local micro-batch — with b's STALE partial sum.
    class Net5(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)

        def forward(self, x, flip):
            return self.a(torch.relu(self.b(x))) if flip else self.b(torch.relu(self.a(x)))

    torch.manual_seed(0)
    n5, r5 = Net5(), Net5()
    r5.load_state_dict(n5.state_dict())
    red5 = parallel.GradReducer(n5, bucket_mb=10.0)   # one bucket
    xin = lambda rr, k: torch.randn(3, 4, generator=torch.Generator().manual_seed(700 + rr + 13 * k))  # noqa: E731
    red5.zero_grad()
    n5(xin(rank, 0), False).square().sum().backward()
    red5.finish()
    red5.zero_grad()
    sparse_before = len(red5._hook_handles) < len(red5.params)
    with red5.no_sync():
        n5(xin(rank, 1), True).square().sum().backward()
    n5(xin(rank, 2), True).square().sum().backward()
    red5.finish()
    got5 = torch.cat([q.grad.flatten() for q in n5.parameters()])
    acc5 = None
    for rr in range(world):
        r5.zero_grad()
        for k in (1, 2):
            r5(xin(rr, k), True).square().sum().backward()
        v = torch.cat([q.grad.flatten() for q in r5.parameters()])
        acc5 = v if acc5 is None else acc5 + v
    checks["accumulation_with_a_changed_order"] = bool(sparse_before and torch.allclose(got5, acc5 / world, atol=1e-6)
                                                       and len(red5._hook_handles) == len(red5.params))
    ok = all(checks.values())
    # SyncBN statistics hook: (sum, sumsq, count) vector is summed in place
    parallel.enable_sync_batchnorm()
    from robot_3dlotus_amd import ops
    s = torch.tensor([1.0 + rank, 2.0, 10.0], dtype=torch.float64)
    ops.BnState.reduce(s)
    ok = ok and s.tolist() == [3.0, 4.0, 20.0]
    checks["syncbn_hook"] = s.tolist() == [3.0, 4.0, 20.0]
    q.put((rank, bool(ok) and checks["syncbn_hook"], checks))
    dist.barrier()
    dist.destroy_process_group()


def _run_reducer_workers(port):
    import queue

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    try:
        res = [q.get(timeout=120) for _ in ps]
    except queue.Empty:
        res = None
    for p in ps:
        p.join(timeout=5 if res is None else 60)
        if p.is_alive():
            p.kill()
    return res


def test_grad_reducer_world2_gloo():
    res = _run_reducer_workers(29500 + (os.getpid() % 2000))
    if res is None:  # stuck rendezvous (port in use on a shared box): one retry elsewhere
        res = _run_reducer_workers(33500 + (os.getpid() % 2000))
    assert res is not None, "gloo workers did not finish"
    assert all(r[1] for r in res), res


def test_survives_convert_sync_batchnorm():
    """train_simple_policy.py:116-117 converts the model when world_size > 1: the BatchNorm containers become
    SyncBatchNorm (not a BatchNorm1d subclass); parameter names, the state_dict and the counter bookkeeping must not
    notice."""
    from robot_3dlotus_amd import config as lcfg
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA

    m = SimplePolicyPTV3CA(lcfg.preset("tiny"))
    keys = list(m.state_dict().keys())
    n_bn = len(m.ptv3_model._bn_counters())
    assert n_bn == sum(k.endswith("num_batches_tracked") for k in keys) > 0
    m2 = torch.nn.SyncBatchNorm.convert_sync_batchnorm(SimplePolicyPTV3CA(lcfg.preset("tiny")))
    assert list(m2.state_dict().keys()) == keys
    assert any(isinstance(x, torch.nn.SyncBatchNorm) for x in m2.modules())
    assert len(m2.ptv3_model._bn_counters()) == n_bn


def test_dropout_seed_streams():
    """Seeds of different ranks / steps / sites are unrelated 64-bit values (no small additive offsets)."""
    from robot_3dlotus_amd import ops

    seeds = {ops.mix_seed(ops.mix_seed(ops.mix_seed(1234, r), step), site) for r in range(4) for step in range(50) for site in range(12)}
    assert len(seeds) == 4 * 50 * 12
    lows = sorted(s & 0xFFFFFFFF for s in seeds)
    assert min(b - a for a, b in zip(lows, lows[1:])) > 0 and len({s >> 32 for s in seeds}) == len(seeds)


# ---------------------------------------------------------------------------------------------------------------------
# world 4, UNEQUAL shards: clouds assigned by parallel.shard_clouds (snake order over the point counts),
# every rank's loss — a mean over ITS clouds, like every loss of the model — scaled by parallel.shard_loss_scale, gradients
# averaged by the GradReducer: the result must be the gradient of the mean over ALL clouds computed by one process.
def _shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from robot_3dlotus_amd import parallel

    torch.manual_seed(0)
    torch.set_num_threads(1)
    r, _, w = parallel.init_distributed(backend="gloo")
    counts = [5, 17, 3, 9, 12, 7, 4, 21, 6, 11, 8]          # 11 clouds over 4 ranks: shards of 3 / 3 / 3 / 2 clouds
    g = torch.Generator().manual_seed(7)
    clouds = [torch.randn(n, 6, generator=g) for n in counts]
    tgts = [torch.randn(4, generator=g) for _ in counts]

    def net():
        torch.manual_seed(1)
        return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.GELU(), torch.nn.Linear(16, 4))

    def cloud_loss(m, i):  # per-cloud loss: features max-pooled over the cloud's points (as the policy head does)
        return (m(clouds[i]).max(0)[0] - tgts[i]).square().mean()

    shards = parallel.shard_clouds(counts, w)
    assert sorted(i for s_ in shards for i in s_) == list(range(len(counts))) and len({len(s_) for s_ in shards}) > 1
    mine = shards[r]
    m = net()
    red = parallel.GradReducer(m, bucket_mb=0.00005)
    ok = True
    for step in range(2):
        red.zero_grad()
        loss = sum(cloud_loss(m, i) for i in mine) / len(mine)
        (loss * parallel.shard_loss_scale(len(mine), len(counts), w)).backward()
        red.finish()
        got = torch.cat([p.grad.flatten() for p in m.parameters()])
        ref = net()
        (sum(cloud_loss(ref, i) for i in range(len(counts))) / len(counts)).backward()
        want = torch.cat([p.grad.flatten() for p in ref.parameters()])
        ok = ok and float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
        # and WITHOUT the scale the plain rank average is a different (shard-size weighted) gradient: the factor matters
        if step == 0:
            red.zero_grad()
            (sum(cloud_loss(m, i) for i in mine) / len(mine)).backward()
            red.finish()
            plain = torch.cat([p.grad.flatten() for p in m.parameters()])
            ok = ok and float((plain - want).abs().max()) > 1e-4 * float(want.abs().max())
    pts = [sum(counts[i] for i in s_) for s_ in shards]
    q.put((rank, bool(ok), {"points_per_rank": pts, "imbalance": max(pts) / (sum(pts) / w)}))
    dist.barrier()
    dist.destroy_process_group()


def test_unequal_shards_world4_gloo_match_the_single_process_gradient():
    import queue

    def run(port):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_shard_worker, args=(r, 4, port, q)) for r in range(4)]
        for p in ps:
            p.start()
        try:
            res = [q.get(timeout=120) for _ in ps]
        except queue.Empty:
            res = None
        for p in ps:
            p.join(timeout=5 if res is None else 60)
            if p.is_alive():
                p.kill()
        return res

    res = run(30500 + (os.getpid() % 2000)) or run(34500 + (os.getpid() % 2000))
    assert res is not None, "gloo workers did not finish"
    assert all(r[1] for r in res), res
    assert res[0][2]["imbalance"] < 1.25, res[0][2]   # snake assignment keeps the point counts within 25 % of the mean


def test_grad_reducer_tapers_the_tail_of_the_arrival_order():
    """The last buckets of the arrival order are small (cap/32, cap/8, cap/2 from the end): the final all-reduce of a step is the
    one nothing overlaps.  The buckets still tile the flat buffer in order and every parameter owns one view."""
    import torch
    from robot_3dlotus_amd import parallel

    torch.manual_seed(0)
    net = torch.nn.Sequential(*[torch.nn.Linear(64, 64, bias=(i % 3 == 0)) for i in range(40)])
    cap_mb = 64 * 64 * 4 * 8 / (1 << 20)  # eight weight matrices per full bucket
    red = parallel.GradReducer(net, bucket_mb=cap_mb)
    cap = red._cap
    sizes = [hi - lo for lo, hi in red.buckets]
    assert sum(sizes) == red.flat.numel() and all(b[0] == a[1] for a, b in zip(red.buckets, red.buckets[1:]))
    assert sizes[-1] <= max(cap // 32, 64 * 64) and sizes[-2] <= max(cap // 8, 64 * 64) and sizes[-3] <= cap // 2
    assert all(n >= cap for n in sizes[:-4])          # the head follows the >= cap rule (the bucket in front of the tail may be short)
    flat_order = [p for ps in red._bparams for p in ps]
    assert flat_order == list(reversed(red.params))   # registration order reversed until the arrival order is learnt
    for p, v in zip(flat_order, [v for vs in red._bviews for v in vs]):
        assert v.shape == p.shape and v.data_ptr() >= red.flat.data_ptr()


# ---------------------------------------------------------------------------------------------------------------------
# world 8: tapered buckets + unequal shards + a parameter whose usage flips from step to step, against the
# gradient ONE process computes for the mean over all clouds.
def _world8_worker(rank, world, port, q, defer=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from robot_3dlotus_amd import parallel

    torch.set_num_threads(1)
    r, _, w = parallel.init_distributed(backend="gloo")
    counts = [5, 17, 3, 9, 12, 7, 4, 21, 6, 11, 8, 13, 2, 10, 15, 6, 9, 4, 14]      # 19 clouds over 8 ranks: shards of 3 / 2 clouds
    g = torch.Generator().manual_seed(11)
    clouds = [torch.randn(n, 8, generator=g) for n in counts]
    tgts = [torch.randn(8, generator=g) for _ in counts]

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(3)
            self.body = torch.nn.Sequential(*[m for i in range(24) for m in (torch.nn.Linear(8, 8, bias=(i % 3 != 1)), torch.nn.Tanh())])
            self.opt = torch.nn.Linear(8, 8)

        def forward(self, x, use):
            h = self.body(x)
            return h + self.opt(h) if use else h

    def cloud_loss(m, i, use):
        return (m(clouds[i], use).max(0)[0] - tgts[i]).square().mean()

    shards = parallel.shard_clouds(counts, w)
    owner = {i: rr for rr, s_ in enumerate(shards) for i in s_}
    mine = shards[r]
    m = Net()
    cap_mb = 72 * 4 * 6 / (1 << 20)                       # six layers per full bucket -> several buckets + a tapered tail
    red = parallel.GradReducer(m, bucket_mb=cap_mb)
    red._defer_flush = defer  # (the one-communicator fallback of the RCCL lanes: every bucket leaves in finish())
    # who runs the optional module: nobody, rank 3 only, nobody again, everybody
    plans = [set(), {3}, set(), set(range(w))]
    ok, detail = True, {}
    for step, users in enumerate(plans):
        red.zero_grad()
        loss = sum(cloud_loss(m, i, r in users) for i in mine) / len(mine)
        (loss * parallel.shard_loss_scale(len(mine), len(counts), w)).backward()
        red.finish()
        ref = Net()
        (sum(cloud_loss(ref, i, owner[i] in users) for i in range(len(counts))) / len(counts)).backward()
        for (name, pr), pm in zip(ref.named_parameters(), m.parameters()):
            if pr.grad is None:            # nobody used it: no gradient on any rank, flagged unused in THIS step
                ok = ok and pm.grad is None and red.used_mask[red._index[pm]].item() == 0
            else:
                ok = ok and pm.grad is not None and red.used_mask[red._index[pm]].item() == 1
                ok = ok and float((pm.grad - pr.grad).abs().max()) <= 1e-5 * max(1e-6, float(pr.grad.abs().max()))
        if step == 1:                      # the layout in use after the order was learnt
            sizes = [hi - lo for lo, hi in red.buckets]
            detail["buckets"] = sizes
            ok = ok and len(sizes) >= 5 and sizes[-2] < sizes[0] and not red._learning
    q.put((rank, bool(ok), detail))
    dist.barrier()
    dist.destroy_process_group()


def test_world8_gloo_tapered_buckets_unequal_shards_and_a_flipping_parameter():
    import queue

    def run(port):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_world8_worker, args=(r, 8, port, q)) for r in range(8)]
        for p in ps:
            p.start()
        try:
            res = [q.get(timeout=240) for _ in ps]
        except queue.Empty:
            res = None
        for p in ps:
            p.join(timeout=5 if res is None else 60)
            if p.is_alive():
                p.kill()
        return res

    res = run(31500 + (os.getpid() % 2000)) or run(35500 + (os.getpid() % 2000))
    assert res is not None, "gloo workers did not finish"
    assert all(r[1] for r in res), res


def test_world4_gloo_buckets_deferred_to_finish_match_the_single_process_gradient():
    """The reducer's fallback when the communication stream cannot run independently of the training stream (one RCCL communicator
    for statistics and buckets: parallel.GradReducer._defer_flush): no bucket is sent from a hook, finish() sends them all in layout
    order — same gradients, usage flags and unused-parameter handling as the overlapped schedule (the world-8 plan on 4 ranks)."""
    import queue

    def run(port):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_world8_worker, args=(r, 4, port, q, True)) for r in range(4)]
        for p in ps:
            p.start()
        try:
            res = [q.get(timeout=240) for _ in ps]
        except queue.Empty:
            res = None
        for p in ps:
            p.join(timeout=5 if res is None else 60)
            if p.is_alive():
                p.kill()
        return res

    res = run(33500 + (os.getpid() % 2000)) or run(37500 + (os.getpid() % 2000))
    assert res is not None, "gloo workers did not finish"
    assert all(r[1] for r in res), res


def test_bench_watchdog_ends_a_rank_that_stopped_ticking():
    """bench.py --gpus N > 1: a rank stuck in a collective must not hold the node until the caller's limit — no tick within
    LOTUS_BENCH_WATCHDOG_S ends the process with exit code 17 and the Python stacks on stderr; a ticking rank lives."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "tick = bench._watchdog(3, 0.6)\n"
            "for _ in range(10): time.sleep(0.2); tick('alive')\n"
            "print('still here', flush=True)\n"
            "time.sleep(30)\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=root)
    assert r.returncode == 17, (r.returncode, r.stderr[-2000:])
    assert "still here" in r.stdout and "rank 3 made no progress" in r.stderr and "last stage: alive" in r.stderr

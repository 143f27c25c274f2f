"""Data-parallel path on a real device: two ranks (gloo transport, both on cuda:0 — one GPU box)
run the HIP model through GradReducer + SyncBN statistics.

(a) replicated shard: the averaged gradients must equal a single-process run on that shard
    (validates bucket averaging and the (sum, sumsq, count) SyncBN algebra);
(b) different shards: all ranks must end with identical gradients and running statistics.
A full-batch vs sharded comparison is not an invariant of the reference either: the voxel lattice
origin is the rank-local coordinate minimum (model.py:96-98), so patch grouping depends on the shard."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    try:
        import faulthandler
        os.makedirs("gpurun_out", exist_ok=True)
        _fh = open(f"gpurun_out/parallel_worker_{rank}.trace", "w")
        faulthandler.dump_traceback_later(120, file=_fh, exit=False)   # where is a stuck worker? (diagnostic)
        _worker_body(rank, world, port, q)
        faulthandler.cancel_dump_traceback_later()
    except BaseException as e:  # report instead of leaving the parent to time out
        import traceback
        q.put((rank, {"error": f"{type(e).__name__}: {e}\n{traceback.format_exc()}"}))
        raise


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", LOTUS_DIST_BACKEND="gloo")
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import golden_util as gu
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import config as lcfg, ops, parallel, synth
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
    from weights_util import seeded_state_dict

    parallel.init_distributed()
    cfg = lcfg.preset("tiny")
    sd = seeded_state_dict(gu.state_template(cfg), 3, "scaled")
    perms = [[1, 0, 3, 2], [2, 3, 0, 1]]

    def build():
        m = SimplePolicyPTV3CA(cfg)
        m.load_state_dict(sd)
        m = m.cuda().train()
        m.ptv3_model.proj_drop = m.ptv3_model.attn_drop = 0.0
        m.act_proj_head.dropout = 0.0
        m.ptv3_model.order_perms = perms
        return m

    def dev(b):
        return {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if k == "disc_pos_probs" else v))
                for k, v in b.items()}

    def run(m, batch, red):
        if red is not None:
            red.zero_grad()
        else:
            m.zero_grad(set_to_none=True)
        _, losses = m(dev(batch), compute_loss=True, compute_final_action=False)
        losses["total"].backward()
        if red is not None:
            red.finish()
        torch.cuda.synchronize()
        return torch.cat([p.grad.flatten() for p in m.parameters()]).clone()

    res = {}
    # (a) replicated shard
    shard = synth.synth_batch(2, 400, ragged=True, seed=50)
    m = build()
    red = parallel.GradReducer(m, bucket_mb=0.5)
    parallel.enable_sync_batchnorm()
    g_dp = run(m, shard, red)
    rs_dp = m.ptv3_model.embedding.stem.norm.running_var.clone()
    # two more steps: from the second backward on the nodes write their gradients straight into the bucket buffer (ops.GRAD_ARENA)
    g_dp2 = run(m, shard, red)
    g_dp3 = run(m, shard, red)
    res["arena_inplace_fraction"] = red.inplace_floats / max(1, red.inplace_floats + red.copied_floats)
    res["arena_steps_equal"] = bool(torch.equal(g_dp2, g_dp3)) and ((g_dp3 - g_dp).norm() / g_dp.norm()).item() < 1e-6
    ops.BnState.reduce = None
    m1 = build()
    g_1 = run(m1, shard, None)
    res["replicated_rel_err"] = ((g_dp - g_1).norm() / g_1.norm()).item()
    res["replicated_rv_err"] = (rs_dp - m1.ptv3_model.embedding.stem.norm.running_var).abs().max().item()
    # (b) different shards
    parallel.enable_sync_batchnorm()
    m2 = build()
    red2 = parallel.GradReducer(m2, bucket_mb=0.5)
    g_r = run(m2, synth.synth_batch(2, 400, ragged=True, seed=60 + rank), red2)
    buf = [torch.zeros_like(g_r) for _ in range(world)]
    dist.all_gather(buf, g_r)
    res["cross_rank_diff"] = (buf[0] - buf[1]).abs().max().item()
    rv = m2.ptv3_model.enc.enc1.down.norm[0].running_var.clone()
    rvs = [torch.zeros_like(rv) for _ in range(world)]
    dist.all_gather(rvs, rv)
    res["cross_rank_rv_diff"] = (rvs[0] - rvs[1]).abs().max().item()
    res["finite"] = bool(torch.isfinite(g_r).all())
    # (c) UNEQUAL shards of one batch (3 clouds / 1 cloud) against a single process on the whole batch.  With a common
    # bounding box (synth.align_extents) every shard has the lattice origin and serialisation depth of the full batch, the
    # BatchNorm statistics are over all points (SyncBN messages carry the counts), and parallel.shard_loss_scale turns the
    # rank average of per-rank means into the mean over all clouds — so every gradient must agree with the one-process run.
    full = synth.align_extents(synth.synth_batch(4, 500, ragged=True, seed=70))
    shards = [[0, 1, 2], [3]]
    parallel.enable_sync_batchnorm()
    m3 = build()
    red3 = parallel.GradReducer(m3, bucket_mb=0.5)
    red3.zero_grad()
    mine = synth.take_clouds(full, shards[rank])
    _, l3 = m3(dev(mine), compute_loss=True, compute_final_action=False)
    (l3["total"] * parallel.shard_loss_scale(len(shards[rank]), 4, world)).backward()
    red3.finish()
    torch.cuda.synchronize()
    g3 = torch.cat([p.grad.flatten() for p in m3.parameters()]).clone()
    rv3 = m3.ptv3_model.enc.enc1.down.norm[0].running_var.clone()
    ops.BnState.reduce = None
    m4 = build()
    g4 = run(m4, full, None)
    res["unequal_shards_rel_err"] = ((g3 - g4).norm() / g4.norm()).item()
    per = []
    o = 0
    for p_ in m4.parameters():
        a, b = g3[o:o + p_.numel()], g4[o:o + p_.numel()]
        per.append(((a - b).norm() / (b.norm() + 1e-3 * g4.norm())).item())
        o += p_.numel()
    res["unequal_shards_worst_param"] = max(per)
    res["unequal_shards_rv_err"] = (rv3 - m4.ptv3_model.enc.enc1.down.norm[0].running_var).abs().max().item()
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(port):
    import queue

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    try:
        out = dict(q.get(timeout=180) for _ in ps)
    except queue.Empty:
        out = None
    for p in ps:
        p.join(timeout=5 if out is None else 60)
        if p.is_alive():
            p.kill()
    return out


def test_two_rank_data_parallel_on_device():
    # a stuck rendezvous / transport (seen once in ~25 runs on a shared box, never reproduced) gets ONE retry on a fresh
    # port; wrong numbers never do
    out = _run_two_ranks(29600 + (os.getpid() % 1000))
    if out is None:
        out = _run_two_ranks(31600 + (os.getpid() % 1000))
    if out is None:
        traces = "".join(open(f).read() for f in ("gpurun_out/parallel_worker_0.trace", "gpurun_out/parallel_worker_1.trace")
                         if os.path.exists(f))
        pytest.fail("two-rank workers did not finish within the time limit (twice)\n" + traces[-4000:])
    for r, res in out.items():
        assert "error" not in res, res["error"]
        assert res["finite"]
        assert res["replicated_rel_err"] < 1e-5, res
        assert res["replicated_rv_err"] < 1e-6, res
        assert res["cross_rank_diff"] == 0.0, res
        assert res["cross_rank_rv_diff"] == 0.0, res
        assert res["unequal_shards_rel_err"] < 1e-5 and res["unequal_shards_worst_param"] < 1e-5, res
        assert res["unequal_shards_rv_err"] < 1e-6, res
        # from the second step on most gradients are born in the reducer's bucket buffer, and nothing changes numerically
        assert res["arena_steps_equal"] and res["arena_inplace_fraction"] > 0.5, res


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` started bare must spawn two ranks (torch.distributed.run on 127.0.0.1), run the
    data-parallel step (GradReducer + SyncBN statistics; gloo here because both ranks share the one device of this
    box) and print ONE JSON line from rank 0 (it used to assert on WORLD_SIZE)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--batch", "4", "--npoints", "1024", "--no-roofline", "--no-other-modes", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["scaling"] == "weak"
    assert out["rccl_ranks"] in (0, 2) and out["value"] > 0


def test_one_rank_rccl_rehearsal_of_the_data_parallel_step():
    """Only one device is reachable here, and RCCL refuses two ranks on one device — but a ONE-rank RCCL communicator
    still sends every collective of the data-parallel step through the real library: parameter broadcast, bucketed AVG
    all-reduce launched from backward hooks on the communication stream, fp64 SyncBN messages on their own communicator
    (LOTUS_FORCE_COLLECTIVES=1).  The step must run and report rccl_ranks = 1."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LOTUS_FORCE_COLLECTIVES"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--batch", "4",
                        "--npoints", "1024", "--no-roofline", "--no-other-modes", "--no-cpu-baseline", "--no-fresh-batches"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["rccl_ranks"] == 1 and out["config"]["dist_backend"] == "nccl" and out["value"] > 0
    # ... and the collectives of the step went through the library's own communicators (csrc/comm.cpp), not ProcessGroupNCCL
    # ... on two streams that lotus_stream_probe found unable to hold each other back
    assert out["reducer"]["native_rccl_lanes"] == {"comm": True, "main": True, "streams_independent": True}, out["reducer"]


_NATIVE_SCRIPT = r"""
import os, sys, json
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, torch.distributed as dist
import robot_3dlotus_amd
from robot_3dlotus_amd import parallel
os.environ["LOTUS_FORCE_COLLECTIVES"] = "1"
parallel.init_distributed()
dev = torch.device("cuda", 0)
torch.cuda.set_stream(parallel.training_stream())  # what a data-parallel trainer does right after init_distributed()
res = {}
lane = parallel.native_comm(None, "main")
res["lane"] = lane is not None
if lane is not None:
    a = torch.arange(257, dtype=torch.float64, device=dev); lane.all_reduce(a, parallel.NativeComm.SUM)
    b = torch.full((1 << 20,), 3.0, device=dev); lane.all_reduce(b, parallel.NativeComm.AVG)
    c = torch.tensor([0, 1, 0, 7], dtype=torch.int32, device=dev); lane.all_reduce(c, parallel.NativeComm.MAX)
    res["answers"] = bool(torch.equal(a, torch.arange(257, dtype=torch.float64, device=dev)) and float(b.min()) == 3.0 == float(b.max())
                          and c.tolist() == [0, 1, 0, 7])
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.GELU(), torch.nn.Linear(256, 256), torch.nn.GELU(), torch.nn.Linear(256, 8)).to(dev)
red = parallel.GradReducer(net, bucket_mb=0.05)
res["lanes"] = [red._lane_comm is not None, red._lane_main is not None]
res["independent"], res["one_comm"], res["retired"] = red.lanes_independent, red._lane_comm is red._lane_main, len(parallel._RETIRED_STREAMS)
x = torch.randn(32, 64, device=dev)
for _ in range(3):
    red.zero_grad()
    net(x).square().sum().backward()
    red.finish()
got = torch.cat([p.grad.flatten() for p in net.parameters()]).clone()
for p in net.parameters():
    p.grad = None
for h in red._hook_handles:
    h.remove()
net(x).square().sum().backward()
want = torch.cat([p.grad.flatten() for p in net.parameters()])
res["grads_equal"] = bool(torch.equal(got, want))
res["mask"] = red.used_mask.tolist() == [1] * len(red.params) and red.unused_of(red.step_id - 1) == set()
torch.cuda.synchronize()
print("RESULT " + json.dumps(res))
dist.destroy_process_group()
"""


@pytest.mark.parametrize("native", ["1", "0", "one-queue"])
def test_native_rccl_lanes_and_their_fallback(native):
    """csrc/comm.cpp: all-reduces issued straight into the caller's stream on the library's own RCCL communicators (one-rank
    communicators here: known answers for the three dtype / op pairs the step uses, the reducer's averaged gradients equal to the
    plain backward pass bit for bit, the usage mask), and LOTUS_DP_NATIVE=0 -> the same results through ProcessGroupNCCL.
    "one-queue": GPU_MAX_HW_QUEUES=1 puts every stream on one hardware queue — lotus_stream_probe must see that a collective parked
    on the communication stream would hold the training stream back (the two-communicator deadlock, csrc/stream_probe.hip), and
    the reducer must then keep ONE communicator for buckets and statistics, with the same gradients."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LOTUS_DP_NATIVE"] = "0" if native == "0" else "1"
    if native == "one-queue":
        env["GPU_MAX_HW_QUEUES"] = "1"
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + _NATIVE_SCRIPT], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["grads_equal"] and res["mask"], res
    if native == "1":
        assert res["lane"] and res["answers"] and res["lanes"] == [True, True], res
        assert res["independent"] is True and not res["one_comm"], res
    elif native == "one-queue":
        assert res["lane"] and res["answers"] and res["lanes"] == [True, True], res
        assert res["independent"] is False and res["one_comm"] and res["retired"] == 4, res
    else:
        assert not res["lane"] and res["lanes"] == [False, False], res

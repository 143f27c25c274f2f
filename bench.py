"""bench.py — keystep-samples/sec (train fwd+bwd) of the 3D-LOTUS v1 policy on N MI355X.

    python bench.py [--gpus N --steps K --warmup W]

N > 1: one process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process is
one rank; started bare, `--gpus N` re-executes itself through `python -m torch.distributed.run --nproc-per-node N`
on 127.0.0.1 and rank 0 prints the JSON line.  Ranks talk over RCCL (backend "nccl"); when fewer than N devices are
visible the ranks share devices and fall back to gloo (a functional check only: `rccl_ranks` in the JSON says which).

One "step" = forward + loss + backward (+ gradient all-reduce when N > 1) of the v1 model over one
synthetic GemBench-shape batch of 16 key-step clouds x 4096 points per GPU (BASELINE.json
configs[1]; weak scaling: 16 clouds per rank).  Inputs are resident in HBM before the timed
region.  Dropout is active (train mode, reference rates), weights are random-init of the v1
architecture.  Prints ONE JSON line on rank 0 with the metric, the roofline object of the
dominant kernel family (dense fp32-MFMA linear layers, timed live with HIP events by replaying
the step's launch list) and the CPU baseline (the oracle timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GFLOP_PER_SAMPLE = 60.5  # fwd+bwd algorithmic work per key-step sample, v1 @ 4096 pts (its probe batch)
DENSE_SHARE = 0.663      # dense (nn.Linear) share of those multiply-adds: 642 of 968 GFLOP at 16 clouds
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (AMD's 5 PF headline includes 2:1 sparsity)
HBM_PEAK_GBPS = 8000.0


def side_workload(extra, timeout_s=300):
    """Run this script again for another workload of BASELINE.json (configs[3] motion planner, configs[4] PerAct bf16) and
    return the essentials of its JSON line: side measurements printed next to the headline, never as `value`."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "20", "--warmup", "6", "--no-cpu-baseline", "--no-fresh-batches",
           "--no-other-modes", "--no-side-workloads"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        o = json.loads(line)
        keep = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "dtype": o["dtype"], "workload": o["config"]["workload"]}
        if "roofline" in o:
            keep["roofline"] = {k: o["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "launches_per_step", "ms_per_step")
                                if k in o["roofline"]}
        return keep
    except Exception as e:  # noqa: BLE001  (a side measurement must never take the headline down)
        return {"value": None, "error": f"{type(e).__name__}: {str(e)[:200]}"}


def dev_batch(batch, dev):
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(dev)
        elif k == "disc_pos_probs":
            out[k] = torch.cat([t.reshape(-1) for t in v]).to(dev)  # packed once, stays in HBM
        elif k == "gt_trajs_disc_pos_probs":
            out[k] = torch.cat([t.reshape(t.shape[0], -1) for t in v], 1).to(dev)  # [T, sum 3*n*nb]
        else:
            out[k] = v
    return out


def gemm_roofline(ops, calls, dev, reps=5, report=None):
    """Replay every dense-linear launch of one training step (forward, dgrad, wgrad) on the
    current stream, bracketed by HIP events, and return (flops per step, ms per step, launches)."""
    import torch

    uniq = {}
    for c in calls:
        uniq[c] = uniq.get(c, 0) + 1
    tot_ms, tot_flop, n_launch = 0.0, 0.0, 0
    for (kind, M, N, K), cnt in uniq.items():
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.02
        dy = torch.randn(M, N, device=dev)
        fn = {"fwd": lambda: ops.linear_fwd(x, w, None), "dgrad": lambda: ops.linear_dgrad(dy, w),
              "wgrad": lambda: ops.linear_wgrad(dy, x)}[kind]
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tot_ms += ms * cnt
        tot_flop += 2.0 * M * N * K * cnt
        n_launch += cnt
        if report is not None:
            report.append((ms * cnt, kind, M, N, K, cnt, ms, 2e-9 * M * N * K / ms))
    return tot_flop, tot_ms, n_launch


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline_measure(cfg_name="v1", clouds=16, npoints=4096, warmup=3, iters=10, budget_s=150.0):
    """the oracle (CPU PyTorch restatement of the reference path, fp32, flash-path semantics) on the
    canonical batch — 16 clouds x 4096 points, seeds 0 — forward + loss + backward, 3 warm-up + 10 timed iterations,
    median; thread count chosen by a short sweep (one iteration each) and stated with the CPU model.  Bounded: when
    the host is too slow for 13 iterations inside `budget_s`, fewer timed iterations are taken and the count is stated."""
    import golden_util as gu
    from oracle.model import Oracle
    from robot_3dlotus_amd import config as lcfg, synth
    from weights_util import seeded_state_dict

    t_begin = time.time()
    ncores = os.cpu_count() or 1
    cfg = lcfg.preset(cfg_name)
    torch.manual_seed(0)
    sd = seeded_state_dict(gu.state_template(cfg), 0, "init")
    batch = synth.synth_batch(clouds, npoints, seed=0)
    perms = [[0, 1, 2, 3]] * len(cfg.ptv3_config.enc_channels)

    def one():
        sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
        t0 = time.time()
        out = Oracle(sdg, lcfg.plain(cfg), training=True).forward(batch, perms)
        out["losses"]["total"].backward()
        return time.time() - t0

    sweep = {}
    for th in sorted({min(ncores, t) for t in (8, 16, 32, 64)}):  # beyond a few dozen threads the small per-level ops lose to fork/join
        torch.set_num_threads(th)
        if not sweep:
            one()  # first touch (allocator, thread pool)
        sweep[th] = one()
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    for _ in range(max(0, warmup - 2)):  # the sweep already ran >= 2 iterations at this size
        one()
    times = []
    while len(times) < iters and (len(times) < 3 or time.time() - t_begin + 1.3 * max(times) < budget_s):
        times.append(one())
    t = sorted(times)[len(times) // 2]
    return {"value": round(clouds / t, 4), "unit": "keystep-samples/s", "cores": threads, "kind": "port",
            "cpu": _cpu_model(), "host_threads": ncores, "gflops": round(clouds * GFLOP_PER_SAMPLE / t, 1),
            "thread_sweep_s_per_iter": {str(k): round(v, 3) for k, v in sweep.items()},
            "sample": f"{clouds} clouds x {npoints} pts (the bench batch), v1 model, fwd+loss+bwd, median of {len(times)} timed "
                      f"iterations after {warmup} warm-up (oracle/model.py, torch CPU fp32, best of the thread sweep = {threads} "
                      f"of {ncores} host threads, {_cpu_model()})"}


def cpu_baseline(timeout_s=240):
    """Run the measurement in a child process so that a slow or oversubscribed host cannot stall the
    bench: bounded to `timeout_s` seconds of wall clock."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True,
                           text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "keystep-samples/s", "cores": 0, "kind": "port", "sample": "failed: " + r.stderr[-200:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "keystep-samples/s", "cores": 0, "kind": "port",
                "sample": f"oracle did not finish on this host within {timeout_s} s"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script through torch.distributed.run
    (env:// rendezvous on 127.0.0.1, the semantics of genrobo3d/train/utils/distributed.py:67-81) and pass the
    JSON line of rank 0 through."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    return subprocess.run(cmd, env=env).returncode


def _watchdog(rank, limit_s):
    """-> tick(what=None).  A daemon thread ends the process (exit code 17, Python stacks of all threads on stderr) when
    `limit_s` seconds pass without a tick."""
    import faulthandler
    import threading

    state = {"t": time.monotonic(), "what": "start-up"}

    def tick(what=None):
        state["t"] = time.monotonic()
        if what is not None:
            state["what"] = what

    def watch():
        while True:
            time.sleep(min(5.0, limit_s / 4))
            if time.monotonic() - state["t"] > limit_s:
                print(f"bench.py: rank {rank} made no progress for {limit_s:.0f} s (last stage: {state['what']}); giving up",
                      file=sys.stderr, flush=True)
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                os._exit(17)

    threading.Thread(target=watch, daemon=True, name="bench-watchdog").start()
    return tick


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="key-step clouds per GPU")
    ap.add_argument("--npoints", type=int, default=4096)
    ap.add_argument("--ragged", action="store_true",
                    help="clouds of n ~ U(npoints / 2, npoints) points (SURVEY 8d: the ragged variant of the canonical batch; the "
                         "headline is quoted on exactly npoints per cloud)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--with-optimizer", action="store_true",
                    help="also run lr schedule + clip_grad_norm_(10) + fused AdamW inside every step (SURVEY 8f rank 1); "
                         "the headline metric of BASELINE.json is forward + backward only, so this is off by default")
    ap.add_argument("--act-storage", choices=["fp32", "bf16"], default="bf16",
                    help="peract workload only: storage type of the activation tensors in HBM (bf16 = BASELINE configs[4])")
    ap.add_argument("--workload", choices=["policy", "mp", "peract"], default="policy",
                    help="policy = 3D-LOTUS v1 (BASELINE configs[1], the headline metric); mp = the 3D-LOTUS++ motion "
                         "planner (configs[3]); peract = RLBench-18task config (configs[4]): the v1 network on dense 4096-point "
                         "clouds augmented as the PerAct job script does (aug_max_rot 45 -> duplicate voxels), bf16 operands — "
                         "side measurements, same step structure")
    ap.add_argument("--no-fresh-batches", action="store_true",
                    help="skip the end-to-end side measurement (a new pinned host batch per step: H2D + front-end in the timed region)")
    ap.add_argument("--gemm-precision", choices=["fp32", "bf16x3", "bf16"], default="fp32",
                    help="operand precision of the dense fwd/dgrad products: fp32 = exact fp32 MFMA (default, the parity mode "
                         "the headline is quoted in); bf16x3 / bf16 are the opt-in faster modes (DESIGN.md 4)")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the bf16x3 / bf16 side measurement")
    ap.add_argument("--no-side-workloads", action="store_true",
                    help="skip the side measurements of the other BASELINE workloads (PerAct bf16 at 16 / 64 clouds, motion planner)")
    ap.add_argument("--gemm-report", default=None, help="write per-shape GEMM timings (diagnostic) to this file")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        import robot_3dlotus_amd  # noqa: F401
        print(json.dumps(cpu_baseline_measure()), flush=True)
        return

    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import config as lcfg, ops, parallel, synth
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    # stdout carries exactly ONE line, the JSON of rank 0: native libraries write banners to file descriptor 1 (RCCL prints
    # its version block there when the communicator is torn down — after our line), so everything else goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.gpus > 1 and torch.cuda.device_count() < args.gpus and "LOTUS_DIST_BACKEND" not in os.environ:
        os.environ["LOTUS_DIST_BACKEND"] = "gloo"  # ranks share a device: RCCL needs one device per rank
    rank, local, world = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus}, "
                         f"or start bench.py bare and let it spawn its ranks)")
    rccl_ranks = world if (dist.is_initialized() and dist.get_backend() == "nccl") else 0
    if world > 1 and torch.cuda.device_count() >= world and rccl_ranks != world:
        raise SystemExit(f"bench.py: {world} ranks on {torch.cuda.device_count()} visible devices must run over RCCL "
                         f"(backend is {dist.get_backend() if dist.is_initialized() else None})")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # N > 1: a collective that never completes (a rank that died, a transport that does not come up) would otherwise hold the
    # node until the caller's own limit; every step ticks, and LOTUS_BENCH_WATCHDOG_S seconds without one end the rank loudly
    tick = _watchdog(rank, float(os.environ.get("LOTUS_BENCH_WATCHDOG_S", "300"))) if world > 1 else (lambda what=None: None)
    dp_stream = None
    if dist.is_initialized() and os.environ.get("LOTUS_DIAG_NO_PRIME") != "1":
        dp_stream = parallel.training_stream()  # the step's four streams, created back to back before any communicator exists
    torch.manual_seed(0)
    mp = args.workload == "mp"
    peract = args.workload == "peract"
    if peract and args.gemm_precision == "fp32":
        args.gemm_precision = "bf16"  # train_3dlotus_policy_peract.sh: the bf16 compute mode of BASELINE configs[4]
    if mp:
        from robot_3dlotus_amd.motion_planner import MotionPlannerPTV3CA
        model = MotionPlannerPTV3CA(lcfg.preset("mp")).to(dev).train()
    else:
        model = SimplePolicyPTV3CA(lcfg.preset("v1")).to(dev).train()
        if peract and args.act_storage == "bf16":
            model.act_storage = "bf16"  # activations stored in bf16 (lotus_b16_* twins), fp32 master weights / gradients
    reducer = None
    if world > 1 or os.environ.get("LOTUS_FORCE_REDUCER") == "1" or dist.is_initialized():  # flat-buffer bucketed RCCL all-reduce overlapped with backward + SyncBN statistics
        reducer = parallel.GradReducer(model, bucket_mb=32.0)
        if os.environ.get("LOTUS_BENCH_NO_SYNCBN") != "1":  # (diagnostic: the reducer alone; never set by the driver)
            parallel.enable_sync_batchnorm()
    if reducer is not None:
        reducer.time_exposed = True
    host_batch = (synth.synth_batch_mp if mp else synth.synth_batch)(args.batch, args.npoints, ragged=args.ragged, seed=rank)
    if peract:  # aug_max_rot 45 + jitter (job_scripts/train_3dlotus_policy_peract.sh:42): 1-7 % duplicate voxels
        host_batch = synth.augment_clouds(host_batch, seed=rank, max_rot_deg=45.0)
    batch = dev_batch(host_batch, dev)

    params = [p for p in model.parameters() if p.requires_grad]
    opt = None
    if args.with_optimizer:
        from types import SimpleNamespace
        from robot_3dlotus_amd import optim as loptim
        topts = SimpleNamespace(learning_rate=1e-4, weight_decay=0.05, optim="adamw", betas=[0.9, 0.98], lr_sched="cosine",
                                warmup_steps=5000, num_train_steps=150000, grad_norm=10.0)  # simple_policy_ptv3.yaml TRAIN
        opt, init_lrs = loptim.build_optimizer(model, topts)
        gstep = [0]
    # gradients are dropped (set to None) before every step and nothing reads them during backward except the
    # reducer's bucket flush (which joins the weight-gradient stream itself): one join per backward pass
    ops.set_wgrad_join("end")
    ops.set_gemm_precision(args.gemm_precision)

    # When is the NEXT batch announced?  Before this step's forward ("early": its integer front-end runs under the forward) when
    # the step is data parallel — the host then runs a whole step ahead, which the reducer's extra host work needs — and always
    # for NEW host batches (the side measurement below: +18 %); after the forward ("late": under the backward pass, the round-5
    # order) for the single-GPU step on a resident batch, where it measures 0.6 % faster.  LOTUS_BENCH_PREFETCH_LATE=0|1 forces one.
    _pl = os.environ.get("LOTUS_BENCH_PREFETCH_LATE")
    PREFETCH_LATE = (reducer is None) if _pl is None else (_pl == "1")
    host_t = [0.0, 0.0, 0.0, 0]  # host seconds inside forward / backward / finish of the steps (enqueue time, no synchronisation)

    def step():
        tick()
        t0 = time.perf_counter()
        if reducer is not None:
            reducer.zero_grad()
        else:
            for p in params:  # model.zero_grad(set_to_none=True) without the module-tree walk (2 ms of host time)
                p.grad = None
        # the NEXT step's batch is announced before this step's forward: its integer front-end is launched by that forward (behind
        # this batch's own tables) and runs under it; LOTUS_BENCH_PREFETCH_LATE=1: the round-5 order (after the forward)
        if not PREFETCH_LATE:
            if model.ptv3_model._pending is None:
                model.prefetch(batch)  # (first step only: this step's own front-end)
            model.prefetch(batch)
        _, losses = model(batch, compute_loss=True, compute_final_action=False)
        if PREFETCH_LATE:
            model.prefetch(batch)
        t1 = time.perf_counter()
        losses["total"].backward()
        t2 = time.perf_counter()
        if reducer is not None:
            reducer.finish()
        host_t[0] += t1 - t0; host_t[1] += t2 - t1; host_t[2] += time.perf_counter() - t2; host_t[3] += 1
        if opt is not None:
            gstep[0] += 1
            loptim.set_lr(opt, init_lrs, gstep[0], topts)
            opt.clip_grad_norm_(topts.grad_norm)
            opt.step()
        return losses

    # the step runs on a high-priority stream: the weight-gradient and front-end side streams then only fill
    # the CUs the critical path leaves idle instead of time-slicing with it
    torch.cuda.synchronize()
    hp = os.environ.get("LOTUS_HIPRIO", "1")
    hi = torch.cuda.Stream(priority=-1) if hp == "1" else (torch.cuda.Stream() if hp == "2" else torch.cuda.current_stream())
    if dp_stream is not None:
        hi = dp_stream
    torch.cuda.set_stream(hi)
    tick("warm-up steps")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    tick("timed steps")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    bn_msgs0 = parallel.BN_MESSAGES
    if reducer is not None:
        reducer.exposed_comm_ms()  # drop the warm-up samples
    # the cyclic garbage collector is parked for the timed regions (as trainers of this size do: gc.freeze() after the warm-up): a
    # generation-2 pass over the step's object graph is a 5-30 ms host stall that lands in one step out of a few hundred
    import gc
    if os.environ.get("LOTUS_BENCH_GC") != "1":
        gc.collect(); gc.freeze(); gc.disable()
    host_t[:] = [0.0, 0.0, 0.0, 0]
    from robot_3dlotus_amd import frontend as lfe
    lfe.SYNC_WAIT = [0.0]
    t0 = time.perf_counter()
    step_marks = [] if os.environ.get("LOTUS_DIAG_STEP_TIMES") == "1" else None
    for _ in range(args.steps):
        if step_marks is not None:
            e_ = torch.cuda.Event(enable_timing=True); e_.record(); step_marks.append(e_)
        losses = step()
    host_ms = [round(1e3 * v / max(1, host_t[3]), 3) for v in host_t[:3]] + [round(1e3 * lfe.SYNC_WAIT[0] / max(1, host_t[3]), 3)]
    lfe.SYNC_WAIT = None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if step_marks:
        print("step times ms:", " ".join("%.2f" % a.elapsed_time(b) for a, b in zip(step_marks, step_marks[1:])), file=sys.stderr)
    bn_msgs = parallel.BN_MESSAGES - bn_msgs0
    # per-rank communication figures (outside the timed region): GPU time the backward stream waited in finish() per step,
    # GPU time of the SyncBN statistics messages of one extra step, and the point counts the ranks hold
    comm_stats = None
    if reducer is not None:
        exposed = reducer.exposed_comm_ms()
        parallel.BN_TIMING = []
        step()
        torch.cuda.synchronize()
        bn_ms = sum(a.elapsed_time(b) for a, b in parallel.BN_TIMING)
        parallel.BN_TIMING = None
        mine = torch.tensor([exposed or 0.0, bn_ms, float(sum(host_batch["npoints_in_batch"]))], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if world > 1:
            dist.all_gather(allr, mine)
        else:
            allr = [mine]
        pts = [float(t[2]) for t in allr]
        comm_stats = {"exposed_comm_ms_per_rank": [round(float(t[0]), 3) for t in allr],
                      "syncbn_ms_per_rank": [round(float(t[1]), 3) for t in allr],
                      "points_per_rank": [int(x) for x in pts], "point_imbalance_max_over_mean": round(max(pts) / (sum(pts) / len(pts)), 4),
                      "note": "exposed = GPU time per step between the end of backward and the last bucket's all-reduce "
                              "(HIP events in GradReducer.finish); syncbn = summed GPU time of the fp64 statistics messages of one step"}
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(losses["total"]).item()

    # side measurement (after the timed region of the headline): the same step with the opt-in operand precisions of the
    # dense and sparse-convolution products; reported next to the headline, never as `value`
    other = {}
    if args.gemm_precision == "fp32" and not mp and not args.no_other_modes:
        for mode in ("bf16x3", "bf16"):
            ops.set_gemm_precision(mode)
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([d2], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                d2 = float(t.item())
            other[mode] = round(args.batch * world * 20 / d2, 2)
        ops.set_gemm_precision("fp32")

    # side measurement: the same step followed by lr schedule + clip_grad_norm_(10) + fused AdamW (what a
    # trainer iteration adds to the headline's forward + backward), reported next to the headline, never as `value`
    with_opt = None
    if opt is None and not mp and not peract and args.gemm_precision == "fp32" and not args.no_other_modes:
        from types import SimpleNamespace
        from robot_3dlotus_amd import optim as loptim
        topts = SimpleNamespace(learning_rate=1e-4, weight_decay=0.05, optim="adamw", betas=[0.9, 0.98], lr_sched="cosine",
                                warmup_steps=5000, num_train_steps=150000, grad_norm=10.0)  # simple_policy_ptv3.yaml TRAIN
        backup = [p.detach().clone() for p in params]
        opt, init_lrs = loptim.build_optimizer(model, topts)
        gstep = [0]
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(15):
            step()
        torch.cuda.synchronize()
        d4 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([d4], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d4 = float(t.item())
        with_opt = {"value": round(args.batch * world * 15 / d4, 2), "unit": "keystep-samples/s", "steps": 15,
                    "ms_per_step": round(d4 / 15 * 1e3, 3),
                    "note": "forward + backward + lr schedule + clip_grad_norm_(10) + fused multi-tensor AdamW (csrc/optim.hip)"}
        opt = None
        with torch.no_grad():
            for p_, b_ in zip(params, backup):
                p_.copy_(b_)

    # end-to-end side measurement: a NEW host batch every step (pinned + packed by the collate function, as a DataLoader
    # with pin_memory would hand it over), uploaded and serialised by prefetch() under the previous step's backward, so
    # H2D and the integer front-end are inside the timed region (genrobo3d/models/base.py:29-34,
    # train_simple_policy.py:202-216).  Host-side synthesis of the clouds is done before the clock starts (the
    # reference's DataLoader workers do that in parallel).
    fresh = None
    if not args.no_fresh_batches and not mp and not peract and args.gemm_precision == "fp32":
        from robot_3dlotus_amd import data as ldata
        nfresh, nwarm = 10, 3

        def host(i):
            b = synth.synth_batch(args.batch, args.npoints, seed=7919 * (rank + 1) + i)
            items = []
            for k in range(len(b["npoints_in_batch"])):
                lo = sum(b["npoints_in_batch"][:k]); hi = lo + b["npoints_in_batch"][k]
                tl = sum(b["txt_lens"][:k]); th = tl + b["txt_lens"][k]
                items.append({"pc_fts": [b["pc_fts"][lo:hi]], "txt_embeds": [b["txt_embeds"][tl:th]], "ee_poses": [b["ee_poses"][k]],
                              "gt_actions": [b["gt_actions"][k]], "step_ids": [int(b["step_ids"][k])],
                              "disc_pos_probs": [b["disc_pos_probs"][k]], "pc_centroids": [], "data_ids": [f"s{i}-{k}"]})
            return ldata.ptv3_collate_fn(items, pack=True, pin=True)

        hb = [host(i) for i in range(nfresh + nwarm)]
        npts = sum(sum(b["npoints_in_batch"]) for b in hb[nwarm:])
        torch.cuda.synchronize()
        model.ptv3_model.drop_prefetch()  # (the resident batch announced by the last timed step)
        model.prefetch(hb[0])
        for i in range(nfresh + nwarm):
            if i == nwarm:
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t1 = time.perf_counter()
            if reducer is not None:
                reducer.zero_grad()
            else:
                for p_ in params:
                    p_.grad = None
            if i + 1 < nfresh + nwarm and _pl != "1":
                model.prefetch(hb[i + 1])
            _, losses = model(hb[i], compute_loss=True, compute_final_action=False)
            if i + 1 < nfresh + nwarm and _pl == "1":
                model.prefetch(hb[i + 1])
            losses["total"].backward()
            if reducer is not None:
                reducer.finish()
        torch.cuda.synchronize()
        d3 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([d3], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d3 = float(t.item())
        fresh = {"value": round(args.batch * world * nfresh / d3, 2), "unit": "keystep-samples/s", "steps": nfresh,
                 "points_per_s": round(npts * world / d3), "h2d_bytes_per_step": int(sum(v.numel() * v.element_size() for v in hb[-1].values() if isinstance(v, torch.Tensor))),
                 "note": "a new pinned host batch every step (same cloud size as the headline, different clouds and level sizes each step): "
                         "H2D upload + integer front-end prefetched under the previous backward, inside the timed region"}

    # one more step on EVERY rank (it contains collectives when N > 1) with the dense launches logged; rank 0 replays them
    calls, events = [], []
    if not args.no_roofline:
        ops.CALL_LOG = calls
        step()
        ops.CALL_LOG = None
        torch.cuda.synchronize()
        # ... and one with every dense launch bracketed by HIP events on its own stream: the in-step durations (the
        # weight gradients run concurrently with the critical stream, so these are longer than the isolated replay)
        ops.EVENT_LOG = events
        step()
        ops.EVENT_LOG = None
        torch.cuda.synchronize()

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = args.batch * world * args.steps / dt
        out = {
            "metric": "keystep-samples/sec (train fwd+bwd) 3D-LOTUS GemBench", "value": round(value, 2),
            "unit": "keystep-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "rccl_ranks": rccl_ranks,
            "host_ms_per_step": {"forward": host_ms[0], "backward": host_ms[1], "finish": host_ms[2], "of_forward_waiting_for_the_prefetched_front_end": host_ms[3],
                                 "note": "host time inside the calls of the timed steps (enqueue; a value near ms_per_step means the host waited for the GPU or bounds the step)"},
            "config": {"workload": f"3D-LOTUS v1 (68.18M params), {args.batch} key-step clouds x {args.npoints} pts "
                                   f"per GPU, fwd+loss+bwd, train mode (dropout on), fp32 exact (MFMA f32)",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "rccl_ranks": rccl_ranks,
                       "dist_backend": dist.get_backend() if dist.is_initialized() else None,
                       "model_gflop_per_sample": GFLOP_PER_SAMPLE,
                       "model_tflops": round(value * GFLOP_PER_SAMPLE / 1e3, 2)},
        }
        if mp:
            out["metric"] = "keystep-samples/sec (train fwd+bwd) 3D-LOTUS++ motion planner"
            out["config"]["workload"] = (f"3D-LOTUS++ motion planner (68.68M params, 5-step trajectory head), {args.batch} "
                                         f"clouds x {args.npoints} pts per GPU, fwd+loss+bwd, train mode, fp32 exact")
            out["config"].pop("model_gflop_per_sample"); out["config"].pop("model_tflops")
        if fresh is not None:
            out["fresh_batches"] = fresh
        if with_opt is not None:
            out["with_optimizer"] = with_opt
        if args.ragged:
            out["config"]["workload"] += f"; RAGGED clouds n ~ U({args.npoints // 2}, {args.npoints}) ({sum(host_batch['npoints_in_batch'])} points on rank 0)"
        if reducer is not None:
            sizes = [(hi_ - lo_) * 4 / 2 ** 20 for lo_, hi_ in reducer.buckets]
            out["comm"] = comm_stats
            out["reducer"] = {"buckets": len(sizes), "bucket_mb": [round(x, 1) for x in sizes], "last_bucket_mb": round(sizes[-1], 1),
                              "gradient_mb": round(sum(sizes), 1), "syncbn_messages_per_step": f"{bn_msgs / max(1, args.steps):.0f} (counted; one fp64 message per BN layer or merged pair and direction, own communicator)",
                              "points_rank0": int(sum(host_batch["npoints_in_batch"])),
                              # collectives issued through the library's own RCCL communicators (csrc/comm.cpp): "comm" = gradient
                              # buckets on the communication stream, "main" = SyncBN statistics + usage flags in the training stream
                              "native_rccl_lanes": {"comm": reducer._lane_comm is not None, "main": reducer._lane_main is not None,
                                                    # lotus_stream_probe: a collective parked on one of the two streams cannot hold the other back
                                                    "streams_independent": reducer.lanes_independent},
                              # gradients written by their backward node straight into the bucket buffer (no pack copy)
                              "gradient_fraction_born_in_bucket": round(reducer.inplace_floats / max(1, reducer.inplace_floats + reducer.copied_floats), 4)}
        if peract:
            out["metric"] = "keystep-samples/sec (train fwd+bwd) 3D-LOTUS RLBench-18task (PerAct) config"
            out["config"]["workload"] = (f"3D-LOTUS v1 network, PerAct preset (BASELINE configs[4]), {args.batch} dense clouds x {args.npoints} pts "
                                         f"per GPU after aug_max_rot 45 + jitter (duplicate voxels present), fwd+loss+bwd, train mode; "
                                         f"products with {args.gemm_precision} operands and fp32 accumulate, " +
                                         ("activations stored in bf16 (every [N, C] tensor in HBM; lotus_b16_* entry points), fp32 master "
                                          "weights, fp32 parameter gradients and statistics" if args.act_storage == "bf16" else
                                          "activations / weights stored in fp32"))
            out["config"]["act_storage"] = args.act_storage
            out["dtype"] = "bf16" if args.gemm_precision == "bf16" else args.gemm_precision
        if other:
            out["opt_in_modes"] = {"unit": "keystep-samples/s", **other,
                                   "note": "same step with ops.set_gemm_precision(mode): dense + sparse-conv + attention products as "
                                           "bf16x3 split (max logit error 2.2e-5, inside the 1e-4 bar) / plain bf16; not the headline"}
        if args.gemm_precision != "fp32" and not peract:
            out["config"]["workload"] += f"; dense fwd/dgrad products in {args.gemm_precision} (opt-in, NOT the headline mode)"
            out["dtype"] = f"f32 storage/accumulate, {args.gemm_precision} GEMM + conv operands"
        if opt is not None:
            out["config"]["workload"] += " + lr schedule + clip_grad_norm_(10) + fused AdamW step"
        if not args.no_roofline and peract:
            # BASELINE configs[4]: the dense family in bf16.  Algorithmic bytes of a launch = its operands once in their
            # storage types (activations 2 B with bf16 storage; weights 2 B when the layers read bf16 shadows, else the
            # 4-byte masters; weight gradients 4 B); bound = whichever of sum(bytes / 8 TB/s), sum(flops / 2.5 PF) is larger
            ab = 2.0 if args.act_storage == "bf16" else 4.0
            wb = 2.0 if (args.act_storage == "bf16" and getattr(model, "weight_shadows", False)) else 4.0
            in_ms = sum(e0.elapsed_time(e1) for *_, e0, e1 in events)
            in_flop = sum(2.0 * M * N * K for _, M, N, K, _, _ in events)
            in_bytes = sum((ab * (M * K + M * N) + 4.0 * N * K) if kind == "wgrad" else (ab * (M * K + M * N) + wb * N * K)
                           for kind, M, N, K, _, _ in events)
            t_hbm, t_mfma = in_bytes / (HBM_PEAK_GBPS * 1e9), in_flop / (MFMA_BF16_PEAK_TFLOPS * 1e12)
            if t_hbm >= t_mfma:
                ach = in_bytes / (in_ms * 1e-3) / 1e9
                out["roofline"] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                   "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None}
            else:
                ach = in_flop / (in_ms * 1e-3) / 1e12
                out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None}
            out["roofline"].update(kernel="gemm_kernel (dense bf16-MFMA linear fwd/dgrad/wgrad, bf16 storage)", launches_per_step=len(events),
                                   ms_per_step=round(in_ms, 3), gflop_per_step=round(in_flop / 1e9, 1),
                                   algorithmic_mb_per_step=round(in_bytes / 1e6, 1), tflops=round(in_flop / (in_ms * 1e-3) / 1e12, 2),
                                   hbm_bound_ms=round(t_hbm * 1e3, 3), mfma_bound_ms=round(t_mfma * 1e3, 3),
                                   measured="HIP events around every dense launch of one training step, on the stream it runs on")
        elif not args.no_roofline:
            rep = [] if args.gemm_report else None
            flop, gms, nl = gemm_roofline(ops, calls, dev, report=rep)
            if rep is not None:
                instep = {}
                for kind, M, N, K, e0, e1 in events:
                    d = instep.setdefault((kind, M, N, K), [0.0, 0])
                    d[0] += e0.elapsed_time(e1)
                    d[1] += 1
                with open(args.gemm_report, "w") as f:
                    f.write("in_step_total_ms kind M N K count ms_each_isolated TFLOPs_isolated ms_each_in_step TFLOPs_in_step hbm_bound_us mfma_bound_us\n")
                    rows = []
                    for tot, kind, M, N, K, cnt, ms, tf in rep:
                        ins = instep.get((kind, M, N, K), [0.0, 1])
                        ims = ins[0] / max(ins[1], 1)
                        byt = 4.0 * (M * K + N * K + M * N)
                        rows.append((ins[0], kind, M, N, K, cnt, ms, tf, ims, 2e-9 * M * N * K / max(ims, 1e-9), byt / 6.3e6, 2.0 * M * N * K / 157.3e6))
                    for r in sorted(rows, reverse=True):
                        f.write("%.3f %s %d %d %d %d %.4f %.1f %.4f %.1f %.1f %.1f\n" % r)
            ach_iso = flop / (gms * 1e-3) / 1e12
            in_ms = sum(e0.elapsed_time(e1) for *_, e0, e1 in events)
            in_flop = sum(2.0 * M * N * K for _, M, N, K, _, _ in events)
            ach = in_flop / (in_ms * 1e-3) / 1e12
            if "model_gflop_per_sample" in out["config"]:
                # the work of THIS batch, not SURVEY's probe batch: the dense launches are logged exactly;
                # the sparse-convolution / attention share is SURVEY 8d's split (dense = 0.663 of the multiply-adds), which
                # scales with the same pooled level sizes
                g = in_flop / 1e9 / DENSE_SHARE / max(1, args.batch)
                out["config"]["model_gflop_per_sample"] = round(g, 1)
                out["config"]["model_tflops"] = round(value * g / 1e3, 2)
                out["config"]["model_gflop_note"] = (f"{in_flop / 1e9:.0f} dense GFLOP logged for this batch / {DENSE_SHARE} (dense share of the "
                                                     f"multiply-adds, SURVEY 8d) / {args.batch} clouds; SURVEY's probe batch: {GFLOP_PER_SAMPLE}")
            # per-launch bound: a dense layer moves A + B + C once (4 bytes each, fp32 storage) and needs 2 M N K flops; it
            # cannot be faster than max(flops / MFMA peak, bytes / achievable HBM rate) — the thin C = 64 / 128 layers of
            # level 0 are HBM-bound, not MFMA-bound
            roof_ms = sum(max(2.0 * M * N * K / (MFMA_F32_PEAK_TFLOPS * 1e12), 4.0 * (M * K + N * K + M * N) / 6.3e12)
                          for _, M, N, K, _, _ in events) * 1e3
            # HBM bytes per GEMM launch and the two counters north_star names, from the committed rocprofv3 PMC passes
            # (profiles/pmc_passes.sh + pmc_report.py).  Only trusted when they were taken on THIS build of the library.
            traffic, evidence = None, None
            try:
                import hashlib
                from robot_3dlotus_amd import _capi as _lc
                import glob
                pmc_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))[-1]  # the latest round's passes
                pmc_name = "profiles/" + os.path.basename(pmc_path)
                pmc = json.load(open(pmc_path))
                sha = hashlib.sha256(open(_lc.LIB_PATH, "rb").read()).hexdigest()[:16]
                if pmc.get("library_sha256_16") == sha:
                    k = pmc["kernels"]
                    # the dense family = gemm_kernel + (round 5) gemm_dma_kernel: launch-weighted mean of their HBM bytes
                    fams = [k[n] for n in k if n.startswith("gemm_kernel") and "TAP" not in n or n.startswith("gemm_dma_kernel")]
                    traffic = round(sum(f["hbm_bytes_per_launch"] * f["launches_per_step"] for f in fams) / max(1e-9, sum(f["launches_per_step"] for f in fams)))
                    evidence = {"source": f"{pmc_name}, {pmc_name.replace('_pmc.json', '_sq.md')} (rocprofv3 --pmc, kernels serialised, this library build)",
                                "attention_mfma_util": {n: k[n]["mfma_util"] for n in ("attn_fwd_kernel", "attn_bwd2_kernel", "attn_bwd_kernel") if n in k},
                                "gemm_mfma_util": {n: k[n]["mfma_util"] for n in k if n.startswith("gemm_kernel") and "TAP" not in n or n.startswith("gemm_dma_kernel")},
                                # the neighbour gather of the convolution: since round 5 the gathered rows of the tap-grouped products
                                # (gemm_dma_tap_kernel) + the tap sum over the partial slab; fe_neighbour_kernel builds the tables
                                "neighbour_gather_GBps": {n: k[n]["achieved_GBps"] for n in k
                                                          if n.startswith(("gemm_dma_tap_kernel", "conv_tap_reduce_kernel", "conv_pairs_kernel", "fe_neighbour_kernel"))},
                                "conv_fetch_bytes_per_launch": {n: k[n]["fetch_bytes_per_launch"] for n in k
                                                                if n.startswith(("gemm_dma_tap_kernel", "conv_tap_reduce_kernel", "conv_pairs_kernel"))},
                                "hbm_peak_GBps": 8000, "hbm_achievable_GBps": 6300}
                else:
                    evidence = {"note": f"{pmc_name} was collected on library {pmc.get('library_sha256_16')}, this run loaded {sha}: "
                                        "counter figures withheld (re-run profiles/pmc_passes.sh)"}
            except (OSError, KeyError, ValueError, IndexError):
                pass
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic,
                               "kernel": "gemm_dma_kernel + gemm_kernel (dense fp32-MFMA linear fwd/dgrad/wgrad)",
                               "launches_per_step": len(events), "ms_per_step": round(in_ms, 3),
                               "per_launch_roof": {"ms_per_step": round(roof_ms, 3), "frac": round(roof_ms / max(in_ms, 1e-9), 4),
                                                   "note": "sum over the launches of max(2MNK / 157.3 TFLOP/s, 4(MK + NK + MN) B / 6.3 TB/s) "
                                                           "divided by the measured in-step time of the same launches"},
                               "avg_launch_us": round(1e3 * in_ms / max(len(events), 1), 2),
                               "gflop_per_step": round(in_flop / 1e9, 1),
                               "measured": "HIP events around every dense launch of one training step, on the stream it runs on "
                                           "(weight gradients overlap the critical stream); each launch includes its split-K "
                                           "reduction / epilogue kernel",
                               "isolated": {"achieved": round(ach_iso, 2), "frac": round(ach_iso / MFMA_F32_PEAK_TFLOPS, 4),
                                            "ms_per_step": round(gms, 3),
                                            "note": "same launches replayed one shape at a time on an otherwise idle GPU"}}
        if not args.no_roofline and not peract and evidence is not None:
            out["counters"] = evidence
        try:
            import hashlib
            from robot_3dlotus_amd import _capi as _lc2
            out["library_sha256_16"] = hashlib.sha256(open(_lc2.LIB_PATH, "rb").read()).hexdigest()[:16]
        except OSError:
            pass
        if world == 1 and not (mp or peract) and not args.no_side_workloads and not args.no_other_modes and args.gemm_precision == "fp32":
            # the other workloads BASELINE.json names, each in a fresh process after the headline: configs[4] (PerAct: bf16
            # activation + weight storage, fp32 masters) at 16 and 64 clouds per GPU, configs[3] (3D-LOTUS++ motion planner)
            torch.cuda.synchronize()
            out["other_workloads"] = {
                "peract_bf16_16_clouds": side_workload(["--workload", "peract"]),
                "peract_bf16_64_clouds": side_workload(["--workload", "peract", "--batch", "64", "--steps", "12", "--warmup", "5"]),
                "motion_planner_16_clouds": side_workload(["--workload", "mp", "--no-roofline"]),
                "note": "side measurements (same step structure, same script, own process): never the headline value"}
        if not args.no_cpu_baseline and world == 1:  # a reported baseline of the N = 1 line only (the other ranks would idle in the barrier)
            out["cpu_baseline"] = cpu_baseline()
        final_line = json.dumps(out)
    else:
        final_line = None
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if final_line is not None:
        os.write(json_fd, (final_line + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()

"""Data-parallel training over RCCL (torch.distributed backend "nccl" == RCCL on ROCm).

The path shards by independent units (key-step clouds): each rank runs the whole model on its own
clouds; the only exchanges are (1) the gradient all-reduce and (2) the BatchNorm statistics, as
in the reference (DDP + SyncBatchNorm, genrobo3d/train/utils/distributed.py:196-205,
train_simple_policy.py:116-117).  Design for xGMI (point-to-point links, no switch): gradients are
packed into ONE flat fp32 buffer, cut into a few large buckets that follow the order in which
backward finishes the gradients (learnt in the first pass, agreed across ranks by a broadcast); a
bucket's all-reduce is launched asynchronously from the post-accumulate hook of its last gradient,
so RCCL overlaps with the remaining backward.  Large buckets keep the collectives bandwidth-bound
over all 7 links instead of latency-bound.
"""
import contextlib
import os

import torch
import torch.distributed as dist

from . import _capi, ops


def init_distributed(backend=None):
    """env:// rendezvous as torchrun sets it up (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and os.environ.get("LOTUS_FORCE_COLLECTIVES") == "1" and torch.cuda.is_available():
        # single-GPU rehearsal of the multi-GPU path: a one-rank RCCL communicator, so that every collective of the
        # data-parallel step (gradient AVG all-reduce on the comm stream, fp64 SyncBN messages on their own communicator,
        # the parameter broadcast) goes through the real library on a box with one device
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 2000))
            dist.init_process_group(backend=backend or "nccl", init_method="env://", rank=0, world_size=1)
        _uniform_stream_priority()
        return 0, 0, 1
    if world == 1:
        return 0, 0, 1
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = os.environ.get("LOTUS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()  # (gloo smoke runs may put several ranks on one device)
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method="env://")
    _uniform_stream_priority()
    return rank, local, world


def _uniform_stream_priority():
    """Data-parallel processes: every stream of the step — the training stream a trainer makes current (high priority, as
    bench.py does), the weight-gradient, front-end and communication streams the package creates — gets the SAME, high priority.
    HIP maps streams onto at most GPU_MAX_HW_QUEUES = 4 hardware queues PER PRIORITY: the step's four streams then own the four
    high-priority queues, and nothing else in the process is created there (RCCL's internal streams, ProcessGroupNCCL's, the
    null stream and torch's pool are normal priority).  With a high-priority training stream over normal-priority side streams
    (the single-GPU arrangement) a side stream whose hardware queue lands next to the training stream's is not scheduled until
    that one runs dry: measured, the front-end's count copy or the weight gradients then finish a whole backward pass late
    (28 instead of 15.5 ms per step), and WHICH stream is hit changes with the order in which the process created its queues."""
    if os.environ.get("LOTUS_STREAM_PRIO") is None and torch.cuda.is_available():
        _capi.STREAM_PRIORITY = -1


def training_stream():
    """The stream a data-parallel trainer should make current for its steps (`torch.cuda.set_stream(parallel.training_stream())`
    right after init_distributed(), before the model and the reducer are built): high priority like the package's own three
    streams, and created together with them (see _capi.prime_step_streams)."""
    return _capi.prime_step_streams()


class NativeComm:
    """An RCCL communicator of the library's own (csrc/comm.cpp, lotus_comm_*): `all_reduce` is ONE ncclAllReduce enqueued on
    the current HIP stream — no ProcessGroupNCCL work object, end event or communicator-internal stream behind it (a blocking
    ProcessGroupNCCL collective costs the issuing stream ~11 us beyond the collective, 26 us through its own stream:
    tools/dbg/msg_cost.py).  torch.distributed stays the bootstrap: it carries rank 0's unique id to the other ranks.
    Construction is a collective over `group`; one instance per stream that sends ("lanes" below)."""
    SUM, MAX, AVG = 0, 1, 2
    _DT = {torch.float32: 0, torch.float64: 1, torch.int32: 2}

    def __init__(self, group=None):
        import ctypes

        import numpy as np

        from . import _capi
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")  # the instance torch itself loaded
        cpath = ctypes.create_string_buffer((path if os.path.exists(path) else "").encode())
        _capi.call_raw("lotus_comm_load", ctypes.addressof(cpath))
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        idb = np.zeros(128, dtype=np.uint8)
        if rank == 0:
            _capi.call_raw("lotus_comm_unique_id", idb.ctypes.data)
        t = torch.from_numpy(idb).cuda()
        dist.broadcast(t, dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        idb = np.ascontiguousarray(t.cpu().numpy())
        self.handle = _capi.query("lotus_comm_create", idb.ctypes.data, world, rank)
        if not self.handle:
            raise _capi.LotusError("lotus_comm_create failed: " + _capi.lib().last_error())
        self.world, self.rank = world, rank
        probe = torch.full((4,), float(rank + 1), dtype=torch.float64, device="cuda")  # known answer before anything relies on it
        self.all_reduce(probe, self.SUM)
        if probe.tolist() != [world * (world + 1) / 2.0] * 4:
            raise _capi.LotusError(f"native RCCL communicator: known-answer all-reduce returned {probe.tolist()}")

    def all_reduce(self, t, op):
        """In place, on the current stream (ops honour _capi.STREAM_OVERRIDE like every other launch)."""
        from . import _capi
        _capi.call("lotus_comm_allreduce", self.handle, t, t.numel(), self._DT[t.dtype], op)


_LANES = {}


def native_comm(group, lane):
    """The native communicator of (`group`, `lane`), created on first use — a COLLECTIVE call: every rank of the group must
    reach it in the same order.  lane "main": collectives issued from the training stream (SyncBatchNorm statistics, usage
    flags); lane "comm": the gradient buckets on the communication stream (one communicator per sending stream: NCCL orders
    the collectives of a communicator, and the two streams run concurrently).  None when the group does not run over RCCL,
    LOTUS_DP_NATIVE=0, or ANY rank failed to create it (decided together, so that no rank is left alone in a collective)."""
    key = (id(group) if group is not None else 0, lane)
    if key in _LANES:
        return _LANES[key]
    c = None
    if (dist.is_initialized() and dist.get_backend(group) == "nccl" and torch.cuda.is_available()
            and os.environ.get("LOTUS_DP_NATIVE", "1") != "0"):
        err = None
        try:
            c = NativeComm(group)
        except Exception as e:  # noqa: BLE001 - whatever went wrong, the ProcessGroup path still works
            err = e
        ok = torch.tensor([0 if c is None else 1], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            if dist.get_rank(group) == 0:
                import sys
                print(f"lotus parallel: native RCCL lane '{lane}' unavailable ({err}); using torch.distributed collectives", file=sys.stderr)
            c = None
    _LANES[key] = c
    return c


_RETIRED_STREAMS = []


def _checked_comm_stream(group):
    """-> (communication stream, independent): the package's "comm" stream, verified with lotus_stream_probe against the CURRENT
    stream (the training stream: make it current before building the reducer) in both directions — a kernel parked on one
    must not stop the other.  HIP deals its streams onto GPU_MAX_HW_QUEUES hardware queues in creation order, so a stream that
    shares the training stream's queue is retired (kept alive: its slot stays taken) and the next one tried, four times.
    `independent` is the MIN over the ranks of the group, so that every rank builds the same communicators.
    LOTUS_DP_PROBE=0 skips the check (independent = True)."""
    st = _capi.step_stream("comm")
    if os.environ.get("LOTUS_DP_PROBE", "1") == "0":
        return st, True
    cur = torch.cuda.current_stream()
    ok = False
    for _ in range(4):
        res = [_capi.query("lotus_stream_probe", a.cuda_stream, b.cuda_stream, 50) for a, b in ((st, cur), (cur, st))]
        if min(res) < 0:
            raise _capi.LotusError("lotus_stream_probe failed: " + _capi.lib().last_error())
        ok = res == [1, 1]
        if ok:
            break
        _RETIRED_STREAMS.append(st)
        st = _capi._STEP_STREAMS["comm"] = torch.cuda.Stream(priority=_capi.STREAM_PRIORITY)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0 and dist.get_rank(group) == 0:
        import sys
        print("lotus parallel: the communication stream cannot run independently of the training stream on some rank "
              "(shared hardware queue); gradient buckets and statistics share ONE communicator", file=sys.stderr)
    return st, bool(int(flag.item()))


class GradReducer:
    """Flat-buffer bucketed gradient averaging with backward overlap.

    Parameter gradients arrive as whatever tensors backward produced (`.grad` is None on entry, so autograd adopts
    them without an accumulation kernel per parameter).  When the last gradient of a bucket has arrived, ONE fused
    multi-tensor copy packs the bucket into its slice of the flat fp32 buffer, `.grad` of those parameters is
    re-pointed at the slice views, and the slice is all-reduced asynchronously (RCCL) while backward continues.

    Bucket order = gradient ARRIVAL order.  It is learnt during the first backward pass (which starts from reverse
    registration order), rank 0's order is broadcast, and the flat buffer is re-laid-out once: with the static order the
    text projection `txt_fc` — whose gradient is complete only at the very end of backward because every
    cross-attention block feeds it — sat in the first bucket and held 76 MB (28 % of all gradients) back until after
    backward.  Parameters that got no gradient in the learning pass form a trailing "cold" bucket.

    Protocol per optimisation step:  zero_grad();  [with no_sync(): backward of micro-batches 1..k-1];  backward of the
    last micro-batch;  finish().  Every bucket is all-reduced exactly once per step on every rank — from its hook when
    all its gradients arrived, else in finish() with the slots of gradient-less parameters zeroed — so ranks whose
    parameter usage differs stay in lock-step (what DistributedDataParallel(find_unused_parameters=True) achieves with
    its usage bitmap, which finish() mirrors with one MAX all-reduce of a per-parameter flag: a parameter no rank
    used ends with `.grad = None`, exactly as on one GPU).  A backward that would add onto already averaged gradients raises instead of silently diverging."""

    def __init__(self, module, bucket_mb=32.0, group=None, broadcast=True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._force = dist.is_initialized() and os.environ.get("LOTUS_FORCE_COLLECTIVES") == "1"  # world-1 rehearsal
        self.params = [p for p in module.parameters() if p.requires_grad]
        if broadcast and (self.world > 1 or self._force):
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0, group=group)
        dev = self.params[0].device
        self.flat = None
        self._cap = int(bucket_mb * (1 << 20) / 4)
        self._layout(list(reversed(self.params)), [])
        # gradient slabs of the backward nodes (ops.GRAD_ARENA): recorded in the learning pass, served from the flat buffer after it
        self._slab_rec, self._slab_seq, self._slab_i, self._pslab, self._arena_off = [], None, 0, {}, dev.type != "cuda"
        self._handles = []
        self.exposed_events = []
        self._avg = (self.world > 1 or self._force) and dist.get_backend(group) == "nccl"  # RCCL averages in the collective
        # native lanes (collective construction, same order on every rank): buckets on the communication stream, usage flags
        # on the training stream
        self._comm, self._keep = None, []
        self._lane_comm = self._lane_main = None
        self.lanes_independent = None
        self._defer_flush = False
        if self._avg and dev.type == "cuda":
            self._lane_main = native_comm(group, "main")
            # two communicators in flight need two streams that cannot hold each other back (csrc/stream_probe.hip): checked
            # on the device, decided together; without it the buckets share the statistics' communicator, whose collectives
            # RCCL orders by itself — slower (a statistics message can queue behind a bucket), never a deadlock
            self._comm, self.lanes_independent = _checked_comm_stream(group)
            self._lane_comm = native_comm(group, "comm") if self.lanes_independent else self._lane_main
            # ONE communicator for both kinds of message: its collectives must be ISSUED in one order on every rank.  The
            # statistics are (the same program everywhere), the buckets are issued where their last gradient arrives — and a
            # rank whose usage pattern differs from the learnt one flushes a bucket later than the others (finish()), i.e.
            # between different statistics messages.  So in this fallback no bucket leaves during backward: finish() sends
            # them all, in layout order, behind the last statistics message — no overlap, but one order by construction.
            self._defer_flush = not self.lanes_independent
        self._arrival, self._seen, self._learning = [], set(), True
        self._sync = True
        self._index = {p: i for i, p in enumerate(self.params)}
        pin = dev.type == "cuda"
        # host bitmap: parameters with a gradient this step (two pinned buffers in turn: the upload of one step may still be in
        # flight when the next step starts to fill its own)
        self._used_bufs = [torch.zeros(len(self.params), dtype=torch.int32, pin_memory=pin) for _ in range(2)]
        self._used = self._used_bufs[0]
        self._used_np = self._used.numpy()
        self.step_id, self._known = 0, {}
        # While the arrival order is being learnt every parameter carries a hook; afterwards only the LAST-arriving parameter
        # of each bucket does (~10 hooks instead of 421 per backward: the per-parameter hooks were most of the 5.5 % the
        # one-rank rehearsal cost before a byte crossed a link) — _hook_last checks that the bucket is
        # complete (every `.grad` set: zero_grad() dropped them) before it flushes, so a changed order can only delay a bucket
        # to finish(), never drop a gradient.  Gradient accumulation voids that check and returns to dense hooks (no_sync).
        self._hook_handles = []
        for p in self.params:
            p.grad = None
            self._hook_handles.append(p.register_post_accumulate_grad_hook(self._hook))

    def _layout(self, order, cold, slabs=None):
        """Contiguous buckets of >= cap elements over `order`, then one bucket with the `cold` parameters; every parameter
        gets a view into the flat buffer.  slabs = (sizes, where): the gradient slabs of the backward nodes learnt in the first
        pass — where[p] = (slab, offset in floats) for a parameter whose gradient is a view of slab number `slab`.  A slab is
        laid out WHOLE (its node's own layout, 256-byte aligned) at the place of its first-arriving parameter, so that the node
        can be handed that slice as its output buffer (_arena) and its gradients need no pack copy."""
        import numpy as np
        sizes_, where = slabs if slabs else ([], {})
        # layout units in arrival order: ("slab", id, floats, params) or ("param", p, floats, [p])
        units, seen_slab = [], {}
        for p in order:
            w = where.get(p)
            if w is None:
                units.append(["param", p, p.numel(), [p]])
            elif w[0] in seen_slab:
                seen_slab[w[0]][3].append(p)
            else:
                u = ["slab", w[0], (sizes_[w[0]] + 63) // 64 * 64, [p]]
                seen_slab[w[0]] = u
                units.append(u)
        total = sum(u[2] for u in units) + sum(p.numel() for p in cold)
        if slabs is not None or getattr(self, "flat", None) is None or self.flat.numel() != total:
            self.flat = torch.zeros(total, dtype=torch.float32, device=self.params[0].device)
        self.buckets, self._bparams, self._bviews, self._slot = [], [], [], {}
        self._slab_at = {}
        state = dict(p=[], v=[], n=0, offset=0, start=0)

        def close():
            if state["p"]:
                self.buckets.append((state["start"], state["offset"]))
                self._bparams.append(state["p"])
                self._bviews.append(state["v"])
                state.update(p=[], v=[], n=0, start=state["offset"])

        # Tapered tail: the LAST bucket of the arrival order is the one whose all-reduce nothing can hide (it starts when
        # backward ends), so the order's tail is cut into buckets of <= cap/32, cap/8 and cap/2 elements (1 / 4 / 16 MB at
        # the default 32 MB) counted from the end; everything before them follows the >= cap rule.  (20 MB of exposed
        # all-reduce at the end of every step otherwise: the encoder's level-2 gradients arrive ~3 ms before the stem's.)
        cuts, end, usz = set(), len(units), [u[2] for u in units]
        if sum(usz) >= 2 * self._cap and os.environ.get("LOTUS_DIAG_NO_TAPER") != "1":
            for tail_cap in (self._cap // 32, self._cap // 8, self._cap // 2):
                n, i = 0, end
                while i > 0 and (n == 0 or n + usz[i - 1] <= tail_cap):
                    i -= 1
                    n += usz[i]
                if i <= 0:
                    break
                cuts.add(i)  # a bucket boundary in front of units[i]
                end = i
        first_tail = min(cuts) if cuts else len(units)
        for k, (kind, ident, n, ps) in enumerate(units):
            if k in cuts:
                close()
            base = state["offset"]
            if kind == "slab":
                self._slab_at[ident] = base
            for p in ps:
                off = base + (where[p][1] if kind == "slab" else 0)
                self._slot[p] = len(self.buckets)
                state["p"].append(p)
                state["v"].append(self.flat[off:off + p.numel()].view_as(p))
            state["n"] += n
            state["offset"] += n
            if k < first_tail and state["n"] >= self._cap:
                close()
        close()
        for p in cold:
            self._slot[p] = len(self.buckets)
            state["p"].append(p)
            state["v"].append(self.flat[state["offset"]:state["offset"] + p.numel()].view_as(p))
            state["offset"] += p.numel()
        close()
        self._count = [len(ps) for ps in self._bparams]
        self._view = {p: v for ps, vs in zip(self._bparams, self._bviews) for p, v in zip(ps, vs)}
        index = {p: i for i, p in enumerate(self.params)}
        self._bindex = [np.asarray([index[p] for p in ps], dtype=np.int64) for ps in self._bparams]
        self._rearm()

    def _arena(self, n, dev):
        """ops.GRAD_ARENA while this reducer's backward runs: the output slab of the next backward node.  Learning pass: a
        plain tensor, remembered so that the hooks can tell which parameter's gradient lives where in it.  Afterwards: the slab's
        place in the flat buffer, as long as the nodes ask in the learnt order with the learnt sizes (anything else gets a
        plain tensor for the rest of the step and is packed by copy as before)."""
        if self._learning:
            t = torch.empty(n, dtype=torch.float32, device=dev)
            self._slab_rec.append((t.data_ptr(), n, t))
            return t
        i, seq = self._slab_i, self._slab_seq
        if seq is None or i >= len(seq) or seq[i][0] != n:
            self._slab_i = 1 << 30
            return None
        self._slab_i = i + 1
        at = seq[i][1]
        return None if at is None else self.flat[at:at + n]

    def _rearm(self):
        self._slab_i = 0
        self._pending = [0] * len(self.buckets)
        self._flushed = [False] * len(self.buckets)
        self._ready = [False] * len(self.buckets)
        self._next = 0

    def _hook(self, p):
        if not self._sync:
            return  # no_sync(): gradients accumulate locally in whatever tensors autograd holds
        b = self._slot[p]
        if self._flushed[b]:
            raise RuntimeError(
                "GradReducer: a gradient arrived for a bucket that was already averaged in this step — call "
                "reducer.zero_grad() before every optimisation step and wrap all but the last micro-batch of a "
                "gradient-accumulation step in `with reducer.no_sync():`")
        if self._learning and p not in self._seen:
            self._seen.add(p)
            self._arrival.append(p)
            if self._slab_rec and p.grad is not None and p.grad.is_contiguous():
                a = p.grad.data_ptr()
                for k, (base, n, _) in enumerate(self._slab_rec):
                    if base <= a and a + 4 * p.numel() <= base + 4 * n:
                        self._pslab[p] = (k, (a - base) // 4)
                        break
        self._pending[b] += 1
        if self._pending[b] == self._count[b]:
            # collectives are matched across ranks by issue order: buckets go out strictly in layout order (which IS the
            # arrival order, so nothing waits unless a rank's usage pattern differs from the learnt one)
            self._ready[b] = True
            while not self._defer_flush and self._next < len(self.buckets) and self._ready[self._next]:
                self._flush(self._next)
                self._next += 1

    def _hook_last(self, p):
        """Hook of the last-arriving parameter of a bucket (sparse mode, after the order is known)."""
        if not self._sync:
            return
        b = self._slot[p]
        if self._flushed[b]:
            raise RuntimeError(
                "GradReducer: a gradient arrived for a bucket that was already averaged in this step — call "
                "reducer.zero_grad() before every optimisation step and wrap all but the last micro-batch of a "
                "gradient-accumulation step in `with reducer.no_sync():`")
        if any(q.grad is None for q in self._bparams[b]):
            return  # (a parameter of the bucket has not arrived — unused this step, or a changed order: finish() flushes it)
        self._ready[b] = True
        while not self._defer_flush and self._next < len(self.buckets) and self._ready[self._next]:
            self._flush(self._next)
            self._next += 1

    def _sparse_hooks(self, order):
        """Replace the per-parameter hooks by one hook per bucket, on its last parameter in arrival order."""
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []
        rank_of = {p: i for i, p in enumerate(order)}
        for ps in self._bparams:
            hot = [p for p in ps if p in rank_of]
            if hot:  # (the bucket of never-seen parameters has no hook: finish() sends it)
                last = max(hot, key=lambda q: rank_of[q])
                self._hook_handles.append(last.register_post_accumulate_grad_hook(self._hook_last))

    def _dense_hooks(self):
        """(Back to) one counting hook on every parameter."""
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation (gradient_accumulation_steps > 1 in the reference trainer): inside this context
        backward only accumulates local gradients; the next backward outside of it packs the accumulated sums and
        all-reduces them once.  Accumulating onto a parameter's gradient is a read of a tensor the weight-gradient
        stream may still be writing, so the per-node join is required (ops.set_wgrad_join("node"), the default)."""
        if ops._JOIN != "node":
            raise RuntimeError("gradient accumulation needs ops.set_wgrad_join('node')")
        # One hook per bucket is only sound while `.grad is None` tells "not arrived in THIS backward": after a local
        # micro-batch every gradient exists, and a bucket whose last parameter (in the learnt order) arrives before another
        # one of its parameters would be packed with that parameter's stale partial sum.  Accumulation therefore
        # switches this reducer back to a hook on every parameter — they count the arrivals of the current backward — for good.
        if self.sparse_hooks:
            self.sparse_hooks = False
            self._dense_hooks()
        self._arena_off, ops.GRAD_ARENA = True, None  # (a second micro-batch would overwrite the first one's gradients in place)
        self._sync = False
        try:
            yield
        finally:
            self._sync = True

    def _flush(self, b):
        ps, views = self._bparams[b], self._bviews[b]
        self._flushed[b] = True
        have = [i for i, p in enumerate(ps) if p.grad is not None]
        bi = self._bindex[b]
        self._used_np[bi if len(have) == len(ps) else bi[have]] = 1  # (one vectorised store: 421 tensor element writes cost ~1 ms)
        lo, hi = self.buckets[b]
        buf = self.flat[lo:hi]
        # gradients that were born in their slot (slabs served by _arena) need no copy
        move = [i for i in have if ps[i].grad.data_ptr() != views[i].data_ptr()]
        f0, f1 = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.flat.numel()
        # (a gradient that sits in the flat buffer but not in its own slot — a node served another node's slab — is copied
        #  out first: its place may be another parameter's destination)
        grads = [ps[i].grad.clone() if f0 <= ps[i].grad.data_ptr() < f1 else ps[i].grad for i in move]
        self.copied_floats += sum(ps[i].numel() for i in move)
        self.inplace_floats += sum(ps[i].numel() for i in have) - sum(ps[i].numel() for i in move)
        dst = [views[i] for i in move]
        missing = [views[i] for i in range(len(ps)) if ps[i].grad is None] if len(have) < len(ps) else []

        def pack():
            if missing:  # finish(): gradient-less parameters contribute zeros to the average
                torch._foreach_zero_(missing)
            if grads and not _DIAG_NO_PACK:
                torch._foreach_copy_(dst, grads)
            self._reduce(buf)

        if buf.is_cuda:
            # pack + all-reduce on a communication stream that waits for the producers (the stream backward runs on and
            # the weight-gradient stream): the critical stream itself never waits for the lagging weight gradients
            if self._comm is None:
                # (measured, round 5: packing and reducing on the weight-gradient stream itself instead — one stream less —
                #  924-928 against 930-944 samples/s in the one-rank rehearsal)
                self._comm = _capi.step_stream("comm")
            comm = self._comm
            comm.wait_stream(torch.cuda.current_stream())
            ops.sync_side_stream(target=comm.cuda_stream)
            if grads or missing or self._lane_comm is None:
                with torch.cuda.stream(comm):
                    pack()
            else:
                # nothing to pack (every gradient of the bucket was born in its slot): the flush is ONE C-ABI call, sent to the
                # communication stream by the launch override — torch's stream context costs ~30 us of host time per flush
                prev, _capi.STREAM_OVERRIDE = _capi.STREAM_OVERRIDE, comm.cuda_stream
                try:
                    self._reduce(buf)
                finally:
                    _capi.STREAM_OVERRIDE = prev
            # the adopted gradient tensors were allocated on the backward stream and are read on `comm`: keep them
            # alive until finish() has made the backward stream wait for `comm` (cheaper than 421 record_stream calls)
            self._keep.extend(grads)
        else:
            pack()
        for i, (p, v) in enumerate(zip(ps, views)):
            if self.world > 1 or p.grad is not None:  # world == 1: an unused parameter keeps .grad None
                p.grad = v

    def _reduce(self, buf):
        if _DIAG_NO_MSG:
            return
        if self.world > 1 or self._force:
            if self._lane_comm is not None and buf.is_cuda:
                self._lane_comm.all_reduce(buf, NativeComm.AVG)  # one RCCL kernel behind the pack, on the communication stream
            elif self._avg and _AR_ON_COMM and buf.is_cuda:
                # a BLOCKING collective of ProcessGroupNCCL runs on the current stream — here the communication stream the
                # bucket was packed on: pack and all-reduce are two launches of ONE stream, no communicator-internal stream
                # (one hardware queue less) and no event hand-over between the two
                dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group)
            elif self._avg:
                self._handles.append(dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
            else:
                buf.mul_(1.0 / self.world)
                self._handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        if self._learning:
            # first backward seen: re-lay the flat buffer in arrival order.  The order is a property of the autograd
            # graph, but ranks whose first batches exercised different modules would disagree: rank 0 decides.  The
            # broadcast is issued by EVERY rank on every zero_grad() while learning (a rank-local condition — "my first
            # backward ran under no_sync / was skipped" — would leave the other ranks alone in the collective); an empty
            # order from rank 0 keeps every rank in learning mode.
            index = {p: i for i, p in enumerate(self.params)}
            order = [index[p] for p in self._arrival]
            if self.world > 1:
                t = torch.full((len(self.params),), -1, dtype=torch.int64, device=self.flat.device)
                if order:
                    t[:len(order)] = torch.tensor(order, dtype=torch.int64)
                dist.broadcast(t, 0, group=self.group)
                order = [i for i in t.tolist() if i >= 0]
            if order:
                hot = [self.params[i] for i in order]
                hot_set = set(hot)
                # slabs: only when every rank can agree on them without another collective — they are a property of the autograd
                # graph, like the order; a rank whose graph differed simply falls back to copies (sizes are checked per call)
                slabs = None
                if self._slab_rec and not self._arena_off and os.environ.get("LOTUS_DP_ARENA", "1") != "0":
                    slabs = ([n for _, n, _ in self._slab_rec], {index[p]: w for p, w in self._pslab.items()})
                if self.world > 1:  # rank 0's slabs, like rank 0's order: every rank must lay the flat buffer out identically
                    box = [slabs]
                    dist.broadcast_object_list(box, 0, group=self.group)
                    slabs = box[0]
                if slabs is not None:
                    slabs = (slabs[0], {self.params[i]: w for i, w in slabs[1].items()})
                self._layout(hot, [p for p in self.params if p not in hot_set], slabs)
                if slabs is not None:  # (served by call order and size; a rank whose nodes ask differently gets plain tensors)
                    self._slab_seq = [(n, self._slab_at.get(k)) for k, n in enumerate(slabs[0])]
                self._slab_rec, self._pslab = [], {}
                self._learning = False
                if self.sparse_hooks:
                    self._sparse_hooks(hot)
            self._arrival, self._seen = [], set()
        self._rearm()
        ops.GRAD_ARENA = None if self._arena_off else self._arena

    def finish(self):
        """Wait for the outstanding bucket all-reduces (call after backward, before the optimiser).  Afterwards
        `.grad` of every parameter is a view into the flat, rank-averaged buffer."""
        for b in range(self._next, len(self.buckets)):  # buckets some (or all) of whose parameters got no gradient
            if not self._flushed[b] and (self.world > 1 or self._force or self._pending[b] > 0 or
                                         any(p.grad is not None for p in self._bparams[b])):
                self._flush(b)
        self._next = len(self.buckets)
        timed = self.time_exposed and self.flat.is_cuda
        if timed:  # how long the backward stream sits behind the last bucket's all-reduce ("exposed" communication)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)
        if timed:
            e1.record()
            self.exposed_events.append((e0, e1))
        self._keep = []
        ops.GRAD_ARENA = None
        self.step_id += 1
        if self.world > 1 or self._force:
            self._sync_usage()
        self._used = self._used_bufs[self.step_id % 2]
        self._used_np = self._used.numpy()
        self._used_np[:] = 0

    def _sync_usage(self):
        """Which parameters did NO rank use in this step?  DistributedDataParallel(find_unused_parameters=True) answers with a
        host bitmap (genrobo3d/train/utils/distributed.py:196-205); here the per-parameter flags are MAX-reduced on the device
        (issued unconditionally: a rank-local shortcut would mismatch the collective) and stay there:

          * `used_mask` (device int32, one flag per parameter of `self.params`) is EXACT for the step just finished and is what
            the fused optimiser consumes (optim.AdamW.step(reducer=...) -> the `used` argument of lotus_adamw_step): a parameter
            whose global usage flips is skipped / updated in the very step it flips in, with no host synchronisation
            (ADVICE r3 / VERDICT r5 item 1c);
          * the HOST learns the flags one step late through a pinned copy (only the very first step blocks once) and uses them for
            what cannot matter numerically: `.grad = None` on parameters nobody used — what one GPU and DDP show, and what a
            non-fused optimiser keys on — and the optimiser's per-parameter step counters (`unused_of`)."""
        dev = self.flat.device
        u = self._used.to(dev, non_blocking=True)
        if self._lane_main is not None and u.is_cuda:
            self._lane_main.all_reduce(u, NativeComm.MAX)
        elif not _DIAG_NO_MSG:
            dist.all_reduce(u, op=dist.ReduceOp.MAX, group=self.group)      # stream-ordered on RCCL, blocking on gloo
        self.used_mask = u
        k = self.step_id
        if self._unused is None or not u.is_cuda:                          # first step (one blocking read) / host tensors
            self._unused = {i for i, f in enumerate(u.tolist()) if not f}
            self._known[k] = self._unused
        else:
            if self._usage_pending is not None:                           # last step's flags (long finished)
                host, ev, kprev = self._usage_pending
                _capi.wait_event(ev)  # (a query loop, never Event.synchronize(): that would wait for the tail of the training stream)
                self._unused = {i for i, f in enumerate(host.tolist()) if not f}
                self._known[kprev] = self._unused
            else:
                host, ev = torch.empty(u.shape, dtype=u.dtype, pin_memory=True), torch.cuda.Event()  # reused every step
            host.copy_(u, non_blocking=True)
            ev.record()
            self._usage_pending = (host, ev, k)
        for old in [j for j in self._known if j < k - 4]:
            del self._known[old]
        for i in self._unused:
            self.params[i].grad = None

    def unused_of(self, step_id):
        """Indices (into self.params) of the parameters no rank used in finish() number `step_id`, or None while the host has
        not seen that step's flags yet (they arrive one step late)."""
        return self._known.get(step_id)

    def view_of(self, p):
        """The slice of the flat, rank-averaged buffer that holds p's gradient (zeros when no rank produced one)."""
        return self._view[p]

    _unused, _usage_pending = None, None
    copied_floats = inplace_floats = 0  # gradient elements packed by copy / born in their bucket slot (since construction)
    used_mask = None       # device int32 [len(params)]: 1 = some rank produced a gradient in the step just finished (exact)
    time_exposed = False   # bench.py: record an event pair around the wait in finish()
    sparse_hooks = True    # one hook per bucket once the arrival order is known (False: a hook on every parameter)

    def exposed_comm_ms(self):
        """Mean GPU time per step the backward stream spent waiting in finish() (time_exposed = True), then reset."""
        ev, self.exposed_events = self.exposed_events, []  # (per instance, set in __init__: two reducers never mix their events)
        if not ev:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)


_BN_GROUP = None


def enable_sync_batchnorm(group=None):
    """SyncBatchNorm semantics: batch statistics over the points of ALL ranks.  One fused message
    per BN layer and direction: (sum, sumsq | sum dz, sum dz*xhat) + count, in fp64.

    The statistics travel on their OWN communicator (collective call: every rank must enter): the messages are on the
    critical path of forward and backward, and on the communicator of the gradient buckets they would queue behind a
    32+ MB all-reduce that is itself waiting for lagging weight gradients.  Over RCCL that communicator is the native "main"
    lane (NativeComm): each message is one RCCL kernel IN the training stream, between the statistics kernel that produces
    the sums and the apply kernel that consumes them."""
    global _BN_GROUP
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and os.environ.get("LOTUS_FORCE_COLLECTIVES") != "1"):
        ops.BnState.reduce = None
        return
    lane = native_comm(group, "main")  # (collective) the statistics as RCCL kernels of the training stream itself
    if group is None and lane is None:  # (availability of the lane is agreed by all ranks: so is this branch)
        if _BN_GROUP is None:
            _BN_GROUP = dist.new_group(backend=dist.get_backend())
        group = _BN_GROUP

    def send(sums):
        if lane is not None and sums.is_cuda:
            lane.all_reduce(sums, NativeComm.SUM)
        else:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)  # in place, no host sync

    def reduce(sums):
        global BN_MESSAGES
        BN_MESSAGES += 1
        if _DIAG_NO_MSG:  # (diagnostic, bench only: the SyncBatchNorm code path without its messages)
            return
        if BN_TIMING is not None and sums.is_cuda:  # bench.py: event pair around every statistics message
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            send(sums)
            e1.record()
            BN_TIMING.append((e0, e1))
            return
        send(sums)

    ops.BnState.reduce = reduce


_AR_ON_COMM = os.environ.get("LOTUS_DP_AR_ON_COMM", "1") == "1"  # bucket all-reduces as launches of the communication stream itself
_DIAG_NO_PACK = os.environ.get("LOTUS_DIAG_NO_PACK") == "1"  # (diagnostic: bucket flushes without the pack copies — wrong gradients)
_DIAG_NO_MSG = os.environ.get("LOTUS_DIAG_NO_MESSAGES") == "1"  # one-rank rehearsal without the collectives: what the plumbing alone costs
BN_MESSAGES = 0  # SyncBN all-reduces issued by this process (bench.py reports the count per step)
BN_TIMING = None  # set to a list to collect (start, end) events of every SyncBN message (bench.py: syncbn_ms per step)


def shard_loss_scale(n_local, n_global, world):
    """Factor for a rank's loss when the ranks hold DIFFERENT numbers of clouds (e.g. shard_clouds on a batch that does not
    divide evenly): every loss of the model is a mean over the rank's clouds, and the reducer averages the ranks' gradients
    (DistributedDataParallel semantics, genrobo3d/train/utils/distributed.py:196-205), which weights a cloud on a small
    shard more than one on a large shard.  loss * n_local * world / n_global restores the gradient of the mean over ALL
    clouds; with equal shards the factor is 1."""
    return float(n_local) * float(world) / float(n_global)


def shard_clouds(counts, world):
    """Point-count balanced assignment of clouds to ranks (sort by size, snake order):
    per-rank time is proportional to the number of points, not clouds."""
    idx = sorted(range(len(counts)), key=lambda i: -counts[i])
    shards = [[] for _ in range(world)]
    for k, i in enumerate(idx):
        r = k % (2 * world)
        shards[r if r < world else 2 * world - 1 - r].append(i)
    return [sorted(s) for s in shards]

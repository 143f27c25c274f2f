"""Autograd operators over the C-ABI (csrc/liblotus_hip.so).

One `torch.autograd.Function` per sub-block of the reference model, each with a hand-written
backward that chains the HIP kernels (fused epilogues, no temporaries beyond what backward needs):

  CpeFn        x + LN(Linear(SubMConv3d(xs)))                      Block.cpe    model.py:615-625,660-662
  SelfAttnFn   x + proj(patch_attention(qkv(LN(x))))               Block.attn   model.py:664-667,468-557
  FfnFn        x + fc2(GELU(fc1(LN(x))))                           Block.mlp / CABlock.mlp
  CrossAttnFn  x + proj(cross_attention(q(LN(x)), kv(context)))    CABlock.attn model_ca.py:135-140
  StemFn / PoolFn / UnpoolFn / HeadLossFn                          Embedding, SerializedPooling,
                                                                   SerializedUnpooling, ActionHead+loss
PyTorch provides device memory, the stream and the autograd graph; all arithmetic is in the kernels.
"""
import os

import torch

from . import _capi
from ._capi import call, query, WS

ACT_NONE, ACT_GELU, ACT_LEAKY = 0, 1, 2
BN_EPS, BN_MOMENTUM = 1e-3, 0.01
CALL_LOG = None  # bench.py sets this to a list to record the (kind, M, N, K) of every dense launch
EVENT_LOG = None  # ... and this to a list to get (kind, M, N, K, start event, end event) of every dense launch, recorded on
                  # the stream the launch went to (the timed quantities of bench.py's in-step roofline figure)


class _Timed:
    """Bracket a dense launch with HIP events on the stream it is enqueued on (diagnostic; only when EVENT_LOG is set)."""

    def __init__(self, key):
        self.key = key

    def __enter__(self):
        if EVENT_LOG is None:
            return self
        # inside a weight-gradient block the launches go to the side stream although torch's current stream is unchanged
        self.st = _SIDES[_CUR][0] if _capi.STREAM_OVERRIDE else torch.cuda.current_stream()
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record(self.st)
        return self

    def __exit__(self, *a):
        if EVENT_LOG is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(self.st)
            EVENT_LOG.append(self.key + (self.e0, e1))


# Weight gradients are off the critical path of backward (nothing downstream consumes them before the
# optimiser / gradient all-reduce) and their kernels are latency-bound streams over dy and x, while the
# dgrad chain is MFMA-bound: running them on a second HIP stream lets the two overlap on the CUs.
# Fork/join per autograd node: the side stream waits for the main stream before each weight-gradient launch
# and every backward() ends with the main stream waiting for the side stream (`_joined`), so whatever
# consumes the returned gradients (accumulation, the GradReducer hooks, the optimiser) is ordered after them.
# On by default; LOTUS_SIDE_STREAM=0 (or enable_side_stream(False)) keeps everything on one stream.
#
# set_wgrad_join("end") defers the join to ONE wait at the end of the backward pass (an autograd engine
# callback): the weight-gradient kernels of a node may then still be running while the main stream is
# several nodes further down the dgrad chain.  Valid when nothing reads a parameter gradient during
# backward — i.e. `.grad is None` on entry (zero_grad(set_to_none=True): autograd adopts the tensor without
# touching it) and no per-parameter hooks; the GradReducer (hooks) and gradient accumulation need the
# default "node".  Tensors the side stream reads are held until that join (_hold_for_side) so the caching allocator does
# not hand their memory to the main stream early.
SIDE = None
_SIDE_ON = os.environ.get("LOTUS_SIDE_STREAM", "1") != "0"
_STEM_WGRAD_MAIN = True  # the last weight gradient of a backward pass runs on the critical stream, idle by then (+0.45 %)
_JOIN = "node"
_END_CB_PENDING = False


_PRECISIONS = {"fp32": 0, "bf16": 1, "bf16x3": 3}
_M64 = (1 << 64) - 1


def mix_seed(seed, k=0):
    """splitmix64 finaliser of (seed, k): a well-mixed 64-bit sub-seed.  The device mask hash adds the low word of the
    seed to index * odd constant and XORs the high word in before its own finaliser (csrc/common.h lotus_hash32), so seeds
    that differ by small integers would give index-shifted copies of one stream; every dropout site therefore gets
    its seed through this function (layer, use site, step and rank all enter as `k` of a nested call)."""
    z = (int(seed) + 0x9E3779B97F4A7C15 * (int(k) + 1)) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def dist_rank():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank()
    return int(os.environ.get("RANK", "0"))


def _env_precision():
    v = os.environ.get("LOTUS_GEMM_PREC", "0")
    return int(v) if v in ("1", "3") else 0


_PREC = _env_precision()  # precision code the next forward pass captures (see set_gemm_precision / precision())


class storage:
    """Context manager: activation STORAGE type of the forward passes started inside.  torch.float32 (default) runs the
    lotus_* entry points; torch.bfloat16 runs their bf16-storage twins (lotus_b16_*, include/lotus_hip_b16.h): every
    activation tensor in HBM is bf16, parameters / parameter gradients / statistics / accumulation stay fp32 — the
    "bf16 activations, fp32 master weights" mode of BASELINE configs[4].  The operand precision inside is 'bf16'
    (one bf16 MFMA product, fp32 accumulate).  Like the precision, the storage type is captured per autograd node and
    replayed in its backward, so models of both kinds coexist in one process."""

    def __init__(self, dtype, shadows=False):
        assert dtype in (None, torch.float32, torch.bfloat16), dtype
        self.bf = dtype is torch.bfloat16
        # shadows: the dense layers read bf16 SHADOWS of their weights (WeightShadows) where a parameter carries one —
        # precision code 5 of the lotus_b16_linear_* entry points; master weights, gradients and the optimiser stay fp32
        self.code = 5 if (self.bf and shadows) else 1

    def __enter__(self):
        global _PREC
        self.prev = (_capi.BF16, _PREC)
        _capi.BF16 = self.bf
        if self.bf:
            _PREC = self.code
        return self

    def __exit__(self, *a):
        global _PREC
        _capi.BF16, _PREC = self.prev


def set_gemm_precision(mode):
    """Default operand precision of the dense, sparse-convolution and attention products for forward passes started
    from now on: 'fp32' = fp32 MFMA, exact products (default, the 1e-4 logit parity mode); 'bf16x3' = split-bf16 products
    hi*hi + hi*lo + lo*hi with fp32 accumulation (~2^-17 per product; measured max logit error 2.2e-5, +23-27 % step
    throughput); 'bf16' = bf16 operands (the bf16 compute mode of BASELINE configs[4]; +20-35 %).

    Host-side only: the precision is an argument of every C-ABI call (no state in the library).  Each autograd node
    captures it in forward and replays it in backward, and a model with a `gemm_precision` attribute overrides this
    default for its own forward passes (`with ops.precision(mode)`), so models of different precisions coexist."""
    global _PREC
    _PREC = _PRECISIONS[mode]


def get_gemm_precision():
    return {v_: k for k, v_ in _PRECISIONS.items()}[_PREC]


class precision:
    """Context manager: operand precision of the forward passes started inside (None = leave the default)."""

    def __init__(self, mode):
        self.code = None if mode is None else _PRECISIONS[mode]

    def __enter__(self):
        global _PREC
        self.prev = _PREC
        if self.code is not None:
            _PREC = self.code
        return self

    def __exit__(self, *a):
        global _PREC
        _PREC = self.prev


def set_wgrad_join(mode):
    global _JOIN
    assert mode in ("node", "end")
    sync_side_stream()
    _JOIN = mode


def enable_side_stream(on=True):
    global SIDE, _SIDE_ON
    _SIDE_ON = bool(on)
    if not on:
        sync_side_stream()
        SIDE = None
        _SIDES.clear()


_IN_NODE = 0  # > 0 while a `_joined` backward is running: only then is the join guaranteed


_NSIDE = 1     # one weight-gradient stream: a second one measured 766 vs 796 samples/s (round 1), 880 vs 930 (round 4)
_SIDES = []     # [(torch.cuda.Stream, hipStream_t)]: weight-gradient producers go round-robin over these
_RR = 0         # next side stream
_CUR = 0        # side stream of the _OnSide block being executed
_LINK = 0       # lotus_streamlink handle (event ring) used for every fork / join


def _side():
    global SIDE, _LINK
    if not _SIDE_ON or _IN_NODE == 0:
        return None
    if SIDE is None:
        # measured and not kept: more than one side stream (766 vs 796 samples/s), CU-masked side streams (side stream on 7/8,
        # 3/4 or 31/32 of the CUs: 923-931 / 929-932 / 875-879 against 930: reserving CUs for the critical stream buys
        # nothing), a lowest-priority side stream (+0.1 %)
        for _ in range(_NSIDE):
            st = _capi.step_stream("side") if not _SIDES else torch.cuda.Stream(priority=_capi.STREAM_PRIORITY)
            _SIDES.append((st, st.cuda_stream))
        if not _LINK:
            _LINK = query("lotus_streamlink_create", 256)
            if not _LINK:
                raise _capi.LotusError("lotus_streamlink_create failed: " + _capi.lib().last_error())
        SIDE = _SIDES[0][0]
    return SIDE


def sync_side_stream(target=None):
    """The current stream (or the stream with raw handle `target`) waits for everything enqueued on the
    weight-gradient stream(s)."""
    if SIDE is not None:
        cur = _capi.stream_ptr() if target is None else target
        for _, ptr in _SIDES:
            _capi.call_raw("lotus_streamlink_wait", _LINK, ptr, cur)


def _side_ws(nbytes, dev):
    """Workspace of the side-stream producer being enqueued (one per side stream); (re)allocated under that stream
    so the caching allocator orders its reuse after the stream's work."""
    slot = 16 + _CUR
    b = WS.buf.get((dev, slot))
    if b is None or b.numel() < nbytes:
        with torch.cuda.stream(_SIDES[_CUR][0]):
            b = WS.get(nbytes, dev, slot=slot)
    return b


class _OnSide:
    """Run a weight-gradient producer on a side stream after the main stream's pending work.  Outputs are
    allocated by the caller on the main stream (they stay alive until after the join); `reads` are the tensors
    the side-stream kernels consume.  Inside the block every ops.call() enqueues on the side stream
    (_capi.STREAM_OVERRIDE); torch's current stream is left alone — nothing in these blocks launches ATen kernels —
    and the fork is one event record + stream wait through the C-ABI stream link."""

    def __init__(self, *reads):
        self.reads = reads

    def __enter__(self):
        global _RR, _CUR
        self.on = _side() is not None
        if not self.on:
            return self
        _CUR = _RR
        _RR = (_RR + 1) % _NSIDE
        side, ptr = _SIDES[_CUR]
        _capi.call_raw("lotus_streamlink_wait", _LINK, _capi.stream_ptr(), ptr)
        if _JOIN == "end":  # no join at the end of the node: the memory must not go back to the main stream early
            _hold_for_side(self.reads, side)
        _capi.STREAM_OVERRIDE = ptr
        return self

    def __exit__(self, *a):
        if self.on:
            _capi.STREAM_OVERRIDE = 0


_HELD = []  # tensors the weight-gradient stream reads, kept alive until the end-of-backward join (deferred-join mode)


def _hold_for_side(reads, side):
    """Deferred-join mode: what the side stream reads must not be recycled by the main stream before the join.  Inside a
    backward pass (the end-of-backward callback is queued) the tensors are simply HELD until that join — `record_stream` would
    make the caching allocator answer every one of them with an event record on the side stream the moment autograd releases
    it: ~300 marker packets of ~5 us per step in between the weight-gradient kernels.  Outside a backward pass (weight packing in
    forward) there is no join to wait for: record_stream."""
    if _END_CB_PENDING:
        _HELD.extend(t for t in reads if t is not None)
    else:
        for t in reads:
            if t is not None:
                t.record_stream(side)


def _end_of_backward():
    global _END_CB_PENDING
    _END_CB_PENDING = False
    sync_side_stream()
    _HELD.clear()  # (freed behind the join: whatever re-uses the memory is enqueued after it)


def _abandoned_backward():
    """A forward pass starts while the end-of-backward callback of an earlier backward pass is still pending: that pass raised (the
    autograd engine drops its queued callbacks then — e.g. an out-of-memory error the trainer catches and skips).  Join the side
    stream and release what was held for it, or every later backward would append to _HELD for good and never join again
    (ADVICE r5)."""
    global _END_CB_PENDING
    _END_CB_PENDING = False
    sync_side_stream()
    _HELD.clear()


def _fwd(fn):
    """forward() decorator: the node remembers the operand precision it was computed in."""
    def wrapped(ctx, *args):
        if _END_CB_PENDING and _IN_NODE == 0:
            _abandoned_backward()
        ctx.prec, ctx.bf = _PREC, _capi.BF16
        return fn(ctx, *args)
    return staticmethod(wrapped)


def _joined(fn):
    """backward() decorator: join the side stream before the gradients leave the node ("node"), or once at the
    end of the backward pass ("end"); the node's products run in the precision its forward captured."""
    def wrapped(ctx, *grads):
        global _IN_NODE, _END_CB_PENDING, _PREC
        _IN_NODE += 1
        prev, _PREC = _PREC, getattr(ctx, "prec", _PREC)
        prev_bf, _capi.BF16 = _capi.BF16, getattr(ctx, "bf", _capi.BF16)
        try:
            if _JOIN == "end" and _SIDE_ON and not _END_CB_PENDING:
                _END_CB_PENDING = True
                torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
            return fn(ctx, *grads)
        finally:
            _PREC = prev
            _capi.BF16 = prev_bf
            _IN_NODE -= 1
            if _JOIN == "node":
                sync_side_stream()
    return staticmethod(wrapped)


def _ws(nbytes, dev):
    return WS.get(nbytes, dev, slot=0)


_COUNTERS = {}


def _counters(dev):
    """Zeroed arrival counters of the fused split-K products, one buffer per stream that launches them (the kernels
    leave them zero)."""
    return _counters_for(dev, _capi.STREAM_OVERRIDE or _capi.stream_ptr())


def _counters_for(dev, ptr):
    """The counter buffer of the stream with raw handle `ptr`.  First use: the zero-fill is enqueued ON that stream (a
    fill on torch's current stream would not be ordered before a side-stream launch whose fork event was already
    recorded: a counter that starts non-zero never reaches gridDim.z - 1 and the tile is never written)."""
    key = (dev, ptr)
    c = _COUNTERS.get(key)
    if c is None:
        owner = next((st for st, sp in _SIDES if sp == ptr), None)
        if owner is not None and ptr != _capi.stream_ptr():
            with torch.cuda.stream(owner):
                c = torch.zeros(query("lotus_splitk_counters_bytes"), dtype=torch.uint8, device=dev)
        else:
            c = torch.zeros(query("lotus_splitk_counters_bytes"), dtype=torch.uint8, device=dev)
        _COUNTERS[key] = c
    return c


def _pa(prec=None):
    """Operand precision for the attention / convolution entry points (and for dense layers given fp32 master weights):
    the shadow flag (5) only concerns the weight pointer of lotus_linear_fwd / _dgrad."""
    code = _PREC if prec is None else prec
    return 1 if code == 5 else code


def _wk(*ws):
    """-> (weights to hand to the kernels, precision code): in shadow mode (precision 5) and when EVERY given parameter
    carries a bf16 shadow (WeightShadows), the shadows and 5; else the fp32 masters and the plain code."""
    if _PREC == 5:
        sh = [getattr(w, "_lotus_b16", None) for w in ws]
        if all(t is not None for t in sh):
            return sh, 5
    return list(ws), _pa()


class WeightShadows:
    """bf16 shadows of a model's dense-layer weights: BASELINE configs[4] stores "bf16 activations / weights with fp32 master
    weights" (job_scripts/train_3dlotus_policy_peract.sh:42-44,61).  The fp32 Parameter stays what autograd, the optimiser
    and the checkpoint see; `p._lotus_b16` is what the bf16-storage products read.  The fused AdamW (optim.AdamW) rewrites
    the shadow in the launch that updates the master; any other in-place change of a master (another optimiser,
    load_state_dict, .copy_) bumps the tensor's version counter and refresh() recasts — one multi-tensor launch."""

    def __init__(self, params):
        self.params = [p for p in params if p.dtype == torch.float32 and p.numel() % 4 == 0]
        self.tables = None

    def invalidate(self):
        """Force the next refresh() to recast every shadow: for code that rewrites masters behind the version counter
        (`p.data.copy_()`, `p.data = ...`, a parameter swap, another raw-pointer kernel) — ADVICE r4."""
        for p in self.params:
            p._lotus_b16_ver = -1

    def refresh(self):
        stale = False
        for p in self.params:
            sh = getattr(p, "_lotus_b16", None)
            if sh is None or sh.device != p.device or sh.shape != p.shape:
                p._lotus_b16 = torch.empty(p.shape, dtype=torch.bfloat16, device=p.device)
                p._lotus_b16_ver = -1
                self.tables = None
            # a shadow is current for (version counter, storage address): `p.data = other` keeps the version but moves the storage
            if p._lotus_b16_ver != p._version or getattr(p, "_lotus_b16_ptr", None) != p.data_ptr():
                stale = True
        if not stale:
            return
        dev = self.params[0].device
        key = tuple(p.data_ptr() for p in self.params)
        if self.tables is None or self.tables[0] != key:
            import numpy as np
            chunk = query("lotus_mt_chunk")
            numel = [p.numel() for p in self.params]
            ch = [(t, c) for t, n in enumerate(numel) for c in range((n + chunk - 1) // chunk)]
            i64 = lambda v: torch.tensor(v, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)  # noqa: E731
            self.tables = (key, i64(list(key)), i64([p._lotus_b16.data_ptr() for p in self.params]), i64(numel),
                           torch.from_numpy(np.asarray(ch, dtype=np.int32)).pin_memory().to(dev, non_blocking=True), len(ch))
        _, src, dst, numel, chunks, nch = self.tables
        call("lotus_shadow_cast", src, dst, numel, chunks, nch)
        for p in self.params:
            p._lotus_b16_ver = p._version
            p._lotus_b16_ptr = p.data_ptr()


# Weight-gradient slabs of a backward node (the flat fp32 buffer its parameter gradients are views of).  A data-parallel reducer
# can own that memory: GRAD_ARENA(n_floats, device) -> a slice of its bucket buffer (or None), so that the gradients are born where
# the all-reduce reads them and the bucket flush has nothing to pack (parallel.GradReducer._arena).
GRAD_ARENA = None


def _grad_slab(n, dev):
    if GRAD_ARENA is not None:
        t = GRAD_ARENA(int(n), dev)
        if t is not None:
            return t
    return torch.empty(int(n), dtype=torch.float32, device=dev)


def _empty_like_rows(x, cols):
    return torch.empty(x.shape[0], cols, dtype=x.dtype, device=x.device)


# ------------------------------------------------------------------------------------ primitives
def _wprec(w, prec=None):
    """Precision code of a dense product from the weight tensor it is given: a bf16 weight IS a shadow (code 5), an fp32
    weight never is — a wrong pairing would make the kernel misread the matrix."""
    if w.dtype == torch.bfloat16:
        assert _capi.BF16, "bf16 weight shadows belong to the bf16-storage mode"
        return 5
    return _pa(prec)


def linear_fwd(x, w, b, residual=None, act=ACT_NONE, save_pre=False, drop_p=0.0, seed=0, prec=None):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=x.dtype, device=x.device)
    pre = torch.empty_like(y) if save_pre else None
    if CALL_LOG is not None:
        CALL_LOG.append(("fwd", M, N, K))
    nb = query("lotus_linear_workspace", M, N, K) if M <= 8192 else 0
    ws = _ws(nb, x.device) if nb else None
    with _Timed(("fwd", M, N, K)):
        call("lotus_linear_fwd", x, w, b, residual, y, pre, M, N, K, act, float(drop_p), int(seed),
             _wprec(w, prec), ws, nb, _counters(x.device) if nb else None)
    return y, pre


def linear_dgrad(dy, w, pre=None, add=None, act=ACT_NONE, drop_p=0.0, seed=0, prec=None):
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty(M, K, dtype=dy.dtype, device=dy.device)
    if CALL_LOG is not None:
        CALL_LOG.append(("dgrad", M, N, K))
    nb = query("lotus_linear_workspace", M, N, K) if M <= 8192 else 0
    ws = _ws(nb, dy.device) if nb else None
    with _Timed(("dgrad", M, N, K)):
        call("lotus_linear_dgrad", dy, w, dx, pre, add, M, N, K, act, float(drop_p), int(seed),
             _wprec(w, prec), ws, nb, _counters(dy.device) if nb else None)
    return dx


def linear_wgrad(dy, x, need_bias=True, prec=None, into=None):
    """dw = dy^T x, db = colsum(dy) on the weight-gradient stream.  `into` = (dw, db) of an earlier call: accumulate
    into them (ordered on that stream; adding the results on the main stream would race with the side stream)."""
    M, N = dy.shape
    K = x.shape[1]
    if into is None:
        buf = _grad_slab(N * K + (N if need_bias else 0), dy.device)
        dw = buf[:N * K].view(N, K)              # one contiguous gradient slab -> a single split-K reduce launch
        db = buf[N * K:] if need_bias else None
    else:
        dw, db = into
    if CALL_LOG is not None:
        CALL_LOG.append(("wgrad", M, N, K))
    nbytes = query("lotus_linear_wgrad_workspace", M, N, K)
    with _OnSide(dy, x):
        ws = _side_ws(nbytes, dy.device) if _side() is not None else WS.get(nbytes, dy.device, slot=0)
        with _Timed(("wgrad", M, N, K)):
            call("lotus_linear_wgrad", dy, x, dw, db, M, N, K, 0 if into is None else 1, _pa(prec), ws,
                 ws.numel(), _counters(dy.device))
    return dw, db


def ln_fwd(x, g, b, res=None, eps=1e-5, save=True):
    M, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device) if save else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if save else None
    call("lotus_layernorm_fwd", x, res, g, b, y, mean, rstd, M, C, float(eps))
    return y, mean, rstd


class Handoff:
    """Backward-pass hand-over between two consecutive sub-blocks of a stage.  Forward order: A -> B, where A ends
    with proj / fc2 dropout (p, seed).  B's backward finishes with a LayerNorm backward that produces dx — exactly the
    gradient A's backward receives, and the first thing A would do with it is dz = dropout_mask(p, seed) * dx in a kernel
    of its own.  B's LayerNorm backward writes dz as a second output instead and parks it here; A picks it up if the
    gradient it was handed is that very tensor (B was the only consumer of A's output)."""
    __slots__ = ("drop", "ptr", "dz")

    def __init__(self):
        self.drop, self.ptr, self.dz = None, 0, None

    def arm(self, p, seed):      # A.forward: this is the mask my backward will need
        self.drop, self.ptr, self.dz = ((float(p), int(seed)) if p > 0.0 else None), 0, None

    def take(self, dy):          # A.backward
        dz, self.dz = self.dz, None
        return dz if (dz is not None and self.ptr == dy.data_ptr() and dz.shape == dy.shape) else None


def _masked(dy, p, seed, hand):
    """dz = dropout mask * dy for a backward pass: from the hand-over if the producer of dy left it, else a launch."""
    if hand is not None:
        dz = hand.take(dy)
        if dz is not None:
            return dz
    return dropout(dy, p, seed)


def ln_bwd(dy, x, mean, rstd, g, add=None, hand=None):
    M, C = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty(C, dtype=torch.float32, device=x.device)
    dz, dp, dseed = None, 0.0, 0
    if hand is not None and hand.drop is not None:
        dp, dseed = hand.drop
        dz = torch.empty_like(x)
        hand.ptr, hand.dz = dx.data_ptr(), dz
    nbytes = query("lotus_layernorm_bwd_workspace", M, C)
    if _side() is None:
        ws = _ws(nbytes, x.device)
        call("lotus_layernorm_bwd", dy, x, mean, rstd, g, add, dx, dg, db, M, C, 0, dz, dp, dseed, ws, ws.numel())
        return dx, dg, db
    # dx on the main stream; the parameter-gradient reduction of the column partials (own workspace slot, joined
    # at the end of the node like every other weight gradient) on the side stream
    # ("end" join: the partials must outlive an unknown amount of main-stream progress -> a fresh buffer)
    ws = WS.get(nbytes, x.device, slot=4) if _JOIN == "node" else torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    call("lotus_layernorm_bwd", dy, x, mean, rstd, g, add, dx, None, None, M, C, 0, dz, dp, dseed, ws, ws.numel())
    with _OnSide(ws):
        call("lotus_layernorm_bwd_params", ws, M, C, dg, db, 0)
    return dx, dg, db


def linear_dgrad_ln(dy, w, x, mean, rstd, g, add=None, drop=None):
    """dx = LN'(dy w; x, mean, rstd, g) + add, dz = dx * mask(drop = (p, seed)), dgamma, dbeta: the input gradient of a linear
    layer fused with the backward of the LayerNorm that feeds it (lotus_linear_dgrad_ln; one kernel where 128-wide tiles
    cover whole rows of many-row levels, the two launches otherwise).  Test / diagnostic entry: the model reaches it through
    the composite backward passes of csrc/blocks.cpp."""
    import numpy as np
    M, N = dy.shape
    K = w.shape[1]
    dx, dn = torch.empty_like(x), torch.empty_like(x)
    dg = torch.empty(K, dtype=torch.float32, device=x.device)
    db = torch.empty(K, dtype=torch.float32, device=x.device)
    dz = torch.empty_like(x) if drop else None
    dp, dseed = drop if drop else (0.0, 0)
    nbytes = query("lotus_layernorm_bwd_workspace", M, K)
    lws = _ws(nbytes, x.device)
    nb = query("lotus_linear_workspace", M, N, K) if M <= 8192 else 0
    ws = WS.get(nb, x.device, slot=5) if nb else None
    nparts = np.zeros(1, dtype=np.int32)
    call("lotus_linear_dgrad_ln", dy, w, x, mean, rstd, g, add, dx, dn, dz, float(dp), int(dseed), M, N, K, _pa(None), ws, nb,
         _counters(x.device) if nb else None, lws, lws.numel(), int(nparts.ctypes.data))
    call("lotus_layernorm_bwd_params_n", lws, int(nparts[0]), K, dg, db, 0)
    return dx, dg, db, dz, int(nparts[0])


def conv_weight_t(w, prec=None):
    """[cout, k,k,k, cin] -> packed MFMA weight fragments for the forward and the input-gradient convolution
    (2 * w.numel() floats; layout in csrc/conv_pairs.hip); use them with the precision they were packed for."""
    cout, cin = w.shape[0], w.shape[-1]
    T = w.numel() // (cout * cin)
    wt = torch.empty(2 * w.numel(), dtype=torch.float32, device=w.device)
    call("lotus_conv_weight_transpose", w, wt, cout, T, cin, _pa(prec))
    return wt


_NO_PACK = {}


def conv_tap_active(lvl, C):
    """True when the 3^3 convolutions of width C at this level take the tap-grouped path of lotus_subm_conv (exact fp32
    products, fp32 storage, a tap plan on the level and an eligible shape): they read the module's own weight tensor, so
    their packed copy need not be produced."""
    return _PREC == 0 and not _capi.BF16 and getattr(lvl, "tap_plan", None) is not None and \
        query("lotus_conv_tap_eligible", lvl.n, C, C) == 1


def no_pack(dev):
    """Stand-in for the packed weights of a convolution that does not read them (null pointer at the C-ABI)."""
    t = _NO_PACK.get(dev)
    if t is None:
        t = _NO_PACK[dev] = torch.empty(0, dtype=torch.float32, device=dev)
    return t


def prepack_conv_weights(weights):
    """Pack the 3^3 convolution weights of a whole forward pass up front, on the weight-gradient stream when it is
    enabled: nine small launches leave the critical stream (they ran in front of every Block.cpe) and overlap the stem.
    The caller orders its stream after them with sync_side_stream() before the first use."""
    global _IN_NODE
    outs = [torch.empty(2 * w.numel(), dtype=torch.float32, device=w.device) for w in weights]
    _IN_NODE += 1  # (the side stream is otherwise reserved for backward nodes)
    try:
        with _OnSide(*weights):
            for w, o in zip(weights, outs):
                cout, cin = w.shape[0], w.shape[-1]
                call("lotus_conv_weight_transpose", w, o, cout, w.numel() // (cout * cin), cin, _pa())
    finally:
        _IN_NODE -= 1
    return outs


def _conv_ws(n, cin, cout, dev):
    nbytes = query("lotus_subm_conv_workspace", n, cin, cout)
    return WS.get(nbytes, dev, slot=2) if nbytes else None


def conv_fwd(x, w, b, nbr, rowidx, add=None, w_t=None, prec=None, tap_plan=None):
    n, cin = x.shape
    cout, T = w.shape[0], nbr.shape[0]
    y = torch.empty(n, cout, dtype=x.dtype, device=x.device)
    if T == 27:
        ws = _conv_ws(n, cin, cout, x.device)
    else:  # thin-input stem kernel: room for the transposed weights
        ws = WS.get(4 * T * cin * cout, x.device, slot=2) if cin <= 8 else None
    call("lotus_subm_conv", 0, x, w, w_t, b, add, y, nbr, rowidx, n, T, cin, cout, _pa(prec), tap_plan, ws,
         ws.numel() if ws is not None else 0)
    return y


def conv_dgrad(dy, w, nbr, rowidx, add=None, w_t=None, lvl=None, prec=None, tap_plan=None):
    """Input gradient of conv_fwd.  The kernel walks the mirrored taps of the forward table, which is the transposed
    pair list exactly when every voxel holds one point; `lvl.n_dup != 0` (augmented real clouds: 1-7 % of the points
    share a cell, SURVEY.md Trap 5) adds the fold / mask passes of lotus_conv_dup_* that make it the true gradient."""
    n, cout = dy.shape
    cin, T = w.shape[-1], nbr.shape[0]
    dups = lvl is not None and lvl.n_dup != 0
    if dups:
        dyr = torch.empty_like(dy)
        call("lotus_conv_dup_fold", dy, lvl.code[0], lvl.order[0], n, cout, dyr)
        dy = dyr
    dx = torch.empty(n, cin, dtype=dy.dtype, device=dy.device)
    ws = _conv_ws(n, cin, cout, dy.device) if T == 27 else None
    call("lotus_subm_conv", 1, dy, w, w_t, None, add, dx, nbr, rowidx, n, T, cin, cout, _pa(prec), tap_plan, ws,
         ws.numel() if ws is not None else 0)
    if dups:
        call("lotus_conv_dup_mask", dx, add, nbr[T // 2], n, cin)
    return dx


def conv_wgrad(dy, x, w_shape, nbr, need_bias=True, prec=None, side=True):
    """side=False keeps the launch on the current stream (the stem: the last weight gradient of a backward pass, when the
    critical stream has nothing left to do and the side stream still has a queue)."""
    n, cout = dy.shape
    cin, T = x.shape[1], nbr.shape[0]
    nw = cout * T * cin
    buf = _grad_slab(nw + (cout if need_bias else 0), dy.device)
    dw = buf[:nw].view(w_shape)
    db = buf[nw:] if need_bias else None
    nbytes = query("lotus_subm_conv_wgrad_workspace", n, T, cin, cout)
    if not side:
        ws = WS.get(nbytes, dy.device, slot=0)
        call("lotus_subm_conv_wgrad", dy, x, dw, db, nbr, n, T, cin, cout, 0, _pa(prec), ws, ws.numel())
        return dw, db
    with _OnSide(dy, x, nbr):
        ws = _side_ws(nbytes, dy.device) if _side() is not None else WS.get(nbytes, dy.device, slot=0)
        # thin-input stem (cin <= 8): one VALU kernel in every mode, exact fp32
        call("lotus_subm_conv_wgrad", dy, x, dw, db, nbr, n, T, cin, cout, 0, _pa(prec), ws, ws.numel())
    return dw, db


def dropout(x, p, seed):
    if p <= 0.0:
        return x
    y = torch.empty_like(x)
    call("lotus_dropout", x, y, x.numel(), float(p), int(seed))
    return y


def pos_targets(pc_fts, off, batch_idx, gt, nb, bin_size, kind="plain", robot_mask=None):
    """Soft position targets on the device (get_disc_gt_pos_prob, utils/action_position_utils.py:7-46), in the
    concatenated per-cloud [3][n_b * nb] layout the loss consumes.  pc_fts: f32 [N, >=3]; gt: f32 [B, >=3]."""
    N, B = pc_fts.shape[0], gt.shape[0]
    tgt = torch.empty(N * 3 * nb, dtype=torch.float32, device=pc_fts.device)
    ws = _ws(query("lotus_pos_workspace", B), pc_fts.device)
    rm = None if robot_mask is None else robot_mask.to(torch.uint8).contiguous()
    call("lotus_pos_targets", pc_fts, pc_fts.stride(0), off, batch_idx, gt, gt.stride(0), rm, B, N, nb, float(bin_size),
         {"plain": 0, "dist": 1}[kind], tgt, ws, ws.numel())
    return tgt


def pos_decode_max(xt, pc_fts, off, B, nb, bin_size):
    """get_best_pos_from_disc_pos(best='max') (utils/action_position_utils.py:48-64) for every cloud: f64 [B, 3]."""
    out = torch.empty(B, 3, dtype=torch.float64, device=xt.device)
    ws = _ws(query("lotus_pos_workspace", B), xt.device)
    call("lotus_pos_decode_max", xt, pc_fts, pc_fts.stride(0), off, B, nb, float(bin_size), out, ws, ws.numel())
    return out


def pos_decode_ens1(xt, pc_fts, counts, nb, bin_size):
    """get_best_pos_from_disc_pos(best='ens1') (utils/action_position_utils.py:66-85) for every cloud: f64 [B, 3].  An
    evaluation-time option of the reference (eval_simple_policy.py:63,83), evaluated on the host like there: per (cloud,
    axis) the softmax of the logits (on the device), then the probabilities summed per 5 mm cell in order of decreasing
    probability (sequential float32 accumulation, as the reference's dict of numpy scalars does it), the first cell with the
    strictly largest sum wins.  Vectorised: sort, np.add.at over cell ids in first-appearance order."""
    import numpy as np

    B = len(counts)
    pos_bins = nb // 2
    shift = np.arange(-pos_bins, pos_bins) * bin_size                      # float64, as the reference builds it
    out = np.zeros((B, 3), dtype=np.float64)
    xyz_all = pc_fts[:, :3].detach().float().cpu().numpy()
    lo = 0
    for b, n in enumerate(counts):
        lg = xt[lo:lo + n].detach().float().view(n, 3, nb).permute(1, 0, 2).reshape(3, n * nb)   # 'n (c b) -> c (n b)'
        prob = torch.softmax(lg, -1).cpu().numpy()
        xyz = xyz_all[lo:lo + n]
        for c in range(3):
            cands = (xyz[:, c:c + 1] + shift[None, :]).reshape(-1)          # float32 + float64 -> float64
            vox = np.round(cands / 0.005).astype(np.int32)
            order = np.argsort(-prob[c])
            v_s, p_s = vox[order], prob[c][order]
            uniq, first, inv = np.unique(v_s, return_index=True, return_inverse=True)
            acc = np.zeros(len(uniq), dtype=p_s.dtype)
            np.add.at(acc, inv, p_s)                                        # sequential, in visiting order
            visit = np.argsort(first, kind="stable")                        # cells in order of first appearance
            out[b, c] = int(uniq[visit[np.argmax(acc[visit])]]) * 0.005     # argmax: the first of the largest sums
        lo += n
    return torch.from_numpy(out).to(xt.device)


def sum_slabs(part):
    """part [G, ...] -> sum over G in fixed order (one launch; G == 1: the slab itself)."""
    if part.shape[0] == 1:
        return part[0]
    out = torch.empty_like(part[0])
    n = out.numel()
    call("lotus_sum_slabs", part, out, n, n, part.shape[0])
    return out


def drop_path(branch, x, p, seed):
    """x + DropPath(branch) (x may be None: the per-row mask alone, i.e. the backward map); timm.DropPath semantics, one
    draw per point row (model.py:655-657)."""
    M, C = branch.shape
    y = torch.empty_like(branch)
    call("lotus_drop_path", branch, x, y, M, C, float(p), int(seed))
    return y


def add(a, b):
    y = torch.empty_like(a)
    call("lotus_add", a, b, y, a.numel())
    return y


# Test-only taps of the two arg-max tables of the model (SerializedPooling's segment max, model.py:760-765, and the head's
# per-cloud max, simple_policy_ptv3.py:117-119).  ARG_TAP: a list that receives (kind, int32 table) in forward order.
# A third routing decision of the same kind ("leaky"): the sign pattern of the head's LeakyReLU(0.02) pre-activation
# (simple_policy_ptv3.py:42,113-115) — a pre-activation within rounding of zero gets slope 1 in one arithmetic and 0.02 in the
# other; the tap / injection is the saved pre-activation tensor itself (only its signs matter to backward).
# ARG_INJECT: a list of int32 tables consumed in the same order — the backward pass then routes the gradient of every
# (segment, channel) to the row the INJECTED table names while the forward values stay the kernel's own.  This is how
# tests/test_gpu_fullsize_oracle.py separates "a near-tie was broken the other way" (a discrete re-routing both fp32
# implementations are entitled to) from a defect: with the oracle's table injected every gradient must meet 1e-4.
ARG_TAP = None
ARG_INJECT = None


def _arg_hook(kind, arg):
    if ARG_TAP is not None:
        ARG_TAP.append((kind, arg))
    if ARG_INJECT:
        k2, inj = ARG_INJECT.pop(0)
        assert k2 == kind and tuple(inj.shape) == tuple(arg.shape), (k2, kind, tuple(inj.shape), tuple(arg.shape))
        return inj.to(device=arg.device, dtype=arg.dtype).contiguous()
    return arg


class BnState:
    """Batch statistics hook: `reduce(sums)` all-reduces the fp64 vector (sum, sumsq, count) across
    ranks when SyncBatchNorm semantics are wanted (train_simple_policy.py:116-117); None = local."""
    reduce = None


def _bn_stats(x, sums):
    M, C = x.shape
    ws = _ws(query("lotus_batchnorm_workspace", M, C), x.device)
    if _BN_FUSED and x.is_cuda and M > 0:  # one launch (last-arrival reduction), the sums alone
        call("lotus_batchnorm_stats_fused", x, sums, None, None, None, None, M, C, 0.0, 0.0, ws, ws.numel(), _bn_counter(x.device))
        return
    call("lotus_batchnorm_stats", x, sums, M, C, ws, ws.numel())


def _bn_finish(x, sums, g, b, rmean, rvar, training, act, momentum, eps):
    M, C = x.shape
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    if training and x.is_cuda:
        # statistics -> (message) -> apply: the apply pass finishes mean / invstd / running averages itself (SyncBatchNorm
        # forward: one launch less between the all-reduce and the consumer)
        y = torch.empty_like(x)
        call("lotus_batchnorm_apply_sums", x, sums, g, b, y, mean, invstd, rmean, rvar, M, C, act, float(eps), float(momentum))
        return y, mean, invstd
    if training:
        call("lotus_batchnorm_finalize", sums, mean, invstd, rmean, rvar, C, float(eps), float(momentum))
    else:
        call("lotus_batchnorm_eval_stats", rmean, rvar, mean, invstd, C, float(eps))
    y = torch.empty_like(x)
    call("lotus_batchnorm_apply", x, mean, invstd, g, b, y, M, C, act)
    return y, mean, invstd


_BN_FUSED = True  # one-launch BatchNorm statistics (two-level last-arrival reduction); False: the partials + reduce launches
_BN_CNT_OFF = None


def _bn_counter(dev):
    """The BatchNorm arrival counters of the stream being launched on (tail of the zeroed split-K counter buffer)."""
    global _BN_CNT_OFF
    if _BN_CNT_OFF is None:
        _BN_CNT_OFF = query("lotus_bn_counters_offset")
    return _counters(dev).data_ptr() + _BN_CNT_OFF


def bn_fwd(x, g, b, rmean, rvar, training, act, momentum=BN_MOMENTUM, eps=BN_EPS):
    sums = None
    if training and _BN_FUSED and BnState.reduce is None and x.shape[0] > 0:
        # local batch statistics: statistics, their reduction, mean / invstd and the running averages in ONE launch
        M, C = x.shape
        sums = torch.empty(2 * C + 1, dtype=torch.float64, device=x.device)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _ws(query("lotus_batchnorm_workspace", M, C), x.device)
        call("lotus_batchnorm_stats_fused", x, sums, mean, invstd, rmean, rvar, M, C, float(eps), float(momentum), ws, ws.numel(),
             _bn_counter(x.device))
        y = torch.empty_like(x)
        call("lotus_batchnorm_apply", x, mean, invstd, g, b, y, M, C, act)
        return y, mean, invstd
    if training:
        sums = torch.empty(2 * x.shape[1] + 1, dtype=torch.float64, device=x.device)
        _bn_stats(x, sums)
        if BnState.reduce is not None:
            BnState.reduce(sums)
    return _bn_finish(x, sums, g, b, rmean, rvar, training, act, momentum, eps)


def bn_fwd_pair(xa, pa, xb, pb, training, act, momentum=BN_MOMENTUM, eps=BN_EPS):
    """Two independent BatchNorms (the two branches of SerializedUnpooling) with ONE statistics message when
    SyncBatchNorm is on: the all-reduces are latency-bound, 26 -> 18 per step.  pa / pb = (g, b, rmean, rvar)."""
    if not training or BnState.reduce is None:
        return bn_fwd(xa, *pa, training, act, momentum, eps), bn_fwd(xb, *pb, training, act, momentum, eps)
    na, nb_ = 2 * xa.shape[1] + 1, 2 * xb.shape[1] + 1
    sums = torch.empty(na + nb_, dtype=torch.float64, device=xa.device)
    _bn_stats(xa, sums[:na])
    _bn_stats(xb, sums[na:])
    BnState.reduce(sums)
    return (_bn_finish(xa, sums[:na], *pa, training, act, momentum, eps),
            _bn_finish(xb, sums[na:], *pb, training, act, momentum, eps))


def _bn_bwd_stats(dy, x, mean, invstd, g, b, act, sums, want_params=False):
    """sums = (sum dz, sum dz * xhat, M) of the local rows; want_params: also (dgamma, dbeta) as fp32 vectors taken from those
    LOCAL sums (SyncBatchNorm: before the all-reduce overwrites them) — written by the statistics kernel itself on the GPU."""
    M, C = x.shape
    ws = _ws(query("lotus_batchnorm_workspace", M, C), x.device)
    if want_params and _BN_FUSED and M > 0 and x.is_cuda:
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        call("lotus_batchnorm_bwd_stats_fused_params", dy, x, mean, invstd, g, b, sums, dg, db, M, C, act, ws, ws.numel(), _bn_counter(x.device))
        return dg, db
    if want_params:
        _bn_bwd_stats(dy, x, mean, invstd, g, b, act, sums)
        return sums[C:2 * C].float(), sums[:C].float()
    if _BN_FUSED and M > 0:
        call("lotus_batchnorm_bwd_stats_fused", dy, x, mean, invstd, g, b, sums, M, C, act, ws, ws.numel(), _bn_counter(x.device))
    else:
        call("lotus_batchnorm_bwd_stats", dy, x, mean, invstd, g, b, sums, M, C, act, ws, ws.numel())


def _bn_bwd_apply(dy, x, mean, invstd, g, b, training, act, sums, reduced):
    # dgamma / dbeta are the LOCAL sums (gradient averaging across ranks is the reducer's job);
    # dx uses the statistics of the whole (all-rank) batch
    M, C = x.shape
    dx = torch.empty_like(x)
    if reduced is not None:  # (dg, db) were taken from the local sums before the all-reduce
        dg, db = reduced
        call("lotus_batchnorm_bwd_apply", dy, x, mean, invstd, g, b, sums, dx, None, None, M, C, act, 1, 0)
    else:  # local statistics: the apply kernel writes the parameter gradients itself
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        call("lotus_batchnorm_bwd_apply", dy, x, mean, invstd, g, b, sums, dx, dg, db, M, C, act,
             1 if training else 0, 0)
    return dx, dg, db


def bn_bwd(dy, x, mean, invstd, g, b, training, act):
    C = x.shape[1]
    sums = torch.empty(2 * C + 1, dtype=torch.float64, device=x.device)
    reduced = None
    if training and BnState.reduce is not None:
        reduced = _bn_bwd_stats(dy, x, mean, invstd, g, b, act, sums, want_params=True)
        BnState.reduce(sums)
    else:
        _bn_bwd_stats(dy, x, mean, invstd, g, b, act, sums)
    return _bn_bwd_apply(dy, x, mean, invstd, g, b, training, act, sums, reduced)


def bn_bwd_pair(a, bb, training, act):
    """Backward of bn_fwd_pair: a / bb = (dy, x, mean, invstd, g, b); one statistics message for both."""
    if not training or BnState.reduce is None:
        return bn_bwd(*a, training, act), bn_bwd(*bb, training, act)
    Ca, Cb = a[1].shape[1], bb[1].shape[1]
    na = 2 * Ca + 1
    sums = torch.empty(na + 2 * Cb + 1, dtype=torch.float64, device=a[1].device)
    sa, sb = sums[:na], sums[na:]
    ra = _bn_bwd_stats(*a, act, sa, want_params=True)
    rb = _bn_bwd_stats(*bb, act, sb, want_params=True)
    BnState.reduce(sums)
    return _bn_bwd_apply(*a, training, act, sa, ra), _bn_bwd_apply(*bb, training, act, sb, rb)


def attention_fwd(q, q_ld, q_off, kv, kv_ld, k_off, v_off, qidx, kidx, owner, tiles, ntiles, qn, kn, out, lse, H, d,
                  drop_p=0.0, seed=0, prec=None, k_max=0):
    call("lotus_attention_fwd", q, q_ld, q_off, kv, kv_ld, k_off, v_off, qidx, kidx, owner, tiles, ntiles,
         qn[0], qn[1], kn[0], kn[1], out, out.stride(0), lse, H, d, float(d ** -0.5), 1e-6, float(drop_p), int(seed),
         _pa(prec), int(k_max))


def attention_bwd(q, q_ld, q_off, kv, kv_ld, k_off, v_off, qidx, kidx, owner, tiles, blocks, nblocks, qn, kn, out,
                  dout, lse, dq, dq_ld, dq_off, dkv, dkv_ld, dk_off, dv_off, part_stride, atomic, H, d, drop_p=0.0, seed=0,
                  kext=None, ext_pos=None, n_extra=0, dkv_extra=None, prec=None, k_max=0):
    dev = q.device
    grads = [torch.empty(d, dtype=torch.float32, device=dev) for _ in range(4)]
    ws = _ws(query("lotus_attention_bwd_workspace", nblocks, H), dev)
    call("lotus_attention_bwd", q, q_ld, q_off, kv, kv_ld, k_off, v_off, qidx, kidx, owner, tiles, blocks, nblocks,
         qn[0], qn[1], kn[0], kn[1], out, dout, out.stride(0), lse, dq, dq_ld, dq_off, dkv, dkv_ld, dk_off, dv_off,
         part_stride, atomic, kext, ext_pos, n_extra, dkv_extra, grads[0], grads[1], grads[2], grads[3], 0, H, d, float(d ** -0.5), 1e-6, float(drop_p),
         int(seed), _pa(prec), int(k_max), ws, ws.numel())
    return grads


# ------------------------------------------------------------------------------------ composite fast path
# One C call per sub-block direction (csrc/blocks.cpp) instead of one per launch: same entry points, same order, same
# streams -> bit-identical results (tests/test_gpu_blocks.py), ~70 % less interpreter time inside the sub-blocks.  The
# per-launch bodies below stay as the readable reference and serve the diagnostics (bench.py's per-launch event log).
_COMPOSITE = os.environ.get("LOTUS_PY_BLOCKS", "0") != "1"
_SIZE_CACHE = {}


def composites_enabled():
    # (one weight-gradient stream: a handed-over dz is only known to be ordered on the stream its producer forked to)
    return _COMPOSITE and CALL_LOG is None and EVENT_LOG is None and _NSIDE == 1


def set_composites(on):
    global _COMPOSITE
    _COMPOSITE = bool(on)


def _sizes(kind, *dims):
    key = (kind, _capi.BF16) + dims
    v = _SIZE_CACHE.get(key)
    if v is None:
        if kind == "ffn":
            M, C, Hd = dims
            v = (query("lotus_ffn_saved_floats", M, C, Hd), query("lotus_ffn_grads_floats", C, Hd),
                 query("lotus_ffn_tmp_floats", M, C, Hd), query("lotus_ffn_ws_main_bytes", M, C, Hd),
                 query("lotus_ffn_ws_side_bytes", M, C, Hd))
        elif kind == "self":
            M, C, H, npad, nblocks, n_extra = dims
            v = (query("lotus_selfattn_saved_floats", M, C, H, npad), query("lotus_selfattn_grads_floats", C, H),
                 query("lotus_selfattn_tmp_floats", M, C, n_extra), query("lotus_selfattn_ws_main_bytes", M, C, H, nblocks),
                 query("lotus_selfattn_ws_side_bytes", M, C))
        elif kind == "cross":
            M, C, H, L, Cc, nblocks, G = dims
            v = (query("lotus_crossattn_saved_floats", M, C, H, L), query("lotus_crossattn_grads_floats", C, H, Cc),
                 query("lotus_crossattn_tmp_floats", M, C, L, G), query("lotus_crossattn_ws_main_bytes", M, C, H, L, Cc, nblocks),
                 query("lotus_crossattn_ws_side_bytes", M, C, L, Cc))
        elif kind == "crosskv":
            M, C, H, L, nblocks, G = dims
            v = (query("lotus_crossattn_kv_saved_floats", M, C, H), query("lotus_crossattn_kv_grads_floats", C, H),
                 query("lotus_crossattn_kv_tmp_floats", M, C, L, G), query("lotus_crossattn_kv_ws_main_bytes", M, C, H, nblocks),
                 query("lotus_crossattn_kv_ws_side_bytes", M, C))
        elif kind == "pair":
            M, C, H, Hd, npad, nst, n_extra, L, nca, G = dims
            v = (query("lotus_pair_acts_floats", M, C), query("lotus_pair_saved_floats", M, C, H, Hd, npad),
                 query("lotus_pair_grads_floats", C, H, Hd), query("lotus_pair_tmp_floats", M, C, Hd, n_extra, L, G),
                 query("lotus_pair_ws_main_bytes", M, C, H, Hd, nst, nca), query("lotus_pair_ws_side_bytes", M, C, Hd),
                 query("lotus_pair_ws_conv_bytes", M, C))
        elif kind == "cpe":
            n, C = dims
            v = (query("lotus_cpe_saved_floats", n, C), query("lotus_cpe_grads_floats", C), query("lotus_cpe_tmp_floats", n, C),
                 query("lotus_cpe_ws_main_bytes", n, C), query("lotus_cpe_ws_side_bytes", n, C), query("lotus_cpe_ws_conv_bytes", n, C))
        _SIZE_CACHE[key] = v
    return v


def _al4(n):
    return (n + 3) & ~3


def _side_ctx(dev, ws_side_bytes, reads, rows=0):
    """(side stream pointer or 0, its workspace, its counters) for a composite backward; in the deferred-join mode the
    tensors the side stream reads are announced to the allocator (as _OnSide does)."""
    global _CUR
    if _side() is None:  # (measured and not kept: the big levels' weight gradients on the critical stream, 907 / 883 / 861 vs 930)
        return 0, None, None
    _CUR = 0
    st, ptr = _SIDES[0]
    if _JOIN == "end":
        _hold_for_side(reads, st)
    ws = _side_ws(ws_side_bytes, dev)
    return ptr, ws, _counters_for(dev, ptr)


# ------------------------------------------------------------------------------------ sub-blocks
class CpeFn(torch.autograd.Function):
    """x1 = x + LN(Linear(SubMConv3d_3(xs))).  In the encoder xs is x; in the decoder xs is the
    stale proj_skip branch (SURVEY.md Trap 3), hence two tensor inputs."""

    @_fwd
    def forward(ctx, x, xs, cw, cb, lw, lb, g, b, lvl, wt=None):
        same = xs is x
        if wt is None:
            wt = conv_weight_t(cw)
        ctx.lvl, ctx.same = lvl, same
        n_, C = x.shape
        (lwk,), pc = _wk(lw)
        ctx.wk, ctx.pc = (lwk,), pc
        ctx.comp = composites_enabled() and C % 64 == 0 and (C == 64 or C % 128 == 0) and cw.shape[0] == C and cw.shape[-1] == C
        if ctx.comp:
            n_saved, _, _, ws_main, _, ws_conv = _sizes("cpe", n_, C)
            saved = torch.empty(n_saved, dtype=torch.float32, device=x.device)
            y = torch.empty_like(x)
            ws = _ws(ws_main, x.device)
            wc = WS.get(ws_conv, x.device, slot=2)
            _capi.call_raw("lotus_cpe_fwd", x, xs, cw, wt, cb, lwk, lb, g, b, y, saved, lvl.nbr27, lvl.order[0], lvl.tap_plan, n_, C, pc, ws,
                           ws.numel(), wc, wc.numel(), _counters(x.device), _capi.stream_ptr())
            ctx.save_for_backward(xs, cw, lw, g, saved, wt)
            return y
        c = conv_fwd(xs, cw, cb, lvl.nbr27, lvl.order[0], w_t=wt, tap_plan=lvl.tap_plan)
        l, _ = linear_fwd(c, lwk, lb)
        y, mean, rstd = ln_fwd(l, g, b, res=x)
        ctx.save_for_backward(xs, cw, lw, g, c, l, mean, rstd, wt)
        return y

    @_joined
    def backward(ctx, dy):
        lvl = ctx.lvl
        dy = dy.contiguous()
        if ctx.comp:
            xs, cw, lw, g, saved, wt = ctx.saved_tensors
            n_, C = xs.shape
            dev = xs.device
            _, n_grads, n_tmp, ws_main, ws_side, ws_conv = _sizes("cpe", n_, C)
            grads = _grad_slab(n_grads, dev)
            tmp = torch.empty(n_tmp, dtype=torch.float32, device=dev)
            dxc = torch.empty_like(xs)
            side, wss, cs = _side_ctx(dev, ws_side, (saved, tmp, xs, lvl.nbr27), n_)
            wsm = _ws(ws_main if side else max(ws_main, ws_side), dev)  # no side stream: the weight gradients use it too
            wc = WS.get(ws_conv, dev, slot=2)
            _capi.call_raw("lotus_cpe_bwd", dy, xs, cw, wt, ctx.wk[0], g, saved, dxc, 1 if ctx.same else 0, grads, tmp, lvl.nbr27,
                           lvl.order[0], lvl.tap_plan, lvl.code[0], lvl.n_dup, n_, C, ctx.pc, wsm, wsm.numel(), wc, wc.numel(), wss,
                           wss.numel() if wss is not None else 0, _counters(dev), cs, _LINK, 0, _capi.stream_ptr(), side)
            o1 = 2 * _al4(C)
            o2 = o1 + _al4(C * C + C)
            dg, db = grads[:C], grads[_al4(C):_al4(C) + C]
            dlw, dlb = grads[o1:o1 + C * C].view(C, C), grads[o1 + C * C:o1 + C * C + C]
            dcw, dcb = grads[o2:o2 + C * 27 * C].view(cw.shape), grads[o2 + C * 27 * C:o2 + C * 27 * C + C]
            if ctx.same:
                return dxc, None, dcw, dcb, dlw, dlb, dg, db, None, None
            return dy, dxc, dcw, dcb, dlw, dlb, dg, db, None, None
        xs, cw, lw, g, c, l, mean, rstd, wt = ctx.saved_tensors
        dl, dg, db = ln_bwd(dy, l, mean, rstd, g)
        dlw, dlb = linear_wgrad(dl, c)
        dc = linear_dgrad(dl, ctx.wk[0])
        dcw, dcb = conv_wgrad(dc, xs, cw.shape, lvl.nbr27)
        if ctx.same:  # d x = dy (residual) + conv dgrad
            dx = conv_dgrad(dc, cw, lvl.nbr27, lvl.order[0], add=dy, w_t=wt, lvl=lvl, tap_plan=lvl.tap_plan)
            return dx, None, dcw, dcb, dlw, dlb, dg, db, None, None
        dxs = conv_dgrad(dc, cw, lvl.nbr27, lvl.order[0], w_t=wt, lvl=lvl, tap_plan=lvl.tap_plan)
        return dy, dxs, dcw, dcb, dlw, dlb, dg, db, None, None


class FfnFn(torch.autograd.Function):
    """y = x + drop(fc2(drop(GELU(fc1(LN(x))))))   (MLP, model.py:577-583; pre-norm residual)."""

    @_fwd
    def forward(ctx, x, g, b, w1, b1, w2, b2, drop_p, seed, hand_in=None, hand_out=None, dpath=0.0):
        # dpath > 0: DropPath on the branch (Block.mlp only, training only; model.py:672) — its own elementwise pass, the
        # per-launch path, no hand-over of a pre-masked gradient (the published models train with drop_path 0)
        ctx.drop = (drop_p, seed)
        ctx.dpath = float(dpath)
        if dpath > 0.0:
            hand_in = None
        ctx.hands = (hand_in, hand_out)
        if hand_in is not None:
            hand_in.arm(drop_p, mix_seed(seed, 1))
        ctx.comp = composites_enabled() and x.shape[1] % 4 == 0 and dpath == 0.0
        (w1k, w2k), pc = _wk(w1, w2)
        ctx.wk, ctx.pc = (w1k, w2k), pc
        if ctx.comp:
            M, C = x.shape
            Hd = w1.shape[0]
            n_saved, _, _, ws_main, _ = _sizes("ffn", M, C, Hd)
            saved = torch.empty(n_saved, dtype=torch.float32, device=x.device)
            y = torch.empty_like(x)
            ws = _ws(ws_main, x.device)
            _capi.call_raw("lotus_ffn_fwd", x, g, b, w1k, b1, w2k, b2, y, saved, M, C, Hd, float(drop_p), int(seed),
                           mix_seed(seed, 1), pc, ws, ws.numel(), _counters(x.device), _capi.stream_ptr())
            ctx.save_for_backward(x, g, w1, w2, saved)
            return y
        n, mean, rstd = ln_fwd(x, g, b)
        a, hpre = linear_fwd(n, w1k, b1, act=ACT_GELU, save_pre=True, drop_p=drop_p, seed=seed)
        if dpath > 0.0:
            br, _ = linear_fwd(a, w2k, b2, drop_p=drop_p, seed=mix_seed(seed, 1))
            y = drop_path(br, x, dpath, mix_seed(seed, 5))
        else:
            y, _ = linear_fwd(a, w2k, b2, residual=x, drop_p=drop_p, seed=mix_seed(seed, 1))
        ctx.save_for_backward(x, g, w1, w2, n, hpre, a, mean, rstd)
        return y

    @_joined
    def backward(ctx, dy):
        p, seed = ctx.drop
        hand_in, hand_out = ctx.hands
        dy = dy.contiguous()
        if ctx.comp:
            x, g, w1, w2, saved = ctx.saved_tensors
            M, C = x.shape
            Hd, dev = w1.shape[0], x.device
            _, n_grads, n_tmp, ws_main, ws_side = _sizes("ffn", M, C, Hd)
            grads = _grad_slab(n_grads, dev)
            tmp = torch.empty(n_tmp, dtype=torch.float32, device=dev)
            dx = torch.empty_like(x)
            dz_in = hand_in.take(dy) if hand_in is not None else None
            dz_out, po, so = None, 0.0, 0
            if hand_out is not None and hand_out.drop is not None:
                po, so = hand_out.drop
                dz_out = torch.empty_like(x)
                hand_out.ptr, hand_out.dz = dx.data_ptr(), dz_out
            side, wss, cs = _side_ctx(dev, ws_side, (saved, tmp, dy, dz_in), M)
            wsm = _ws(ws_main if side else max(ws_main, ws_side), dev)  # no side stream: the weight gradients use it too
            _capi.call_raw("lotus_ffn_bwd", dy, dz_in, x, g, ctx.wk[0], ctx.wk[1], saved, dx, dz_out, po, so, grads, tmp, M, C, Hd, float(p),
                           int(seed), mix_seed(seed, 1), ctx.pc, wsm, wsm.numel(), wss, wss.numel() if wss is not None else 0,
                           _counters(dev), cs, _LINK, 0, _capi.stream_ptr(), side)
            o1 = 2 * _al4(C)
            o2 = o1 + _al4(Hd * C + Hd)
            return (dx, grads[:C], grads[_al4(C):_al4(C) + C], grads[o1:o1 + Hd * C].view(Hd, C), grads[o1 + Hd * C:o1 + Hd * C + Hd],
                    grads[o2:o2 + C * Hd].view(C, Hd), grads[o2 + C * Hd:o2 + C * Hd + C], None, None, None, None, None)
        x, g, w1, w2, n, hpre, a, mean, rstd = ctx.saved_tensors
        dyb = drop_path(dy, None, ctx.dpath, mix_seed(seed, 5)) if ctx.dpath > 0.0 else dy
        dz2 = _masked(dyb, p, mix_seed(seed, 1), hand_in)
        dw2, db2 = linear_wgrad(dz2, a)
        dh = linear_dgrad(dz2, ctx.wk[1], pre=hpre, act=ACT_GELU, drop_p=p, seed=seed)
        dw1, db1 = linear_wgrad(dh, n)
        dn = linear_dgrad(dh, ctx.wk[0])
        dx, dg, db = ln_bwd(dn, x, mean, rstd, g, add=dy, hand=hand_out)
        return dx, dg, db, dw1, db1, dw2, db2, None, None, None, None, None


class SelfAttnFn(torch.autograd.Function):
    """y = x + drop(proj(PatchAttention(qkv(LN(x)))))   (SerializedAttention flash path)."""

    @_fwd
    def forward(ctx, x, g, b, wqkv, bqkv, qnw, qnb, knw, knb, wp, bp, lvl, H, drop_p, seed, attn_p=0.0, hand_in=None, dpath=0.0):
        N, C = x.shape
        d = C // H
        ctx.meta = (lvl, H, d, drop_p, seed, attn_p)
        ctx.dpath = float(dpath)  # DropPath on the branch (model.py:666), see FfnFn
        if dpath > 0.0:
            hand_in = None
        ctx.hand_in = hand_in
        if hand_in is not None:
            hand_in.arm(drop_p, seed)
        ctx.comp = composites_enabled() and C % 4 == 0 and dpath == 0.0
        (wqkvk, wpk), pc = _wk(wqkv, wp)
        ctx.wk, ctx.pc = (wqkvk, wpk), pc
        if ctx.comp:
            n_saved, _, _, ws_main, _ = _sizes("self", N, C, H, lvl.npad, lvl.n_self_tiles, lvl.n_extra)
            saved = torch.empty(n_saved, dtype=torch.float32, device=x.device)
            y = torch.empty_like(x)
            ws = _ws(ws_main, x.device)
            _capi.call_raw("lotus_selfattn_fwd", x, g, b, wqkvk, bqkv, qnw, qnb, knw, knb, wpk, bp, y, saved, lvl.gidx, lvl.owner,
                           lvl.self_tiles, lvl.n_self_tiles, lvl.npad, N, C, H, float(d ** -0.5), float(drop_p), int(seed),
                           float(attn_p), mix_seed(seed, 1), pc, ws, ws.numel(), _counters(x.device), _capi.stream_ptr())
            ctx.save_for_backward(x, g, wqkv, qnw, qnb, knw, knb, wp, saved)
            return y
        n, mean, rstd = ln_fwd(x, g, b)
        qkv, _ = linear_fwd(n, wqkvk, bqkv)
        att = torch.empty(N, C, dtype=x.dtype, device=x.device)
        lse = torch.empty(lvl.npad, H, dtype=torch.float32, device=x.device)
        attention_fwd(qkv, 3 * C, 0, qkv, 3 * C, C, 2 * C, lvl.gidx, lvl.gidx, lvl.owner, lvl.self_tiles,
                      lvl.n_self_tiles, (qnw, qnb), (knw, knb), att, lse, H, d, attn_p, mix_seed(seed, 1))
        if dpath > 0.0:
            br, _ = linear_fwd(att, wpk, bp, drop_p=drop_p, seed=seed)
            y = drop_path(br, x, dpath, mix_seed(seed, 5))
        else:
            y, _ = linear_fwd(att, wpk, bp, residual=x, drop_p=drop_p, seed=seed)
        ctx.save_for_backward(x, g, wqkv, qnw, qnb, knw, knb, wp, n, qkv, att, lse, mean, rstd)
        return y

    @_joined
    def backward(ctx, dy):
        lvl, H, d, p, seed, attn_p = ctx.meta
        dy = dy.contiguous()
        if ctx.comp:
            x, g, wqkv, qnw, qnb, knw, knb, wp, saved = ctx.saved_tensors
            N, C = x.shape
            dev = x.device
            _, n_grads, n_tmp, ws_main, ws_side = _sizes("self", N, C, H, lvl.npad, lvl.n_self_tiles, lvl.n_extra)
            grads = _grad_slab(n_grads, dev)
            tmp = torch.empty(n_tmp, dtype=torch.float32, device=dev)
            dx = torch.empty_like(x)
            dz_in = ctx.hand_in.take(dy) if ctx.hand_in is not None else None
            side, wss, cs = _side_ctx(dev, ws_side, (saved, tmp, dy, dz_in), N)
            wsm = _ws(ws_main if side else max(ws_main, ws_side), dev)  # no side stream: the weight gradients use it too
            _capi.call_raw("lotus_selfattn_bwd", dy, dz_in, x, g, ctx.wk[0], qnw, qnb, knw, knb, ctx.wk[1], saved, dx, grads, tmp, lvl.gidx,
                           lvl.owner, lvl.self_tiles, lvl.self_blocks, lvl.n_self_tiles, lvl.kext, lvl.ext_pos, lvl.n_extra, lvl.npad,
                           N, C, H, float(d ** -0.5), float(p), int(seed), float(attn_p), mix_seed(seed, 1), ctx.pc, wsm, wsm.numel(),
                           wss, wss.numel() if wss is not None else 0, _counters(dev), cs, _LINK, 0, _capi.stream_ptr(), side)
            d4 = _al4(d)
            o1 = 2 * _al4(C)
            o2 = o1 + _al4(3 * C * C + 3 * C)
            o3 = o2 + 4 * d4
            return (dx, grads[:C], grads[_al4(C):_al4(C) + C], grads[o1:o1 + 3 * C * C].view(3 * C, C),
                    grads[o1 + 3 * C * C:o1 + 3 * C * C + 3 * C], grads[o2:o2 + d], grads[o2 + d4:o2 + d4 + d],
                    grads[o2 + 2 * d4:o2 + 2 * d4 + d], grads[o2 + 3 * d4:o2 + 3 * d4 + d], grads[o3:o3 + C * C].view(C, C),
                    grads[o3 + C * C:o3 + C * C + C], None, None, None, None, None, None, None)
        x, g, wqkv, qnw, qnb, knw, knb, wp, n, qkv, att, lse, mean, rstd = ctx.saved_tensors
        N, C = x.shape
        dyb = drop_path(dy, None, ctx.dpath, mix_seed(seed, 5)) if ctx.dpath > 0.0 else dy
        dz = _masked(dyb, p, seed, ctx.hand_in)
        dwp, dbp = linear_wgrad(dz, att)
        datt = linear_dgrad(dz, ctx.wk[1])
        # every (point, q|k|v column) is written exactly once by its owner position; the k/v gradients of the
        # borrowed tail-patch copies go to a small side buffer and are added afterwards (no atomics, no memset)
        dqkv = torch.empty(N, 3 * C, dtype=x.dtype, device=x.device)
        extra = torch.empty(max(lvl.n_extra, 1), 2 * C, dtype=x.dtype, device=x.device)
        gq, bq, gk, bk = attention_bwd(qkv, 3 * C, 0, qkv, 3 * C, C, 2 * C, lvl.gidx, lvl.gidx, lvl.owner,
                                       lvl.self_tiles, lvl.self_blocks, lvl.n_self_tiles, (qnw, qnb), (knw, knb), att,
                                       datt, lse, dqkv, 3 * C, 0, dqkv, 3 * C, C, 2 * C, 0, 0, H, d, attn_p, mix_seed(seed, 1),
                                       lvl.kext, lvl.ext_pos, lvl.n_extra, extra)
        dwqkv, dbqkv = linear_wgrad(dqkv, n)
        dn = linear_dgrad(dqkv, ctx.wk[0])
        dx, dg, db = ln_bwd(dn, x, mean, rstd, g, add=dy)
        return dx, dg, db, dwqkv, dbqkv, gq, bq, gk, bk, dwp, dbp, None, None, None, None, None, None, None


class CrossAttnFn(torch.autograd.Function):
    """y = x + drop(proj(CrossAttention(q(LN(x)), kv(context))))   (model_ca.py:46-101, :135-140)."""

    @_fwd
    def forward(ctx, x, context, g, b, wq, bq, wkv, bkv, qnw, qnb, knw, knb, wp, bp, lvl, H, drop_p, seed, attn_p=0.0,
                hand_in=None, hand_out=None):
        N, C = x.shape
        d = C // H
        ctx.meta = (lvl, H, d, drop_p, seed, attn_p)
        ctx.hands = (hand_in, hand_out)
        if hand_in is not None:
            hand_in.arm(drop_p, seed)
        ctx.comp = composites_enabled() and C % 4 == 0
        (wqk, wkvk, wpk), pc = _wk(wq, wkv, wp)
        ctx.wk, ctx.pc = (wqk, wkvk, wpk), pc
        if ctx.comp:
            L, Cc = context.shape
            n_saved, _, _, ws_main, _ = _sizes("cross", N, C, H, L, Cc, lvl.n_ca_blocks, lvl.ca_groups)
            saved = torch.empty(n_saved, dtype=torch.float32, device=x.device)
            y = torch.empty_like(x)
            ws = _ws(ws_main, x.device)
            _capi.call_raw("lotus_crossattn_fwd", x, context, g, b, wqk, bq, wkvk, bkv, qnw, qnb, knw, knb, wpk, bp, y, saved,
                           lvl.ca_tiles, lvl.n_ca_tiles, N, C, H, L, Cc, float(d ** -0.5), float(drop_p), int(seed), float(attn_p),
                           mix_seed(seed, 1), pc, lvl.ca_kmax, ws, ws.numel(), _counters(x.device), _capi.stream_ptr())
            ctx.save_for_backward(x, context, g, wq, wkv, qnw, qnb, knw, knb, wp, saved)
            return y
        n, mean, rstd = ln_fwd(x, g, b)
        q, _ = linear_fwd(n, wqk, bq)
        kv, _ = linear_fwd(context, wkvk, bkv)
        att = torch.empty(N, C, dtype=x.dtype, device=x.device)
        lse = torch.empty(N, H, dtype=torch.float32, device=x.device)
        attention_fwd(q, C, 0, kv, 2 * C, 0, C, None, None, None, lvl.ca_tiles, lvl.n_ca_tiles, (qnw, qnb), (knw, knb),
                      att, lse, H, d, attn_p, mix_seed(seed, 1), k_max=lvl.ca_kmax)
        y, _ = linear_fwd(att, wpk, bp, residual=x, drop_p=drop_p, seed=seed)
        ctx.save_for_backward(x, context, g, wq, wkv, qnw, qnb, knw, knb, wp, n, q, kv, att, lse, mean, rstd)
        return y

    @_joined
    def backward(ctx, dy):
        lvl, H, d, p, seed, attn_p = ctx.meta
        hand_in, hand_out = ctx.hands
        dy = dy.contiguous()
        if ctx.comp:
            x, context, g, wq, wkv, qnw, qnb, knw, knb, wp, saved = ctx.saved_tensors
            N, C = x.shape
            L, Cc = context.shape
            dev, G = x.device, lvl.ca_groups
            _, n_grads, n_tmp, ws_main, ws_side = _sizes("cross", N, C, H, L, Cc, lvl.n_ca_blocks, G)
            grads = _grad_slab(n_grads, dev)
            tmp = torch.empty(n_tmp, dtype=torch.float32, device=dev)
            dx = torch.empty_like(x)
            dctx = torch.empty_like(context) if ctx.needs_input_grad[1] else None
            dz_in = hand_in.take(dy) if hand_in is not None else None
            dz_out, po, so = None, 0.0, 0
            if hand_out is not None and hand_out.drop is not None:
                po, so = hand_out.drop
                dz_out = torch.empty_like(x)
                hand_out.ptr, hand_out.dz = dx.data_ptr(), dz_out
            side, wss, cs = _side_ctx(dev, ws_side, (saved, tmp, dy, dz_in, context), N)
            wsm = _ws(ws_main if side else max(ws_main, ws_side), dev)  # no side stream: the weight gradients use it too
            _capi.call_raw("lotus_crossattn_bwd", dy, dz_in, x, context, g, ctx.wk[0], ctx.wk[1], qnw, qnb, knw, knb, ctx.wk[2], saved, dx, dctx, dz_out,
                           po, so, grads, tmp, lvl.ca_tiles, lvl.ca_blocks, lvl.n_ca_blocks, G, N, C, H, L, Cc, float(d ** -0.5),
                           float(p), int(seed), float(attn_p), mix_seed(seed, 1), ctx.pc, lvl.ca_kmax, wsm, wsm.numel(), wss,
                           wss.numel() if wss is not None else 0, _counters(dev), cs, _LINK, 0, _capi.stream_ptr(), side)
            d4 = _al4(d)
            o1 = 2 * _al4(C)
            o2 = o1 + _al4(C * C + C)
            o3 = o2 + _al4(2 * C * Cc + 2 * C)
            o4 = o3 + 4 * d4
            return (dx, dctx, grads[:C], grads[_al4(C):_al4(C) + C], grads[o1:o1 + C * C].view(C, C), grads[o1 + C * C:o1 + C * C + C],
                    grads[o2:o2 + 2 * C * Cc].view(2 * C, Cc), grads[o2 + 2 * C * Cc:o2 + 2 * C * Cc + 2 * C], grads[o3:o3 + d],
                    grads[o3 + d4:o3 + d4 + d], grads[o3 + 2 * d4:o3 + 2 * d4 + d], grads[o3 + 3 * d4:o3 + 3 * d4 + d],
                    grads[o4:o4 + C * C].view(C, C), grads[o4 + C * C:o4 + C * C + C], None, None, None, None, None, None, None)
        x, context, g, wq, wkv, qnw, qnb, knw, knb, wp, n, q, kv, att, lse, mean, rstd = ctx.saved_tensors
        N, C = x.shape
        dev = x.device
        dz = _masked(dy, p, seed, hand_in)
        dwp, dbp = linear_wgrad(dz, att)
        datt = linear_dgrad(dz, ctx.wk[2])
        dq = torch.empty(N, C, dtype=x.dtype, device=dev)
        G, L = lvl.ca_groups, kv.shape[0]
        dkv_part = torch.empty(G, L, 2 * C, dtype=x.dtype, device=dev)
        gq, bq_, gk, bk_ = attention_bwd(q, C, 0, kv, 2 * C, 0, C, None, None, None, lvl.ca_tiles, lvl.ca_blocks,
                                         lvl.n_ca_blocks, (qnw, qnb), (knw, knb), att, datt, lse, dq, C, 0, dkv_part,
                                         2 * C, 0, C, L * 2 * C, 0, H, d, attn_p, mix_seed(seed, 1), k_max=lvl.ca_kmax)
        dkv = sum_slabs(dkv_part)
        dwkv, dbkv = linear_wgrad(dkv, context)
        dctx = linear_dgrad(dkv, ctx.wk[1]) if ctx.needs_input_grad[1] else None
        dwq, dbq = linear_wgrad(dq, n)
        dn = linear_dgrad(dq, ctx.wk[0])
        dx, dg, db = ln_bwd(dn, x, mean, rstd, g, add=dy, hand=hand_out)
        return dx, dctx, dg, db, dwq, dbq, dwkv, dbkv, gq, bq_, gk, bk_, dwp, dbp, None, None, None, None, None, None, None


class KvBank:
    """The keys / values of every CABlock of one forward pass, projected together.  `kv` [L, sum 2C] holds
    Linear(ctx -> 2C_b)(context) of block b in columns offs[b] : offs[b] + 2C_b; `dkv` (same shape, allocated by the first
    backward that needs it) collects d kv of every block, each CrossAttnKvFn.backward writing its own column slice."""
    __slots__ = ("kv", "dkv", "offs", "widths", "slices")

    def slice(self, b):
        return self.slices[b]

    def grad_slice(self, b):
        if self.dkv is None:
            self.dkv = torch.empty_like(self.kv)
        return self.dkv[:, self.offs[b]:self.offs[b] + self.widths[b]]


class KvAllFn(torch.autograd.Function):
    """kv_b = Linear(context; W_b, bias_b) for every CABlock b in ONE product (model_ca.py:46-67 evaluates the nine
    [L, 256] x [256, 2C_b] products one by one, each 3 row tiles tall): context [L, Cc] x cat(W_b) [sum 2C_b, Cc].  The weights
    stay the modules' own parameters (state_dict layout unchanged); their concatenation is one copy launch per step.
    Returns the column slices (views of one slab).  Backward: one input-gradient and one weight-gradient product over the
    shared d kv slab that the CrossAttnKvFn backward passes filled."""

    @_fwd
    def forward(ctx, context, bank, *wb):
        ws, bs = wb[0::2], wb[1::2]
        wks, _ = _wk(*ws)   # (bf16 storage with weight shadows: the concatenation of the shadows, half the copy)
        W = torch.cat(wks, 0)
        bias = torch.cat(bs, 0)
        kv, _ = linear_fwd(context, W, bias)
        bank.kv, bank.dkv = kv, None
        bank.widths = [w.shape[0] for w in ws]
        bank.offs = [0]
        for w_ in bank.widths[:-1]:
            bank.offs.append(bank.offs[-1] + w_)
        bank.slices = tuple(kv[:, o:o + w_] for o, w_ in zip(bank.offs, bank.widths))
        ctx.bank = bank
        ctx.save_for_backward(context, W)
        return bank.slices

    @_joined
    def backward(ctx, *gs):
        context, W = ctx.saved_tensors
        bank = ctx.bank
        if bank.dkv is None:
            bank.dkv = torch.empty_like(bank.kv)
        dkv = bank.dkv
        for b, g in enumerate(gs):  # normally every g IS its slice of the slab (written in place by the block's backward)
            sl = dkv[:, bank.offs[b]:bank.offs[b] + bank.widths[b]]
            if g is None:
                sl.zero_()
            elif g.data_ptr() != sl.data_ptr() or g.stride() != sl.stride():
                sl.copy_(g)
        dW, db = linear_wgrad(dkv, context)
        dctx = linear_dgrad(dkv, W) if ctx.needs_input_grad[0] else None
        out = [dctx, None]
        for o, w_ in zip(bank.offs, bank.widths):
            out += [dW[o:o + w_], db[o:o + w_]]
        bank.dkv = None
        return tuple(out)


class CrossAttnKvFn(torch.autograd.Function):
    """CrossAttnFn with the keys / values taken from a KvBank slice (projected once for all blocks by KvAllFn):
    y = x + drop(proj(CrossAttention(q(LN(x)), kv)))   (model_ca.py:46-101, :135-140)."""

    @_fwd
    def forward(ctx, x, kv, g, b, wq, bq, qnw, qnb, knw, knb, wp, bp, lvl, H, drop_p, seed, attn_p, hand_in, hand_out, bank, bidx):
        N, C = x.shape
        d = C // H
        ctx.meta = (lvl, H, d, drop_p, seed, attn_p, bank, bidx)
        ctx.hands = (hand_in, hand_out)
        if hand_in is not None:
            hand_in.arm(drop_p, seed)
        L, kv_ld = kv.shape[0], kv.stride(0)
        ctx.comp = composites_enabled() and C % 4 == 0
        (wqk, wpk), pc = _wk(wq, wp)
        ctx.wk, ctx.pc = (wqk, wpk), pc
        if ctx.comp:
            n_saved, _, _, ws_main, _ = _sizes("crosskv", N, C, H, L, lvl.n_ca_blocks, lvl.ca_groups)
            saved = torch.empty(n_saved, dtype=torch.float32, device=x.device)
            y = torch.empty_like(x)
            ws = _ws(ws_main, x.device)
            _capi.call_raw("lotus_crossattn_kv_fwd", x, kv, kv_ld, g, b, wqk, bq, qnw, qnb, knw, knb, wpk, bp, y, saved, lvl.ca_tiles,
                           lvl.n_ca_tiles, N, C, H, float(d ** -0.5), float(drop_p), int(seed), float(attn_p), mix_seed(seed, 1), pc,
                           lvl.ca_kmax, ws, ws.numel(), _counters(x.device), _capi.stream_ptr())
            ctx.save_for_backward(x, kv, g, wq, qnw, qnb, knw, knb, wp, saved)
            return y
        n, mean, rstd = ln_fwd(x, g, b)
        q, _ = linear_fwd(n, wqk, bq)
        att = torch.empty(N, C, dtype=x.dtype, device=x.device)
        lse = torch.empty(N, H, dtype=torch.float32, device=x.device)
        attention_fwd(q, C, 0, kv, kv_ld, 0, C, None, None, None, lvl.ca_tiles, lvl.n_ca_tiles, (qnw, qnb), (knw, knb),
                      att, lse, H, d, attn_p, mix_seed(seed, 1), k_max=lvl.ca_kmax)
        y, _ = linear_fwd(att, wpk, bp, residual=x, drop_p=drop_p, seed=seed)
        ctx.save_for_backward(x, kv, g, wq, qnw, qnb, knw, knb, wp, n, q, att, lse, mean, rstd)
        return y

    @_joined
    def backward(ctx, dy):
        lvl, H, d, p, seed, attn_p, bank, bidx = ctx.meta
        hand_in, hand_out = ctx.hands
        dy = dy.contiguous()
        dkv = bank.grad_slice(bidx)
        if ctx.comp:
            x, kv, g, wq, qnw, qnb, knw, knb, wp, saved = ctx.saved_tensors
            N, C = x.shape
            L, dev, G = kv.shape[0], x.device, lvl.ca_groups
            _, n_grads, n_tmp, ws_main, ws_side = _sizes("crosskv", N, C, H, L, lvl.n_ca_blocks, G)
            grads = _grad_slab(n_grads, dev)
            tmp = torch.empty(n_tmp, dtype=torch.float32, device=dev)
            dx = torch.empty_like(x)
            dz_in = hand_in.take(dy) if hand_in is not None else None
            dz_out, po, so = None, 0.0, 0
            if hand_out is not None and hand_out.drop is not None:
                po, so = hand_out.drop
                dz_out = torch.empty_like(x)
                hand_out.ptr, hand_out.dz = dx.data_ptr(), dz_out
            side, wss, cs = _side_ctx(dev, ws_side, (saved, tmp, dy, dz_in), N)
            wsm = _ws(ws_main if side else max(ws_main, ws_side), dev)
            _capi.call_raw("lotus_crossattn_kv_bwd", dy, dz_in, x, kv, kv.stride(0), g, ctx.wk[0], qnw, qnb, knw, knb, ctx.wk[1], saved, dx, dkv,
                           dkv.stride(0), dz_out, po, so, grads, tmp, lvl.ca_tiles, lvl.ca_blocks, lvl.n_ca_blocks, G, N, C, H, L,
                           float(d ** -0.5), float(p), int(seed), float(attn_p), mix_seed(seed, 1), ctx.pc, lvl.ca_kmax, wsm, wsm.numel(),
                           wss, wss.numel() if wss is not None else 0, _counters(dev), cs, _LINK, 0, _capi.stream_ptr(), side)
            d4 = _al4(d)
            o1 = 2 * _al4(C)
            o2 = o1 + _al4(C * C + C)
            o3 = o2 + 4 * d4
            return (dx, dkv, grads[:C], grads[_al4(C):_al4(C) + C], grads[o1:o1 + C * C].view(C, C), grads[o1 + C * C:o1 + C * C + C],
                    grads[o2:o2 + d], grads[o2 + d4:o2 + d4 + d], grads[o2 + 2 * d4:o2 + 2 * d4 + d], grads[o2 + 3 * d4:o2 + 3 * d4 + d],
                    grads[o3:o3 + C * C].view(C, C), grads[o3 + C * C:o3 + C * C + C]) + (None,) * 9
        x, kv, g, wq, qnw, qnb, knw, knb, wp, n, q, att, lse, mean, rstd = ctx.saved_tensors
        N, C = x.shape
        dev = x.device
        dz = _masked(dy, p, seed, hand_in)
        dwp, dbp = linear_wgrad(dz, att)
        datt = linear_dgrad(dz, ctx.wk[1])
        dq = torch.empty(N, C, dtype=x.dtype, device=dev)
        G, L = lvl.ca_groups, kv.shape[0]
        if G > 1:
            dkv_part = torch.empty(G, L, 2 * C, dtype=x.dtype, device=dev)
            tgt, tld, tps = dkv_part, 2 * C, L * 2 * C
        else:
            tgt, tld, tps = dkv, dkv.stride(0), 0
        gq, bq_, gk, bk_ = attention_bwd(q, C, 0, kv, kv.stride(0), 0, C, None, None, None, lvl.ca_tiles, lvl.ca_blocks,
                                         lvl.n_ca_blocks, (qnw, qnb), (knw, knb), att, datt, lse, dq, C, 0, tgt, tld, 0, C, tps, 0, H, d,
                                         attn_p, mix_seed(seed, 1), k_max=lvl.ca_kmax)
        if G > 1:
            call("lotus_sum_slabs_ld", dkv_part, dkv, L, 2 * C, dkv.stride(0), L * 2 * C, G)
        dwq, dbq = linear_wgrad(dq, n)
        dn = linear_dgrad(dq, ctx.wk[0])
        dx, dg, db = ln_bwd(dn, x, mean, rstd, g, add=dy, hand=hand_out)
        return (dx, dkv, dg, db, dwq, dbq, gq, bq_, gk, bk_, dwp, dbp) + (None,) * 9


# ------------------------------------------------------------------------------------ (Block, CABlock) pair in one call
# "auto" (default): the pair node is used where the host, not the GPU, sets the pace — bf16 storage, and every stage whose
# LEVEL holds fewer than ~40 k points (round 6: decided per stage, not per step: 7 of the 9 stages of the 16 x 4096 step, host
# floor 10.8 -> 8.0 ms per step at unchanged throughput, 1 083.3 against 1 083.4 samples/s — a process on a slow host stays
# GPU-bound) — and the five sub-block nodes on the large levels.  Measured on MI355X (A/B inside one call): the pair node lowers
# the host's enqueue time per step from 12.1 to 8.4 ms and lifts the PerAct bf16 step at 16 clouds from 1216-1356 to
# 1447-1452 samples/s, but costs the fp32 step at 16 x 4096 points 1.5 % (880 -> 866): its temporaries of all five
# sub-blocks are one allocation per pair, while separate nodes hand the same few hundred MB back to the allocator and get
# them again for the next sub-block (smaller live footprint in the 256 MB Infinity Cache / TLB).
_PAIR = {"0": False, "1": True}.get(os.environ.get("LOTUS_PAIR", "auto"), "auto")
_PAIR_AUTO_ROWS = 40000
_PP = None  # index tables of csrc/blocks.cpp: enum PairPtr / PairInt (kept in step by tests/test_gpu_round4.py)
_PP_NAMES = ("X XS KV Y ACTS SAVED CW CWP CB LW LB G0 B0 G1 B1 WQKV BQKV QNW QNB KNW KNB WP BP G2 B2 W1 B1F W2 B2F G3 B3 WQ BQ CQNW CQNB "
             "CKNW CKNB CWP2 CBP2 G4 B4 W3 B3F W4 B4F NBR27 ORDER0 TAPPLAN CODE0 GIDX OWNER STILES SBLOCKS KEXT EXTPOS CATILES CABLOCKS WS_MAIN "
             "WS_SIDE WS_CONV CNT_MAIN CNT_SIDE STREAM SIDE DY DX DXS DKV GRADS TMP").split()
_PI_NAMES = ("M C H HD NPAD NSTILES NEXTRA L NCATILES NCABLOCKS G KMAX NDUP SAME PREC KV_LD DKV_LD WS_MAIN WS_SIDE WS_CONV LINK SEED_SELF "
             "SEED_FFN1 SEED_CROSS SEED_FFN2").split()
# parameter slots of PairFn in call order -> their PairPtr names
_PAIR_PARAM_SLOTS = ("CW CB LW LB G0 B0 G1 B1 WQKV BQKV QNW QNB KNW KNB WP BP G2 B2 W1 B1F W2 B2F G3 B3 WQ BQ CQNW CQNB CKNW CKNB CWP2 CBP2 "
                     "G4 B4 W3 B3F W4 B4F").split()
_PAIR_LINEAR = (2, 8, 14, 18, 20, 24, 30, 34, 36)   # indices (in that order) of the dense-layer weights: candidates for bf16 shadows


def pair_enabled(rows, rows0=None):
    """rows: points of the level the stage runs on, rows0: points of the step's input level (the "auto" rule looks at them)."""
    if not composites_enabled():
        return False
    if _PAIR == "auto":
        if _capi.BF16:
            return True
        if BnState.reduce is not None or GRAD_ARENA is not None:
            # data parallel: the pair node hands its 38 gradients to the reducer at once, the buckets leave in bursts (one-rank
            # rehearsal 1 055-1 056 against 1 058-1 062 samples/s) — per step as in round 5
            return (rows if rows0 is None else rows0) <= _PAIR_AUTO_ROWS
        return rows <= _PAIR_AUTO_ROWS
    return _PAIR


def set_pair(on):
    """True / False, or "auto" (see above)."""
    global _PAIR
    _PAIR = on if on == "auto" else bool(on)


def _pair_tables():
    global _PP
    if _PP is None:
        import numpy as np
        assert query("lotus_pair_nptr") == len(_PP_NAMES) and query("lotus_pair_nint") == len(_PI_NAMES), \
            "ops._PP_NAMES / _PI_NAMES are out of step with enum PairPtr / PairInt of csrc/blocks.cpp"
        _PP = (np, {n: i for i, n in enumerate(_PP_NAMES)}, {n: i for i, n in enumerate(_PI_NAMES)},
               [_PP_NAMES.index(n) for n in _PAIR_PARAM_SLOTS])
    return _PP


def _pair_grad_sizes(C, H, Hd):
    key = ("pairgrads", C, H, Hd)
    v = _SIZE_CACHE.get(key)
    if v is None:
        d = C // H
        assert C % 4 == 0 and d % 4 == 0 and Hd % 4 == 0
        cpe = [C, C, C * C, C, 27 * C * C, C]
        att = [C, C, 3 * C * C, 3 * C, d, d, d, d, C * C, C]
        ffn = [C, C, Hd * C, Hd, C * Hd, C]
        ca = [C, C, C * C, C, d, d, d, d, C * C, C]
        v = _SIZE_CACHE[key] = cpe + att + ffn + ca + ffn
    return v


class PairFn(torch.autograd.Function):
    """Block i + CABlock i of a stage (model_ca.py:270-310) as ONE autograd node over csrc/blocks.cpp lotus_pair_fwd / _bwd:
    the launches of CpeFn -> SelfAttnFn -> FfnFn -> CrossAttnKvFn -> FfnFn in the same order on the same streams
    (bit-identical, tests/test_gpu_round4.py), one Python -> C transition and one set of buffers per direction.
    Inputs: x, xs (x itself in the encoder, the stale skip branch for the first Block of a decoder stage), the block's
    KvBank slice, the packed convolution weights, the 38 parameters in _PAIR_PARAM_SLOTS order, and a meta tuple."""

    @_fwd
    def forward(ctx, x, xs, kv, wt, *rest):
        np, PP, PI, slots = _pair_tables()
        params, meta = rest[:38], rest[38]
        lvl, lvl_ca, H, Hd, drop_p, attn_p, s_self, s_f1, s_cross, s_f2, bank, bidx = meta
        M, C = x.shape
        L = kv.shape[0]
        same = xs is x
        dev = x.device
        n_acts, n_saved, n_grads, n_tmp, ws_main, ws_side, ws_conv = _sizes(
            "pair", M, C, H, Hd, lvl.npad, lvl.n_self_tiles, lvl.n_extra, L, lvl_ca.n_ca_blocks, lvl_ca.ca_groups)
        acts = torch.empty(n_acts, dtype=torch.float32, device=dev)
        saved = torch.empty(n_saved, dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        ws = _ws(ws_main, dev)
        wc = WS.get(ws_conv, dev, slot=2)
        # dense-layer weights: bf16 shadows when this forward runs in shadow mode and every one of them has a shadow
        lin = [params[i] for i in _PAIR_LINEAR]
        link, pc = _wk(*lin)
        P = np.zeros(len(_PP_NAMES), dtype=np.uint64)
        I = np.zeros(len(_PI_NAMES), dtype=np.uint64)
        for slot, t in zip(slots, params):
            P[slot] = t.data_ptr()
        for i, t in zip(_PAIR_LINEAR, link):
            P[slots[i]] = t.data_ptr()
        P[PP["X"]], P[PP["XS"]], P[PP["KV"]], P[PP["Y"]] = x.data_ptr(), xs.data_ptr(), kv.data_ptr(), y.data_ptr()
        P[PP["ACTS"]], P[PP["SAVED"]], P[PP["CWP"]] = acts.data_ptr(), saved.data_ptr(), wt.data_ptr()
        P[PP["NBR27"]], P[PP["ORDER0"]], P[PP["CODE0"]] = lvl.nbr27.data_ptr(), lvl.order[0].data_ptr(), lvl.code[0].data_ptr()
        P[PP["TAPPLAN"]] = lvl.tap_plan.data_ptr() if lvl.tap_plan is not None else 0
        P[PP["GIDX"]], P[PP["OWNER"]], P[PP["STILES"]] = lvl.gidx.data_ptr(), lvl.owner.data_ptr(), lvl.self_tiles.data_ptr()
        P[PP["SBLOCKS"]], P[PP["KEXT"]], P[PP["EXTPOS"]] = lvl.self_blocks.data_ptr(), lvl.kext.data_ptr(), lvl.ext_pos.data_ptr()
        P[PP["CATILES"]], P[PP["CABLOCKS"]] = lvl_ca.ca_tiles.data_ptr(), lvl_ca.ca_blocks.data_ptr()
        P[PP["WS_MAIN"]], P[PP["WS_CONV"]] = ws.data_ptr(), wc.data_ptr()
        P[PP["CNT_MAIN"]], P[PP["STREAM"]] = _counters(dev).data_ptr(), _capi.stream_ptr()
        for k, v_ in (("M", M), ("C", C), ("H", H), ("HD", Hd), ("NPAD", lvl.npad), ("NSTILES", lvl.n_self_tiles),
                      ("NEXTRA", lvl.n_extra), ("L", L), ("NCATILES", lvl_ca.n_ca_tiles), ("NCABLOCKS", lvl_ca.n_ca_blocks),
                      ("G", lvl_ca.ca_groups), ("KMAX", lvl_ca.ca_kmax), ("NDUP", lvl.n_dup & 0xFFFFFFFF), ("SAME", 1 if same else 0),
                      ("PREC", pc), ("KV_LD", kv.stride(0)), ("WS_MAIN", ws.numel()), ("WS_CONV", wc.numel()),
                      ("SEED_SELF", s_self), ("SEED_FFN1", s_f1), ("SEED_CROSS", s_cross), ("SEED_FFN2", s_f2)):
            I[PI[k]] = v_
        F = np.array([float(drop_p), float(attn_p), float((C // H) ** -0.5)], dtype=np.float64)
        _capi.call_raw("lotus_pair_fwd", P.ctypes.data, I.ctypes.data, F.ctypes.data)
        ctx.save_for_backward(xs, kv, wt, acts, saved)
        ctx.keep = (params, link)   # parameters (leaves) and their shadows: referenced, not version-tracked
        ctx.tabs = (P, I, F)
        ctx.meta = (lvl, lvl_ca, H, Hd, same, bank, bidx, M, C, L)
        return y

    @_joined
    def backward(ctx, dy):
        np, PP, PI, slots = _pair_tables()
        xs, kv, wt, acts, saved = ctx.saved_tensors
        params, _ = ctx.keep
        lvl, lvl_ca, H, Hd, same, bank, bidx, M, C, L = ctx.meta
        P, I, F = ctx.tabs
        dev = xs.device
        dy = dy.contiguous()
        _, _, n_grads, n_tmp, ws_main, ws_side, ws_conv = _sizes(
            "pair", M, C, H, Hd, lvl.npad, lvl.n_self_tiles, lvl.n_extra, L, lvl_ca.n_ca_blocks, lvl_ca.ca_groups)
        grads = _grad_slab(n_grads, dev)
        tmp = torch.empty(n_tmp, dtype=torch.float32, device=dev)
        dx = torch.empty_like(dy)
        dxs = None if same else torch.empty_like(xs)
        dkv = bank.grad_slice(bidx)
        side, wss, cs = _side_ctx(dev, ws_side, (saved, acts, tmp, dy, xs, kv, lvl.nbr27), M)
        wsm = _ws(ws_main if side else max(ws_main, ws_side), dev)
        wc = WS.get(ws_conv, dev, slot=2)
        P[PP["DY"]], P[PP["DX"]], P[PP["DXS"]] = dy.data_ptr(), dx.data_ptr(), (0 if dxs is None else dxs.data_ptr())
        P[PP["DKV"]], P[PP["GRADS"]], P[PP["TMP"]] = dkv.data_ptr(), grads.data_ptr(), tmp.data_ptr()
        P[PP["WS_MAIN"]], P[PP["WS_CONV"]], P[PP["WS_SIDE"]] = wsm.data_ptr(), wc.data_ptr(), (0 if wss is None else wss.data_ptr())
        P[PP["CNT_MAIN"]], P[PP["CNT_SIDE"]] = _counters(dev).data_ptr(), (0 if cs is None else cs.data_ptr())
        P[PP["STREAM"]], P[PP["SIDE"]] = _capi.stream_ptr(), side
        I[PI["WS_MAIN"]], I[PI["WS_CONV"]], I[PI["WS_SIDE"]] = wsm.numel(), wc.numel(), (0 if wss is None else wss.numel())
        I[PI["DKV_LD"]], I[PI["LINK"]] = dkv.stride(0), _LINK
        _capi.call_raw("lotus_pair_bwd", P.ctypes.data, I.ctypes.data, F.ctypes.data)
        g = grads.split(_pair_grad_sizes(C, H, Hd))
        cw = params[0]
        out = (g[4].view(cw.shape), g[5], g[2].view(C, C), g[3], g[0], g[1],                                  # cpe
               g[6], g[7], g[8].view(3 * C, C), g[9], g[10], g[11], g[12], g[13], g[14].view(C, C), g[15],     # self-attention
               g[16], g[17], g[18].view(Hd, C), g[19], g[20].view(C, Hd), g[21],                               # mlp
               g[22], g[23], g[24].view(C, C), g[25], g[26], g[27], g[28], g[29], g[30].view(C, C), g[31],     # cross-attention
               g[32], g[33], g[34].view(Hd, C), g[35], g[36].view(C, Hd), g[37])                               # mlp
        return (dx, dxs, dkv, None) + out + (None,)


class StemFn(torch.autograd.Function):
    """Embedding: GELU(BN(SubMConv3d_5(x)))   (model.py:844-861; conv has no bias)."""

    @_fwd
    def forward(ctx, x, cw, g, b, rmean, rvar, lvl, training):
        c = conv_fwd(x, cw, None, lvl.nbr125, lvl.order[0])
        y, mean, invstd = bn_fwd(c, g, b, rmean, rvar, training, ACT_GELU)
        ctx.save_for_backward(x, cw, g, b, c, mean, invstd)
        ctx.meta = (lvl, training)
        return y

    @_joined
    def backward(ctx, dy):
        x, cw, g, b, c, mean, invstd = ctx.saved_tensors
        lvl, training = ctx.meta
        dc, dg, db = bn_bwd(dy.contiguous(), c, mean, invstd, g, b, training, ACT_GELU)
        # policy (no input gradient): this is the LAST weight gradient of the backward pass and the critical stream has
        # nothing after the BatchNorm backward above — run it here instead of queueing it behind the side stream's tail
        dcw, _ = conv_wgrad(dc, x, cw.shape, lvl.nbr125, need_bias=False, side=ctx.needs_input_grad[0] or not _STEM_WGRAD_MAIN)
        # the policy feeds raw point features (no gradient); the motion planner concatenates a learned label embedding
        dx = conv_dgrad(dc, cw, lvl.nbr125, lvl.order[0], lvl=lvl) if ctx.needs_input_grad[0] else None
        return dx, dcw, dg, db, None, None, None, None


class PoolFn(torch.autograd.Function):
    """SerializedPooling: GELU(BN(segment_max(Linear(x))))   (model.py:760-790)."""

    @_fwd
    def forward(ctx, x, w, bias, g, b, rmean, rvar, child, training):
        (wk,), _ = _wk(w)
        ctx.wk = wk
        proj, _ = linear_fwd(x, wk, bias)
        C = w.shape[0]
        pooled = torch.empty(child.n, C, dtype=x.dtype, device=x.device)
        arg = torch.empty(child.n, C, dtype=torch.int32, device=x.device)
        call("lotus_pool_max_fwd", proj, child.members, child.seg_start, child.n, C, pooled, arg)
        if ARG_TAP is not None or ARG_INJECT:
            arg = _arg_hook("pool", arg)
        y, mean, invstd = bn_fwd(pooled, g, b, rmean, rvar, training, ACT_GELU)
        ctx.save_for_backward(x, w, g, b, pooled, arg, mean, invstd)
        ctx.meta = (child, training)
        return y

    @_joined
    def backward(ctx, dy):
        x, w, g, b, pooled, arg, mean, invstd = ctx.saved_tensors
        child, training = ctx.meta
        C = w.shape[0]
        dpool, dg, db = bn_bwd(dy.contiguous(), pooled, mean, invstd, g, b, training, ACT_GELU)
        dproj = torch.empty(x.shape[0], C, dtype=x.dtype, device=x.device)
        call("lotus_pool_max_bwd", dpool, arg, child.cluster, x.shape[0], C, dproj)
        dw, dbias = linear_wgrad(dproj, x)
        dx = linear_dgrad(dproj, ctx.wk)
        return dx, dw, dbias, dg, db, None, None, None, None


class UnpoolFn(torch.autograd.Function):
    """SerializedUnpooling: skip = GELU(BN(Linear_skip(parent))), up = GELU(BN(Linear(point)));
    returns (skip + up[cluster], skip)   (model.py:817-828).  `skip` alone feeds the decoder CPE conv."""

    @_fwd
    def forward(ctx, xc, xp, wu, bu, gu, betau, rmu, rvu, ws_, bs, gs, betas, rms, rvs, child, training):
        (wuk, wsk), _ = _wk(wu, ws_)
        ctx.wk = (wuk, wsk)
        lu, _ = linear_fwd(xc, wuk, bu)
        ls, _ = linear_fwd(xp, wsk, bs)
        (up, mu, iu), (skip, ms, is_) = bn_fwd_pair(lu, (gu, betau, rmu, rvu), ls, (gs, betas, rms, rvs), training, ACT_GELU)
        x = torch.empty_like(skip)
        call("lotus_unpool_fwd", skip, up, child.cluster, skip.shape[0], skip.shape[1], x)
        ctx.save_for_backward(xc, xp, wu, gu, betau, ws_, gs, betas, lu, ls, mu, iu, ms, is_)
        ctx.meta = (child, training)
        return x, skip

    @_joined
    def backward(ctx, dx, dskip):
        xc, xp, wu, gu, betau, ws_, gs, betas, lu, ls, mu, iu, ms, is_ = ctx.saved_tensors
        child, training = ctx.meta
        C = wu.shape[0]
        dx = dx.contiguous()
        dup = torch.empty(child.n, C, dtype=dx.dtype, device=dx.device)
        call("lotus_unpool_bwd", dx, child.members, child.seg_start, child.n, C, dup)
        dsk = add(dx, dskip.contiguous()) if dskip is not None else dx
        (dlu, dgu, dbetau), (dls, dgs, dbetas) = bn_bwd_pair((dup, lu, mu, iu, gu, betau), (dsk, ls, ms, is_, gs, betas),
                                                            training, ACT_GELU)
        dwu, dbu = linear_wgrad(dlu, xc)
        dxc = linear_dgrad(dlu, ctx.wk[0])
        dws, dbs = linear_wgrad(dls, xp)
        dxp = linear_dgrad(dls, ctx.wk[1])
        return dxc, dxp, dwu, dbu, dgu, dbetau, None, None, dws, dbs, dgs, dbetas, None, None, None, None


class LinearFn(torch.autograd.Function):
    """Plain nn.Linear (txt_fc, simple_policy_ptv3.py:387,414)."""

    @_fwd
    def forward(ctx, x, w, b):
        y, _ = linear_fwd(x, w, b)
        ctx.save_for_backward(x, w)
        return y

    @_joined
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dw, db = linear_wgrad(dy, x, need_bias=ctx.needs_input_grad[2])
        dx = linear_dgrad(dy, w) if ctx.needs_input_grad[0] else None
        return dx, dw, (db if ctx.needs_input_grad[2] else None)


class HeadLossFn(torch.autograd.Function):
    """ActionHead (heatmap_disc / max / euler_disc) + compute_loss, simple_policy_ptv3.py:113-157,
    :308-373.  Returns (losses[4] = pos, rot, open, total; xt [N, 3*2*pos_bins]; ae [B, 217])."""

    @_fwd
    def forward(ctx, x, hw0, hb0, hw3, hb3, aw0, ab0, aw3, ab3, lvl, tgt, gt, pos_w, rot_w, drop_p, seed, with_loss):
        dev = x.device
        N, C = x.shape
        B = len(lvl.counts)
        h, hpre = linear_fwd(x, hw0, hb0, act=ACT_LEAKY, save_pre=True, drop_p=drop_p, seed=seed)
        xt, _ = linear_fwd(h, hw3, hb3)
        if ARG_TAP is not None or ARG_INJECT:
            hpre = _arg_hook("leaky", hpre)
        pc = torch.empty(B, C, dtype=x.dtype, device=dev)
        arg = torch.empty(B, C, dtype=torch.int32, device=dev)
        ws = _ws(query("lotus_cloud_max_workspace", B, C), x.device)
        call("lotus_cloud_max_fwd", x, lvl.off, B, C, pc, arg, ws, ws.numel())
        if ARG_TAP is not None or ARG_INJECT:
            arg = _arg_hook("cloud", arg)
        a, apre = linear_fwd(pc, aw0, ab0, act=ACT_LEAKY, save_pre=True, drop_p=drop_p, seed=mix_seed(seed, 1))
        ae, _ = linear_fwd(a, aw3, ab3)
        losses = torch.zeros(4, dtype=torch.float32, device=dev)
        nb = xt.shape[1] // 3
        nrot = (ae.shape[1] - 1) // 3
        stats = torch.empty(query("lotus_loss_stats_floats", B), dtype=torch.float32, device=dev)
        dae = torch.empty(ae.shape, dtype=torch.float32, device=dev)  # saved gradient factors stay fp32
        if with_loss:
            call("lotus_loss_fwd", xt, ae, tgt, gt, lvl.off, B, nb, nrot, gt.shape[1], float(pos_w), float(rot_w),
                 losses, stats, dae)
        ctx.save_for_backward(x, hw0, hw3, aw0, aw3, h, hpre, xt, pc, arg, a, apre, stats, dae, tgt)
        ctx.meta = (lvl, pos_w, rot_w, drop_p, seed, with_loss, nb, nrot)
        ctx.mark_non_differentiable(xt, ae)
        return losses, xt, ae

    @_joined
    def backward(ctx, gl, _gxt, _gae):
        x, hw0, hw3, aw0, aw3, h, hpre, xt, pc, arg, a, apre, stats, dae, tgt = ctx.saved_tensors
        lvl, pos_w, rot_w, p, seed, with_loss, nb, nrot = ctx.meta
        assert with_loss, "backward through the head requires compute_loss=True"
        dev = x.device
        N, C = x.shape
        B = len(lvl.counts)
        gl = gl.contiguous()
        dxt = torch.empty_like(xt)
        dae_o = torch.empty(dae.shape, dtype=xt.dtype, device=dev)
        call("lotus_loss_bwd", xt, tgt, lvl.off, lvl.batch, stats, dae, gl, float(pos_w), float(rot_w), B, N, nb, nrot,
             dxt, dae_o)
        # action branch (B rows)
        daw3, dab3 = linear_wgrad(dae_o, a)
        dapre = linear_dgrad(dae_o, aw3, pre=apre, act=ACT_LEAKY, drop_p=p, seed=mix_seed(seed, 1))
        daw0, dab0 = linear_wgrad(dapre, pc)
        dpc = linear_dgrad(dapre, aw0)
        # heatmap branch (N rows)
        dhw3, dhb3 = linear_wgrad(dxt, h)
        dhpre = linear_dgrad(dxt, hw3, pre=hpre, act=ACT_LEAKY, drop_p=p, seed=seed)
        dhw0, dhb0 = linear_wgrad(dhpre, x)
        dxh = linear_dgrad(dhpre, hw0)
        dx = torch.empty_like(x)
        call("lotus_cloud_max_bwd", dpc, arg, lvl.batch, N, C, dxh, dx)
        return (dx, dhw0, dhb0, dhw3, dhb3, daw0, dab0, daw3, dab3) + (None,) * 8


class StepHeadFn(torch.autograd.Function):
    """Heat-map branch of the trajectory head for ALL steps (motion_planner_ptv3.py:88-97,113-114):
    xt_t = Linear3(dropout(LeakyReLU(base + step_bias[t]))), t = 0..T-1, with `base` = the point-feature part of the first
    Linear (shared by the steps).  The hidden layers are never kept: backward regenerates them (same counter-hash masks),
    accumulates d base over the steps in the activation-backward kernel and sums the weight gradients of the T products."""

    @_fwd
    def forward(ctx, base, step_bias, w3, b3, drop_p, seed):
        T = step_bias.shape[0]
        M, C = base.shape
        outs = []
        for t in range(T):
            h = torch.empty_like(base)
            call("lotus_step_act_fwd", base, step_bias[t], h, M, C, ACT_LEAKY, float(drop_p), mix_seed(seed, t))
            outs.append(linear_fwd(h, w3, b3)[0])
        ctx.save_for_backward(base, step_bias, w3)
        ctx.meta = (float(drop_p), int(seed))
        return tuple(outs)

    @_joined
    def backward(ctx, *dxts):
        base, step_bias, w3 = ctx.saved_tensors
        p, seed = ctx.meta
        T = step_bias.shape[0]
        M, C = base.shape
        dbase = torch.empty_like(base)
        dsb = torch.empty_like(step_bias)
        dw3 = db3 = None
        ws = _ws(query("lotus_step_act_bwd_workspace", M, C), base.device)
        first = True
        keep = []  # everything the weight-gradient stream reads stays alive until the node's join (h_t and a contiguous
        #            copy of dxt_t would otherwise be recycled by the allocator while wgrad_t is still in flight)
        for t in range(T):
            if dxts[t] is None:
                dsb[t].zero_()
                continue
            dxt = dxts[t].contiguous()
            h = torch.empty_like(base)
            keep.append((dxt, h))
            if _JOIN == "end" and _side() is not None:
                dxt.record_stream(_SIDES[0][0]); h.record_stream(_SIDES[0][0])
            call("lotus_step_act_fwd", base, step_bias[t], h, M, C, ACT_LEAKY, p, mix_seed(seed, t))
            dw3, db3 = linear_wgrad(dxt, h, into=None if dw3 is None else (dw3, db3))
            dh = linear_dgrad(dxt, w3)
            call("lotus_step_act_bwd", dh, base, step_bias[t], dbase, dsb[t], M, C, ACT_LEAKY, p, mix_seed(seed, t),
                 0 if first else 1, ws, ws.numel())
            first = False
        if first:
            dbase.zero_()
        # `keep` dies with this frame: whatever re-uses the blocks is enqueued after the join `_joined` performs next
        return dbase, dsb, dw3, db3, None, None


class PosCEFn(torch.autograd.Function):
    """Soft-target heatmap cross entropy per (cloud, axis): F.cross_entropy(rearrange(pred, 'c n b -> c (n b)'),
    probs, reduction='none') of motion_planner_ptv3.py:329-334 for one trajectory step.  Returns [B, 3]."""

    @_fwd
    def forward(ctx, xt, tgt, lvl):
        B = len(lvl.counts)
        nb = xt.shape[1] // 3
        stats = torch.empty(query("lotus_loss_stats_floats", B), dtype=torch.float32, device=xt.device)
        call("lotus_pos_ce_fwd", xt, tgt, lvl.off, B, nb, stats)
        ctx.save_for_backward(xt, tgt, stats)
        ctx.lvl = lvl
        return stats[:B * 12].view(B, 3, 4)[:, :, 0].clone()

    @_joined
    def backward(ctx, g):
        xt, tgt, stats = ctx.saved_tensors
        lvl = ctx.lvl
        dxt = torch.empty_like(xt)
        call("lotus_pos_ce_bwd", xt, tgt, lvl.off, lvl.batch, stats, g.contiguous().float(), len(lvl.counts), xt.shape[0],
             xt.shape[1] // 3, dxt)
        return dxt, None, None


class TrajLossFn(torch.autograd.Function):
    """The [B, T]-sized losses of the motion planner in one launch (motion_planner_ptv3.py:327-397): masked rotation
    cross entropy, openness / stop BCE and the masked mean of the heatmap cross entropies ce [B, T, 3].  Returns the
    vector (pos, rot, open, stop, total)."""

    @_fwd
    def forward(ctx, ae, ce, gt, stop, mask, nrot, pos_w, rot_w):
        B, T = mask.shape
        ae, ce = ae.contiguous(), ce.contiguous().float()
        losses = torch.empty(5, dtype=torch.float32, device=ae.device)
        dae = torch.empty(ae.shape, dtype=torch.float32, device=ae.device)  # saved gradient factors stay fp32
        dce = torch.empty_like(ce)
        call("lotus_mp_loss_fwd", ae, gt, stop, mask, ce, B, T, nrot, gt.shape[-1], float(pos_w), float(rot_w), losses, dae, dce)
        ctx.save_for_backward(dae, dce)
        ctx.dims = (B, T, nrot, float(pos_w), float(rot_w))
        return losses

    @_joined
    def backward(ctx, g):
        dae, dce = ctx.saved_tensors
        B, T, nrot, pos_w, rot_w = ctx.dims
        dae_o = torch.empty(dae.shape, dtype=torch.bfloat16 if _capi.BF16 else torch.float32, device=dae.device)
        dce_o = torch.empty_like(dce)
        call("lotus_mp_loss_bwd", dae, dce, g.contiguous().float(), pos_w, rot_w, B, T, nrot, dae_o, dce_o)
        return dae_o, dce_o, None, None, None, None, None, None


class CloudMaxFn(torch.autograd.Function):
    """torch.stack([torch.max(x, 0)[0] for x in torch.split(feat, npoints_in_batch)]),
    motion_planner_ptv3.py:117-119 / simple_policy_ptv3.py:117-119."""

    @_fwd
    def forward(ctx, x, lvl):
        B, C = len(lvl.counts), x.shape[1]
        y = torch.empty(B, C, dtype=x.dtype, device=x.device)
        arg = torch.empty(B, C, dtype=torch.int32, device=x.device)
        ws = _ws(query("lotus_cloud_max_workspace", B, C), x.device)
        call("lotus_cloud_max_fwd", x, lvl.off, B, C, y, arg, ws, ws.numel())
        if ARG_TAP is not None or ARG_INJECT:
            arg = _arg_hook("cloud", arg)
        ctx.save_for_backward(arg)
        ctx.lvl, ctx.n = lvl, x.shape[0]
        return y

    @_joined
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        dx = torch.empty(ctx.n, dy.shape[1], dtype=dy.dtype, device=dy.device)
        call("lotus_cloud_max_bwd", dy.contiguous(), arg, ctx.lvl.batch, ctx.n, dy.shape[1], None, dx)
        return dx, None

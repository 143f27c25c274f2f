"""Host side of the integer front-end: one call builds every index table of a forward pass.

Mirrors, for all levels at once, what the reference computes piecemeal with dozens of host
synchronisations: `Point.serialization` + `Point.sparsify` (PointTransformerV3/model.py:83-176), the
index part of every `SerializedPooling.forward` (:713-772), `get_padding_and_inverse` (:410-466) and
spconv's neighbour lookup.  The device pipeline (csrc/front_end.hip) runs without a host sync; ONE
small device->host copy at the end returns the per-level point counts, after which the exactly
sized neighbour / patch / tile tables are built.
"""
import contextlib
import os

import numpy as np
import torch

from . import _capi
from ._capi import call, query, WS

# build the exact-size tables of a PREFETCHED front-end on the front-end stream (FrontEnd.finish)
SYNC_WAIT = None  # set to [0.0] to accumulate the host time FrontEnd.finish() waits for the device (diagnostic)
FINISH_ON_SIDE = True  # exact-size tables of a prefetched batch are built on the front-end stream (+0.6 % with fresh batches)

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")


class _Arena:
    """Bump allocator over ONE device block.  The front-end's tables are allocated under the front-end stream and consumed on
    the training stream, so each of them needs `record_stream` — and the caching allocator answers every such tensor, when it is
    freed at the end of the step, with an event record on the consuming stream: ~85 marker packets of ~4.7 us each between the
    last kernel of a step and the first of the next (0.4 ms of an idle GPU per step, measured with tools/tail_probe.py).  Carved
    out of one block per half of the front-end that is two events."""

    def __init__(self, nbytes, dev):
        self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        self.off = 0

    _SIZE = {torch.int32: 4, torch.int64: 8, torch.float32: 4, torch.uint8: 1, torch.int16: 2, torch.float64: 8}

    def empty(self, shape, dtype):
        n = self._SIZE[dtype]
        for d in shape:  # (plain Python: this runs ~70 times per step on the host's critical path)
            n *= int(d)
        off = (self.off + 255) & ~255
        if off + n > self.buf.numel():
            return None
        self.off = off + n
        return self.buf[off:off + n].view(dtype).view(*shape)


class Level:
    """Index tables of one resolution level (device tensors unless noted)."""
    __slots__ = ("n", "counts", "off", "off_host", "grid", "batch", "code", "order", "inverse", "depth", "nbr27",
                 "nbr125", "gidx", "owner", "kext", "ext_pos", "n_extra", "npad", "self_tiles", "self_blocks", "n_self_tiles", "ca_tiles",
                 "ca_blocks", "n_ca_tiles", "n_ca_blocks", "ca_groups", "cluster", "seg_start", "members",
                 "coord", "parent", "n_dup", "patch", "_views", "ca_kmax", "_patch_args", "tap_plan")

    def for_order(self, k):
        """The level as the k-th block of a stage sees it: `Block(order_index = i % len(order))` attends along curve slot k
        (model.py:480-481, model_ca.py:285,354), so only the patch gather tables differ; everything else is shared."""
        if k == 0:
            return self
        v = self._views.get(k)
        if v is None:
            v = Level()
            for name in Level.__slots__:
                if name not in ("_views",) and hasattr(self, name):  # (shares `patch` / `_patch_args` with the base level)
                    setattr(v, name, getattr(self, name))
            while len(self.patch) <= k:  # built on first use: a stage of depth 1 never asks for the other curve slots
                j = len(self.patch)
                order, off, offp, B, K, i32 = self._patch_args
                tabs_k = (torch.empty(self.npad, **i32), torch.empty(self.npad, **i32), torch.empty(self.npad, **i32),
                          torch.empty(max(self.n_extra, 1), **i32))
                call("lotus_fe_patch", order[j], off, offp, B, K, self.npad, *tabs_k)
                self.patch.append(tabs_k)
            v.gidx, v.owner, v.kext, v.ext_pos = self.patch[k]
            v._views = {}
            self._views[k] = v
        return v


def draw_order_perms(n_levels, shuffle=True):
    """The reference draws torch.randperm(4) from the global CPU generator once at input and once per
    pooling (model.py:130-134, :750-754), also in eval mode (SURVEY.md Trap 4).  Drawing the same
    number of permutations in the same order keeps seeded runs comparable."""
    if not shuffle:
        return [list(range(4)) for _ in range(n_levels)]
    return [torch.randperm(4).tolist() for _ in range(n_levels)]


class FrontEnd:
    def __init__(self, n_levels, patch_size=128, grid_size=0.01, orders=ORDERS, n_patch_orders=1, conv_widths=None):
        self.n_levels = n_levels
        # widest 3^3 convolution of every level (None: unknown): levels with few rows and wide layers get a tap plan
        # (lotus_fe_tap_plan) for the tap-grouped convolution path, csrc/conv.hip
        self.conv_widths = list(conv_widths) if conv_widths is not None else None
        self.n_patch_orders = max(1, min(4, int(n_patch_orders)))  # curve slots whose patch tables are built (stage depth)
        self.K = patch_size
        self.grid_size = float(np.float32(grid_size))
        self.order_ids = [ORDERS.index(o) for o in orders]
        assert len(self.order_ids) == 4, "the HIP front-end is built for the 4-curve configuration"
        self.depth_bound = 16  # tightened to the observed depth after the first batch

    @torch.no_grad()
    def build(self, pc_fts, counts, ctx_counts, perms, need_coord=False):
        """pc_fts: f32 [N, >=3] CUDA (xyz = first three columns).  counts / ctx_counts: python
        lists (points / instruction tokens per cloud).  perms: n_levels permutations of range(4)."""
        return self.finish(self.launch(pc_fts, counts, perms), ctx_counts, need_coord)

    @torch.no_grad()
    def launch(self, pc_fts, counts, perms, stream=None, wait_current=True):
        """First half of build(): enqueue the sync-free device pipeline (grid coordinates, codes, sorts,
        pooling of every level) and the asynchronous device->host copy of the per-level counts.  With
        `stream` (a side stream) the work is ordered after everything already enqueued on the current stream
        and runs concurrently with whatever the current stream gets next — the input-dependent integer part of
        the NEXT batch can be prefetched under the backward pass of the current one.  finish() completes it."""
        cur = torch.cuda.current_stream()
        if stream is not None:
            if wait_current:  # (not needed when pc_fts was produced on `stream` itself, e.g. uploaded there)
                stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                pend = self._launch(pc_fts, counts, perms, ws_slot=5)
        else:
            pend = self._launch(pc_fts, counts, perms, ws_slot=1)
        pend["stream"] = stream
        return pend

    def _launch(self, pc_fts, counts, perms, ws_slot):
        dev = pc_fts.device
        N, B, Lv = int(pc_fts.shape[0]), len(counts), self.n_levels
        assert N == sum(counts) and pc_fts.stride(1) == 1
        ld = pc_fts.stride(0)
        i32 = dict(dtype=torch.int32, device=dev)
        # meta: [0:8] state, [8:8+Lv] n per level, then per-level per-cloud counts
        meta = torch.zeros(8 + Lv + Lv * B, **i32)
        state, n_dev = meta[0:8], meta[8:8 + Lv]
        cnts = meta[8 + Lv:].view(Lv, B)
        # host -> device through pinned memory: a pageable copy would make the host wait for the whole queue
        hc = torch.empty(B + 1, dtype=torch.int32, pin_memory=True)
        hc[0] = N
        hc[1:] = torch.as_tensor(counts, dtype=torch.int32)
        dc = hc.to(dev, non_blocking=True)
        n_dev[0:1].copy_(dc[0:1])
        cnts[0].copy_(dc[1:])
        batch0 = torch.repeat_interleave(torch.arange(B, **i32), dc[1:], output_size=N)

        bbits = max(1, (B - 1).bit_length())
        ws_sort = WS.get(query("lotus_fe_sort_workspace", N), dev, slot=ws_slot)
        scratch = torch.empty(8, **i32)
        gmax = scratch[4:5]
        raw = []
        # every table of the sync-free half out of one block (see _Arena): per level 4 N codes + 4 N sort keys (int64), order,
        # inverse, grid, batch, cluster, seg (int32)
        arena = _Arena((Lv + 1) * (N + 64) * (2 * 32 + 16 + 16 + 12 + 4 + 4 + 4) + 4096, dev)
        keep = [meta, batch0, scratch, arena.buf]

        def empty(*shape, dtype=torch.int32):
            t = arena.empty(shape, dtype)
            if t is None:  # (cannot happen with the bound above; a plain tensor is still correct)
                t = torch.empty(*shape, dtype=dtype, device=dev)
                keep.append(t)
            return t

        grid = empty(N, 3)
        call("lotus_fe_grid", pc_fts, ld, N, self.grid_size, grid, gmax, scratch)
        code = empty(4, N, dtype=torch.int64)
        perm0 = (np.asarray(self.order_ids, dtype=np.int32)[np.asarray(perms[0])]).astype(np.int32)
        call("lotus_fe_encode", grid, batch0, N, gmax, perm0.ctypes.data, self.depth_bound, state, code, N)
        batch = batch0
        for s in range(Lv):
            skeys = empty(4, N, dtype=torch.int64)
            order = empty(4, N)
            inverse = empty(4, N)
            key_bits = max(1, 3 * max(self.depth_bound - s, 0) + bbits)
            call("lotus_fe_sort", code, N, n_dev[s:s + 1], N, key_bits, skeys, order, inverse, ws_sort, ws_sort.numel())
            raw.append(dict(grid=grid, batch=batch, code=code, order=order, inverse=inverse, skeys=skeys))
            if s + 1 < Lv:
                cluster = empty(N)
                seg = empty(N + 1)
                ccode = empty(4, N, dtype=torch.int64)
                cgrid = empty(N, 3)
                cbatch = empty(N)
                perm = np.asarray(perms[s + 1], dtype=np.int32)
                call("lotus_fe_pool", code, skeys, order, grid, batch, n_dev[s:s + 1], N, perm.ctypes.data, B, cluster,
                     seg, n_dev[s + 1:s + 2], ccode, cgrid, cbatch, cnts[s + 1], state[2:3] if s == 0 else None)
                raw[-1].update(cluster=cluster, seg=seg)
                grid, batch, code = cgrid, cbatch, ccode
        meta_h = torch.empty(meta.shape, dtype=torch.int32, pin_memory=True)
        meta_h.copy_(meta, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return dict(raw=raw, meta_h=meta_h, event=ev, keep=keep, pc_fts=pc_fts, counts=list(counts), perms=perms,
                    depth_bound=self.depth_bound)

    @torch.no_grad()
    def finish(self, pend, ctx_counts, need_coord=False):
        """Second half of build(): wait for the counts (the one host synchronisation of the front-end; free when
        launch() ran ahead on a side stream), then build the exactly sized neighbour / patch / tile tables on the
        current stream."""
        pc_fts, counts, perms, raw = pend["pc_fts"], pend["counts"], pend["perms"], pend["raw"]
        dev = pc_fts.device
        N, B, Lv = int(pc_fts.shape[0]), len(counts), self.n_levels
        i32 = dict(dtype=torch.int32, device=dev)
        if SYNC_WAIT is not None:  # (bench.py: host seconds spent waiting for the prefetched counts)
            import time
            t0 = time.perf_counter()
            _capi.wait_event(pend["event"])
            SYNC_WAIT[0] += time.perf_counter() - t0
        else:
            _capi.wait_event(pend["event"])  # (never Event.synchronize(): see _capi.wait_event)
        fe = pend["stream"]
        if fe is not None:  # tables were allocated on the side stream and are consumed here
            cur = torch.cuda.current_stream()
            cur.wait_event(pend["event"])
            for t in pend["keep"]:
                t.record_stream(cur)
        meta_h = pend["meta_h"].numpy()
        if meta_h[0] & 1:
            depth = int(meta_h[1])
            if depth > 16:
                raise ValueError(f"serialisation depth {depth} > 16 (model.py:110)")
            self.depth_bound = 16
            return self.build(pc_fts, counts, ctx_counts, perms, need_coord)
        depth0 = int(meta_h[1])
        self.depth_bound = max(depth0, 1)
        if depth0 < Lv - 1:
            raise NotImplementedError("cloud extent below 2^(levels-1) voxels: pooling_depth 0 path not built")
        ns = meta_h[8:8 + Lv].tolist()
        cnt_h = meta_h[8 + Lv:].reshape(Lv, B)

        # ---- exact-size tables per level
        levels = []
        host_tabs, host_slices = [], []
        K = self.K

        def push(arr):
            arr = np.ascontiguousarray(arr, dtype=np.int32).reshape(-1)
            start = sum(a.size for a in host_tabs)
            host_tabs.append(arr)
            return (start, arr.size)

        ctx_off = np.concatenate([[0], np.cumsum(ctx_counts)]).astype(np.int64)
        ctx_cnt = np.asarray(ctx_counts, dtype=np.int64)
        plans = []
        ar_b = np.arange(B)
        for s in range(Lv):
            # vectorised over the clouds (a Python loop per cloud and tile cost 0.7 ms of host time per step)
            c = cnt_h[s].astype(np.int64)
            off = np.concatenate([[0], np.cumsum(c)])
            cpad = np.where(c > K, (c + K - 1) // K * K, c)
            offp = np.concatenate([[0], np.cumsum(cpad)])
            # self-attention patches: one tile per patch of <= K serialised points (a short cloud is one tile)
            nseg = np.where(c == 0, 0, np.where(c <= K, 1, cpad // K))
            nt_self = int(nseg.sum())
            cb = np.repeat(ar_b, nseg)
            within = np.arange(nt_self) - np.repeat(np.cumsum(nseg) - nseg, nseg)
            st = offp[cb] + within * K
            ln = np.where(c[cb] <= K, c[cb], K)
            tiles = np.stack([st, ln, st, ln], 1)
            blocks = np.stack([np.arange(nt_self), np.ones(nt_self, np.int64), np.ones(nt_self, np.int64), np.zeros(nt_self, np.int64), st, ln], 1)
            # cross-attention: 128-row tiles of a cloud's points x that cloud's instruction tokens; backward blocks = G
            # interleaved tile groups per cloud (one key-side partial slot each)
            nt = (c + 127) // 128
            G = int(min(8, max(1, int(nt.max()) if B else 1)))
            n_ca = int(nt.sum())
            cbt = np.repeat(ar_b, nt)
            base = np.cumsum(nt) - nt
            ti = np.arange(n_ca) - np.repeat(base, nt)
            ca_tiles = np.stack([off[cbt] + ti * 128, np.minimum(128, c[cbt] - ti * 128), ctx_off[cbt], ctx_cnt[cbt]], 1)
            gb = np.repeat(ar_b, G)
            gg = np.tile(np.arange(G), B)
            ca_blocks = np.stack([base[gb] + gg, np.maximum(0, (nt[gb] - gg + G - 1) // G), np.full(B * G, G), gg, ctx_off[gb], ctx_cnt[gb]], 1)
            plans.append(dict(off=push(off), offp=push(offp), tiles=push(tiles), blocks=push(blocks),
                              ca_tiles=push(ca_tiles), ca_blocks=push(ca_blocks), n_tiles=nt_self,
                              n_ca_tiles=n_ca, n_ca_blocks=B * G, G=G, npad=int(offp[-1]),
                              off_host=off))
        ntab = sum(a.size for a in host_tabs)
        tabs_h = torch.empty(ntab, dtype=torch.int32, pin_memory=True)  # pinned: the upload must not drain the queue
        np.concatenate(host_tabs, out=tabs_h.numpy())
        # A prefetched front-end also builds its exact-size tables on the front-end stream: the host is ahead of the GPU
        # when forward() gets here, so the ~0.5 ms of hash / neighbour / patch kernels run under the tail of the previous
        # backward pass instead of in front of the stem convolution on the critical stream (the current stream waits for
        # one event at the end; LOTUS_FE_FINISH_SIDE=0 keeps them on the current stream).
        side = fe is not None and FINISH_ON_SIDE
        made = []  # tensors allocated under the front-end stream and consumed on the current one
        with (torch.cuda.stream(fe) if side else contextlib.nullcontext()):
            levels = self._finish_tables(pend, tabs_h, plans, ns, cnt_h, depth0, meta_h, ctx_counts, need_coord, made,
                                         ws_slot=5 if side else 1)
        if side:
            ev = torch.cuda.Event()
            ev.record(fe)
            cur.wait_event(ev)
            for t in made:
                t.record_stream(cur)
        return levels

    def _finish_tables(self, pend, tabs_h, plans, ns, cnt_h, depth0, meta_h, ctx_counts, need_coord, made, ws_slot):
        pc_fts, counts, raw = pend["pc_fts"], pend["counts"], pend["raw"]
        dev = pc_fts.device
        B, Lv, K = len(counts), self.n_levels, self.K
        i32 = dict(dtype=torch.int32, device=dev)
        levels = []

        # the exactly sized tables out of one block (see _Arena): 27 n neighbours, the tap plan, 3 npad + n_extra patch tables
        # and 3 n pooled coordinates per level, 125 n stem neighbours at level 0
        need = 125 * ns[0] * 4 + 4096
        for s_ in range(Lv):
            n_ = ns[s_]
            plan_ints = query("lotus_fe_tap_plan_ints", n_) if (self.conv_widths is not None and n_ > 0) else 0
            need += 4 * (27 * n_ + plan_ints + 3 * plans[s_]["npad"] + max(plans[s_]["npad"] - n_, 1) + 3 * n_) + 8 * 256
        arena = _Arena(need, dev)
        made.append(arena.buf)

        def empty(*shape, **kw):
            t = arena.empty(shape, kw.get("dtype", torch.int32))
            if t is None:
                t = torch.empty(*shape, **(kw or i32))
                made.append(t)
            return t

        tabs = tabs_h.to(dev, non_blocking=True)
        made.append(tabs)

        def view(sl, cols=None):
            t = tabs[sl[0]:sl[0] + sl[1]]
            return t.view(-1, cols) if cols else t

        ws_n = WS.get(query("lotus_fe_neighbours_workspace", ns[0]), dev, slot=ws_slot)
        for s in range(Lv):
            n, r, pl = ns[s], raw[s], plans[s]
            lv = Level()
            lv.n, lv.counts, lv.depth = n, cnt_h[s].tolist(), depth0 - s
            # points sharing their voxel with a lower-indexed point: only the input level can have them (pooled levels
            # are one point per cell by construction); a single-level model has no pooling pass to count them
            lv.n_dup = (int(meta_h[2]) if Lv > 1 else -1) if s == 0 else 0
            lv.off, lv.off_host = view(pl["off"]), pl["off_host"]
            lv.grid, lv.batch = r["grid"][:n], r["batch"][:n]
            lv.code, lv.order, lv.inverse = r["code"][:, :n], r["order"][:, :n], r["inverse"][:, :n]
            lv.nbr27 = empty(27, n)
            call("lotus_fe_neighbours", lv.grid, lv.batch, n, 3, lv.nbr27, ws_n, ws_n.numel())
            lv.tap_plan = None
            if self.conv_widths is not None and s < len(self.conv_widths) and n > 0 and \
                    query("lotus_conv_tap_eligible", n, self.conv_widths[s], self.conv_widths[s]):
                lv.tap_plan = empty(query("lotus_fe_tap_plan_ints", n))
                call("lotus_fe_tap_plan", lv.nbr27, lv.order[0], n, lv.tap_plan)
            lv.nbr125 = None
            if s == 0:
                lv.nbr125 = empty(125, n)
                call("lotus_fe_neighbours", lv.grid, lv.batch, n, 5, lv.nbr125, ws_n, ws_n.numel())
            lv.npad = pl["npad"]
            lv.gidx = empty(lv.npad)
            lv.owner = empty(lv.npad)
            lv.kext = empty(lv.npad)
            lv.n_extra = lv.npad - n
            lv.ext_pos = empty(max(lv.n_extra, 1))
            call("lotus_fe_patch", r["order"], lv.off, view(pl["offp"]), B, K, lv.npad, lv.gidx, lv.owner, lv.kext,
                 lv.ext_pos)
            # deeper stages: block i attends along curve slot i % 4 — those tables are built by Level.for_order on first use
            # (every level used to get max(depths) tables although most stages of [2, 2, 2, 6, 2] need 2)
            lv.patch, lv._views = [(lv.gidx, lv.owner, lv.kext, lv.ext_pos)], {}
            lv._patch_args = (r["order"], lv.off, view(pl["offp"]), B, K, i32)
            lv.self_tiles, lv.self_blocks = view(pl["tiles"], 4), view(pl["blocks"], 6)
            lv.n_self_tiles = pl["n_tiles"]
            lv.ca_tiles, lv.ca_blocks = view(pl["ca_tiles"], 4), view(pl["ca_blocks"], 6)
            lv.n_ca_tiles, lv.n_ca_blocks, lv.ca_groups = pl["n_ca_tiles"], pl["n_ca_blocks"], pl["G"]
            lv.ca_kmax = int(max(ctx_counts)) if len(ctx_counts) else 0  # longest instruction: selects the short-key kernels
            lv.cluster = lv.seg_start = lv.members = lv.coord = lv.parent = None
            if s > 0:
                pr = raw[s - 1]
                lv.cluster = pr["cluster"][:ns[s - 1]]   # child id of every parent point
                lv.seg_start = pr["seg"][:n + 1]          # CSR of the children into members
                lv.members = pr["order"][0, :ns[s - 1]]   # parent rows sorted by parent code[0]
                lv.parent = levels[s - 1]
            levels.append(lv)
        if need_coord:
            levels[0].coord = pc_fts[:, :3]
            for s in range(1, Lv):
                lv = levels[s]
                lv.coord = empty(lv.n, 3, dtype=torch.float32, device=dev)
                call("lotus_fe_pool_coord", levels[s - 1].coord.contiguous(), lv.members, lv.seg_start, lv.n, lv.coord)
        return levels

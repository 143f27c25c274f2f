"""ORACLE (test infrastructure, never shipped as product): integer front-end of 3D-LOTUS.

CPU/numpy restatement of the reference's voxelisation + serialisation + index tables.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.
Pinned against the *unshimmed* reference functions by tests/test_oracle_vs_reference.py
(container only) and by the committed fixtures tests/golden/*.npz.

Each function cites the reference lines it follows (paths relative to /root/reference).
All integer outputs are bit-exact targets for the HIP front-end.
"""
import numpy as np

GRID_SIZE = np.float32(0.01)


# ----------------------------------------------------------------------------- voxel ids
def grid_coord(coord, grid_size=GRID_SIZE):
    """genrobo3d/models/PointTransformerV3/model.py:96-98
    grid = int32(trunc((coord - coord.min(0)) / grid_size)) with the *batch-global* min and a true
    IEEE fp32 division."""
    coord = np.ascontiguousarray(coord, dtype=np.float32)
    mn = coord.min(axis=0)
    d = (coord - mn).astype(np.float32)
    q = np.divide(d, np.float32(grid_size), dtype=np.float32)
    return np.trunc(q).astype(np.int32)


def serialized_depth(grid):
    """model.py:102  depth = int(grid_coord.max()).bit_length()"""
    return int(grid.max()).bit_length()


# ----------------------------------------------------------------------------- curves
def z_order_code(grid, depth):
    """serialization/z_order.py:40-49,66-101: bit i of x -> bit 3i+2, y -> 3i+1, z -> 3i.
    (The reference masks coordinates to `depth` bits through its 8-bit LUT passes.)"""
    g = grid.astype(np.int64)
    x, y, z = g[:, 0], g[:, 1], g[:, 2]
    key = np.zeros_like(x)
    for i in range(depth):
        m = np.int64(1) << i
        key |= ((x & m) << (2 * i + 2)) | ((y & m) << (2 * i + 1)) | ((z & m) << (2 * i))
    return key


def hilbert_code(grid, depth):
    """serialization/hilbert.py:91-198 (Skilling 2004, 'transpose' undo loop on bit planes,
    interleave dims per bit MSB-first, Gray->binary prefix xor, pack to int64).
    Integer restatement of the reference's bit-plane tensor program."""
    X = [grid[:, d].astype(np.int64).copy() for d in range(3)]
    for bit in range(depth):  # bit 0 = MSB of the depth-bit field   (hilbert.py:154-172)
        pos = depth - 1 - bit
        Q = np.int64(1) << pos
        P = Q - 1
        for dim in range(3):
            on = (X[dim] & Q) != 0
            # on: invert lower bits of dim 0; off: exchange lower bits of dim and dim 0
            t = np.where(on, 0, (X[0] ^ X[dim]) & P)
            X[0] = np.where(on, X[0] ^ P, X[0] ^ t)
            if dim != 0:
                X[dim] = X[dim] ^ t
    # interleave (hilbert.py:175): for each bit MSB-first: dim0, dim1, dim2
    g = np.zeros_like(X[0])
    for pos in range(depth):
        for dim in range(3):
            g |= ((X[dim] >> pos) & 1) << (3 * pos + (2 - dim))
    # Gray -> binary (hilbert.py:178, :70-88): prefix xor from the MSB
    s = 1
    while s < 3 * depth:
        g ^= g >> s
        s *= 2
    return g


def encode(grid, batch, depth, order):
    """serialization/default.py:9-24"""
    if order == "z":
        code = z_order_code(grid, depth)
    elif order == "z-trans":
        code = z_order_code(grid[:, [1, 0, 2]], depth)
    elif order == "hilbert":
        code = hilbert_code(grid, depth)
    elif order == "hilbert-trans":
        code = hilbert_code(grid[:, [1, 0, 2]], depth)
    else:
        raise NotImplementedError(order)
    return (batch.astype(np.int64) << (depth * 3)) | code


ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")


def stable_argsort(code):
    """Tie rule of the build (SURVEY.md Trap 5): stable by (code, index)."""
    return np.argsort(code, axis=-1, kind="stable")


def inverse_perm(order):
    """model.py:122-128"""
    inv = np.empty_like(order)
    n = order.shape[-1]
    if order.ndim == 1:
        inv[order] = np.arange(n, dtype=order.dtype)
    else:
        for k in range(order.shape[0]):
            inv[k, order[k]] = np.arange(n, dtype=order.dtype)
    return inv


def serialization(grid, batch, depth=None, orders=ORDERS, perm=None):
    """model.py:83-138.  `perm` = the injected shuffle_orders permutation (Trap 4)."""
    if depth is None:
        depth = serialized_depth(grid)
    nb = int(batch.max()) + 1
    assert depth * 3 + nb.bit_length() <= 63 and depth <= 16  # model.py:105,110
    code = np.stack([encode(grid, batch, depth, o) for o in orders])
    order = stable_argsort(code)
    inverse = inverse_perm(order)
    if perm is not None:
        perm = np.asarray(perm)
        code, order, inverse = code[perm], order[perm], inverse[perm]
    return dict(depth=depth, code=code, order=order, inverse=inverse)


# ----------------------------------------------------------------------------- patches
def padding_tables(counts, patch_size):
    """SerializedAttention.get_padding_and_inverse, model.py:410-466.
    counts: points per cloud.  Returns pad i64[N_pad], unpad i64[N], cu_seqlens i32[P+1]."""
    counts = np.asarray(counts, dtype=np.int64)
    K = patch_size
    cpad = np.where(counts > K, (counts + K - 1) // K * K, counts)
    off = np.concatenate([[0], np.cumsum(counts)])
    offp = np.concatenate([[0], np.cumsum(cpad)])
    pad = np.arange(offp[-1], dtype=np.int64)
    unpad = np.arange(off[-1], dtype=np.int64)
    cu = []
    for i in range(len(counts)):
        unpad[off[i]:off[i + 1]] += offp[i] - off[i]
        if counts[i] != cpad[i]:
            r = counts[i] % K
            pad[offp[i + 1] - K + r: offp[i + 1]] = pad[offp[i + 1] - 2 * K + r: offp[i + 1] - K]
        pad[offp[i]:offp[i + 1]] -= offp[i] - off[i]
        cu.append(np.arange(offp[i], offp[i + 1], K, dtype=np.int32))
    cu = np.concatenate(cu + [np.array([offp[-1]], dtype=np.int32)])
    return pad, unpad, cu.astype(np.int32)


# ----------------------------------------------------------------------------- pooling
def pooling_tables(code, grid, batch, depth, stride=2, perm=None):
    """Index part of SerializedPooling.forward, model.py:713-772.
    code: i64[4,N] (already shuffled) of the parent level.  Returns the child level's tables.
    Points of one cluster are kept in ascending parent index (stable) order."""
    pooling_depth = (int(np.ceil(stride)) - 1).bit_length()
    if pooling_depth > depth:
        pooling_depth = 0
    c = code >> (pooling_depth * 3)
    uniq, cluster, counts = np.unique(c[0], return_inverse=True, return_counts=True)
    indices = np.argsort(cluster, kind="stable")
    idx_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    head = indices[idx_ptr[:-1]]
    ccode = c[:, head]
    order = stable_argsort(ccode)
    inverse = inverse_perm(order)
    if perm is not None:
        perm = np.asarray(perm)
        ccode, order, inverse = ccode[perm], order[perm], inverse[perm]
    return dict(cluster=cluster.astype(np.int64), cluster_counts=counts.astype(np.int64), indices=indices,
                idx_ptr=idx_ptr, head=head, grid=(grid[head] >> pooling_depth).astype(np.int32),
                batch=batch[head], depth=depth - pooling_depth, code=ccode, order=order,
                inverse=inverse)


# ----------------------------------------------------------------------------- neighbours
def neighbour_table(grid, batch, ksize):
    """Submanifold-conv neighbour lookup (spconv.SubMConv3d call sites model.py:615-622,
    :844-853; spconv 2.3.6 is un-vendored => parity unpinned, semantics per SURVEY.md §8c):
    nbr[p, t] = index of the active site at grid[p] + delta_t in the same cloud, or -1.
    Taps x-major over (x, y, z): t = ((dx+r)*k + (dy+r))*k + (dz+r).  Duplicate voxels ->
    lowest index."""
    r = ksize // 2
    n = grid.shape[0]
    g = grid.astype(np.int64)
    S = int(g.max()) + ksize + 2

    def key(b, x, y, z):
        return ((b * S + (x + r)) * S + (y + r)) * S + (z + r)

    keys = key(batch.astype(np.int64), g[:, 0], g[:, 1], g[:, 2])
    sidx = np.argsort(keys, kind="stable")
    skeys = keys[sidx]
    first = np.ones(n, dtype=bool)
    first[1:] = skeys[1:] != skeys[:-1]
    ukeys, uidx = skeys[first], sidx[first]
    out = np.full((n, ksize ** 3), -1, dtype=np.int32)
    t = 0
    for dx in range(-r, r + 1):
        for dy in range(-r, r + 1):
            for dz in range(-r, r + 1):
                q = key(batch.astype(np.int64), g[:, 0] + dx, g[:, 1] + dy, g[:, 2] + dz)
                pos = np.minimum(np.searchsorted(ukeys, q), len(ukeys) - 1)
                hit = ukeys[pos] == q
                out[hit, t] = uidx[pos[hit]]
                t += 1
    return out


def offset2batch(counts):
    return np.repeat(np.arange(len(counts), dtype=np.int64), counts)


def build_all_levels(coord, counts, n_levels, patch_size=128, perms=None, grid_size=GRID_SIZE):
    """All integer tables of one forward (model_ca.py:383-397 + every SerializedPooling):
    level 0 from coordinates, levels 1.. by grid pooling.  perms: list of n_levels permutations."""
    coord = np.asarray(coord, dtype=np.float32)
    batch = offset2batch(counts)
    grid = grid_coord(coord, grid_size)
    ser = serialization(grid, batch, perm=None if perms is None else perms[0])
    levels = []
    lvl = dict(grid=grid, batch=batch, counts=np.asarray(counts, dtype=np.int64), **ser)
    for s in range(n_levels):
        pad, unpad, cu = padding_tables(lvl["counts"], patch_size)
        lvl["pad"], lvl["unpad"], lvl["cu_seqlens"] = pad, unpad, cu
        lvl["nbr27"] = neighbour_table(lvl["grid"], lvl["batch"], 3)
        if s == 0:
            lvl["nbr125"] = neighbour_table(lvl["grid"], lvl["batch"], 5)
        levels.append(lvl)
        if s + 1 < n_levels:
            pt = pooling_tables(lvl["code"], lvl["grid"], lvl["batch"], lvl["depth"],
                                perm=None if perms is None else perms[s + 1])
            nb = len(counts)
            pt["counts"] = np.bincount(pt["batch"], minlength=nb).astype(np.int64)
            lvl = pt
    return levels

"""Diagnostic: stand-alone timing of the attention kernels (patch self attention and point <-> instruction cross
attention) per level of the v1 hierarchy at the bench size, tile kernels vs query-per-lane kernels (LOTUS_XQ = 0 / 2, one
child process each).   python tools/attn_ab.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    import robot_3dlotus_amd  # noqa: F401
    from robot_3dlotus_amd import ops, synth
    from robot_3dlotus_amd.frontend import FrontEnd

    batch = synth.synth_batch(16, 4096, seed=0)
    perms = [[0, 1, 2, 3], [1, 0, 3, 2], [2, 3, 0, 1], [3, 2, 1, 0], [0, 2, 1, 3]]
    levels = FrontEnd(5).build(batch["pc_fts"].cuda(), batch["npoints_in_batch"], batch["txt_lens"], perms)
    Lctx = sum(batch["txt_lens"])
    out = {}

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        return round(best, 1)

    for lv, C, H in ((0, 64, 2), (0, 128, 4), (1, 128, 4), (2, 256, 8), (3, 512, 16), (4, 768, 32)):
        L = levels[lv]
        d = C // H
        g = torch.Generator(device="cuda").manual_seed(lv)
        qkv = torch.randn(L.n, 3 * C, device="cuda", generator=g)
        dout = torch.randn(L.n, C, device="cuda", generator=g)
        qn = (torch.ones(d, device="cuda"), torch.zeros(d, device="cuda"))
        att = torch.empty(L.n, C, device="cuda")
        lse = torch.empty(L.npad, H, device="cuda")
        dqkv = torch.empty(L.n, 3 * C, device="cuda")
        extra = torch.empty(max(L.n_extra, 1), 2 * C, device="cuda")
        f = lambda: ops.attention_fwd(qkv, 3 * C, 0, qkv, 3 * C, C, 2 * C, L.gidx, L.gidx, L.owner, L.self_tiles, L.n_self_tiles, qn, qn, att, lse, H, d)
        b = lambda: ops.attention_bwd(qkv, 3 * C, 0, qkv, 3 * C, C, 2 * C, L.gidx, L.gidx, L.owner, L.self_tiles, L.self_blocks, L.n_self_tiles,
                                      qn, qn, att, dout, lse, dqkv, 3 * C, 0, dqkv, 3 * C, C, 2 * C, 0, 0, H, d, 0.0, 0, L.kext, L.ext_pos,
                                      L.n_extra, extra)
        out[f"self L{lv} C{C} fwd"] = timeit(f)
        out[f"self L{lv} C{C} bwd"] = timeit(b)
        q = torch.randn(L.n, C, device="cuda", generator=g)
        kv = torch.randn(Lctx, 2 * C, device="cuda", generator=g)
        lse2 = torch.empty(L.n, H, device="cuda")
        dq = torch.empty(L.n, C, device="cuda")
        G = L.ca_groups
        dkvp = torch.empty(G, Lctx, 2 * C, device="cuda")
        f2 = lambda: ops.attention_fwd(q, C, 0, kv, 2 * C, 0, C, None, None, None, L.ca_tiles, L.n_ca_tiles, qn, qn, att, lse2, H, d, k_max=L.ca_kmax)
        b2 = lambda: ops.attention_bwd(q, C, 0, kv, 2 * C, 0, C, None, None, None, L.ca_tiles, L.ca_blocks, L.n_ca_blocks, qn, qn, att, dout, lse2,
                                       dq, C, 0, dkvp, 2 * C, 0, C, Lctx * 2 * C, 0, H, d, k_max=L.ca_kmax)
        out[f"cross L{lv} C{C} fwd"] = timeit(f2)
        out[f"cross L{lv} C{C} bwd"] = timeit(b2)
    print("RESULT " + json.dumps(out))


if os.environ.get("ATTN_AB_CHILD"):
    child()
    sys.exit(0)
vals = sys.argv[1:] or ["0", "2"]
res = {}
for v in vals:
    r = subprocess.run([sys.executable, __file__], env=dict(os.environ, ATTN_AB_CHILD="1", LOTUS_XQ=v), capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print(r.stderr[-1500:])
        sys.exit(1)
    res[v] = json.loads(line[0][7:])
print("kernel (us, stand-alone) " + " ".join(f"LOTUS_XQ={v}" for v in vals))
for k in res[vals[0]]:
    print(f"{k:24s} " + " ".join(f"{res[v][k]:8.1f}" for v in vals))
print("sum", {v: round(sum(res[v].values()), 1) for v in vals})

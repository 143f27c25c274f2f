"""Stand-alone time of the self-attention backward per level of the headline batch (CLOUDS x NPOINTS, default 16 x 4096), attn_drop 0.1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import ops, synth  # noqa: E402
from robot_3dlotus_amd.frontend import FrontEnd  # noqa: E402

batch = synth.synth_batch(int(os.environ.get("CLOUDS", "16")), int(os.environ.get("NPOINTS", "4096")), seed=0)
perms = [[0, 1, 2, 3], [1, 0, 3, 2], [2, 3, 0, 1], [3, 2, 1, 0], [0, 2, 1, 3]]
levels = FrontEnd(5).build(batch["pc_fts"].cuda(), batch["npoints_in_batch"], batch["txt_lens"], perms)
tot = 0.0
for lv, C, H in ((0, 64, 2), (1, 128, 4), (2, 256, 8), (3, 512, 16), (4, 768, 32), (3, 512, 16), (2, 256, 8), (1, 128, 4), (0, 128, 4)):
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
    ops.attention_fwd(qkv, 3 * C, 0, qkv, 3 * C, C, 2 * C, L.gidx, L.gidx, L.owner, L.self_tiles, L.n_self_tiles, qn, qn, att, lse, H, d, 0.1, 7)
    b = lambda: ops.attention_bwd(qkv, 3 * C, 0, qkv, 3 * C, C, 2 * C, L.gidx, L.gidx, L.owner, L.self_tiles, L.self_blocks, L.n_self_tiles,
                                  qn, qn, att, dout, lse, dqkv, 3 * C, 0, dqkv, 3 * C, C, 2 * C, 0, 0, H, d, 0.1, 7, L.kext, L.ext_pos,
                                  L.n_extra, extra)
    for _ in range(3):
        b()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    tot += best
    print(f"self-attention bwd L{lv} n={L.n} C={C} H={H} d={d}: {best:.1f} us (incl. the ln reduce launch)", flush=True)
print(f"sum {tot:.1f} us")

"""Diagnostic: time the tile-attention kernels on the level-0 patches of the canonical batch."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import robot_3dlotus_amd
from robot_3dlotus_amd import ops, synth, _capi
from robot_3dlotus_amd.frontend import FrontEnd
drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
batch = synth.synth_batch(16, 4096, seed=0)
lv = FrontEnd(5).build(batch["pc_fts"].cuda(), batch["npoints_in_batch"], batch["txt_lens"], [[0, 1, 2, 3]] * 5)
for (s, C, H) in [(0, 64, 2), (0, 128, 4), (1, 128, 4), (2, 256, 8), (3, 512, 16), (4, 768, 32)]:
    L = lv[s]; n = L.n; d = C // H
    qkv = torch.randn(n, 3 * C, device="cuda"); qn = (torch.ones(d, device="cuda"), torch.zeros(d, device="cuda"))
    att = torch.empty(n, C, device="cuda"); lse = torch.empty(L.npad, H, device="cuda"); dout = torch.randn(n, C, device="cuda")
    dqkv = torch.empty(n, 3 * C, device="cuda"); extra = torch.empty(max(L.n_extra, 1), 2 * C, device="cuda")
    f = lambda: ops.attention_fwd(qkv, 3 * C, 0, qkv, 3 * C, C, 2 * C, L.gidx, L.gidx, L.owner, L.self_tiles, L.n_self_tiles, qn, qn, att, lse, H, d, drop, 7)
    b = lambda: ops.attention_bwd(qkv, 3 * C, 0, qkv, 3 * C, C, 2 * C, L.gidx, L.gidx, L.owner, L.self_tiles, L.self_blocks, L.n_self_tiles, qn, qn, att, dout, lse, dqkv, 3 * C, 0, dqkv, 3 * C, C, 2 * C, 0, 0, H, d, drop, 7, L.kext, L.ext_pos, L.n_extra, extra)
    for name, fn in (("fwd", f), ("bwd", b)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); e1.synchronize()
        print(f"L{s} n={n} C={C} H={H} tiles={L.n_self_tiles} {name}: {e0.elapsed_time(e1)/3*1e3:.1f} us", flush=True)
    if os.environ.get("LOTUS_ATTN_CLK"):
        buf = np.zeros(64, dtype=np.int64)
        _capi.lib().cdll.lotus_debug_attn_clock(ctypes.c_void_p(buf.ctypes.data))
        nst = int(buf[63]); print("   bwd phases us:", [round(float(x) / 100.0, 1) for x in np.diff(buf[:nst])])

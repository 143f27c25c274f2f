"""Diagnostic: coarse phase times (100 MHz wall clock) of block (0,0,0) of the pair-compacted convolution
(LOTUS_CONV_CLK=1): start | neighbour loads | compaction | hashing + work list | image visible | pipeline | epilogue."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import robot_3dlotus_amd
from robot_3dlotus_amd import ops, synth, _capi
from robot_3dlotus_amd.frontend import FrontEnd
batch = synth.synth_batch(16, 4096, seed=0)
lv = FrontEnd(5).build(batch["pc_fts"].cuda(), batch["npoints_in_batch"], batch["txt_lens"], [[0, 1, 2, 3]] * 5)
for s, C in [(0, 64), (0, 128), (1, 128), (2, 256), (3, 512)]:
    L = lv[s]
    x = torch.randn(L.n, C, device="cuda"); w = torch.randn(C, 3, 3, 3, C, device="cuda") * 0.02; b = torch.zeros(C, device="cuda")
    wt = ops.conv_weight_t(w)
    for _ in range(3): ops.conv_fwd(x, w, b, L.nbr27, L.order[0], w_t=wt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv_fwd(x, w, b, L.nbr27, L.order[0], w_t=wt)
    e1.record(); e1.synchronize()
    buf = np.zeros(64, dtype=np.int64)
    _capi.lib().cdll.lotus_debug_conv_clock(ctypes.c_void_p(buf.ctypes.data))
    n = int(buf[63])
    print(f"L{s} n={L.n} C={C}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us/launch; block 0 phases us:", [round(float(v) / 100.0, 1) for v in np.diff(buf[:n])])

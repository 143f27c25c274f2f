"""Diagnostic: cycle stamps of the first group steps of conv_pairs block 0 / wave 0 (LOTUS_CONV_CLK=1 LOTUS_CONV_DBG=64)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import robot_3dlotus_amd
from robot_3dlotus_amd import ops, synth, _capi
from robot_3dlotus_amd.frontend import FrontEnd
batch = synth.synth_batch(16, 4096, seed=0)
lv = FrontEnd(5).build(batch["pc_fts"].cuda(), batch["npoints_in_batch"], batch["txt_lens"], [[0, 1, 2, 3]] * 5)
s, C = (0, 128) if len(sys.argv) < 2 else (int(sys.argv[1]), int(sys.argv[2]))
L = lv[s]
x = torch.randn(L.n, C, device="cuda"); w = torch.randn(C, 3, 3, 3, C, device="cuda") * 0.02; b = torch.zeros(C, device="cuda")
wt = ops.conv_weight_t(w)
for _ in range(3): ops.conv_fwd(x, w, b, L.nbr27, L.order[0], w_t=wt)
torch.cuda.synchronize()
buf = np.zeros(64, dtype=np.int64)
_capi.lib().cdll.lotus_debug_conv_clock(ctypes.c_void_p(buf.ctypes.data))
t = buf[12:62].reshape(10, 5)
print("per-step stage cycles (Q1 | rowids+Q2 | reads+Q3 | writes+Q4) and step-to-step:")
for a in range(10):
    d = np.diff(t[a]); nxt = (t[a + 1, 0] - t[a, 0]) if a < 9 else 0
    print(a, d.tolist(), "step", int(nxt))

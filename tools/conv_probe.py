"""Diagnostic: time lotus_subm_conv fwd/dgrad/wgrad on the levels of the canonical 16 x 4096 batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import robot_3dlotus_amd
from robot_3dlotus_amd import ops, synth
from robot_3dlotus_amd.frontend import FrontEnd

only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
batch = synth.synth_batch(16, 4096, seed=0)
lv = FrontEnd(5).build(batch["pc_fts"].cuda(), batch["npoints_in_batch"], batch["txt_lens"], [[0, 1, 2, 3]] * 5)
chans = [(0, 64), (0, 128), (1, 128), (2, 256), (3, 512), (4, 768)]
for i, (s, C) in enumerate(chans):
    if only >= 0 and i != only:
        continue
    L = lv[s]
    x = torch.randn(L.n, C, device="cuda"); w = torch.randn(C, 3, 3, 3, C, device="cuda") * 0.02; b = torch.zeros(C, device="cuda")
    wt = ops.conv_weight_t(w)
    pairs = int((L.nbr27 >= 0).sum())
    fl = 2.0 * pairs * C * C
    for kind, fn in (("fwd", lambda: ops.conv_fwd(x, w, b, L.nbr27, L.order[0], w_t=wt)),
                     ("dgrad", lambda: ops.conv_dgrad(x, w, L.nbr27, L.order[0], w_t=wt)),
                     ("wgrad", lambda: ops.conv_wgrad(x, x, w.shape, L.nbr27))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if os.environ.get("LOTUS_CONV_CLK") and kind != "wgrad":
            import ctypes, numpy as np
            from robot_3dlotus_amd import _capi
            buf = np.zeros(64, dtype=np.int64)
            _capi.lib().cdll.lotus_debug_conv_clock(ctypes.c_void_p(buf.ctypes.data))
            nst = int(buf[63]); d = np.diff(buf[:nst])
            print("   phase ticks (100 MHz wall clock -> us):", [round(float(x) / 100.0, 1) for x in d])
        print(f"L{s} n={L.n:6d} C={C:4d} {kind:6s} {ms*1e3:8.1f} us  pairs/pt={pairs/L.n:5.2f}  useful {fl/ms*1e-9:6.1f} TF", flush=True)

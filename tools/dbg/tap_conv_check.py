import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import torch, numpy as np
import robot_3dlotus_amd
from robot_3dlotus_amd import ops, synth
from robot_3dlotus_amd.frontend import FrontEnd
from robot_3dlotus_amd._capi import query
dev = torch.device("cuda", 0)
b = synth.synth_batch(16, 4096, seed=0)
levels = FrontEnd(5, conv_widths=[128, 128, 256, 512, 768]).build(b["pc_fts"].to(dev), b["npoints_in_batch"], b["txt_lens"], [[0, 1, 2, 3]] * 5)
import os
for li, C in [(int(a), int(b)) for a, b in (p.split(":") for p in os.environ.get("TAP_CASES", "2:256,3:512,4:768").split(","))]:
    L = levels[li]
    assert L.tap_plan is not None
    n64 = (L.n + 63) // 64 * 64
    cnt = L.tap_plan[:27].cpu().numpy()
    ref = (L.nbr27 >= 0).sum(1).cpu().numpy()
    assert (cnt == ref).all(), (cnt, ref)
    torch.manual_seed(li)
    x = torch.randn(L.n, C, device=dev); w = torch.randn(C, 3, 3, 3, C, device=dev) / (C * 9) ** 0.5
    bias = torch.randn(C, device=dev); add = torch.randn(L.n, C, device=dev)
    wt = ops.conv_weight_t(w)
    y0 = ops.conv_fwd(x, w, bias, L.nbr27, L.order[0], add=add, w_t=wt)
    y1 = ops.conv_fwd(x, w, bias, L.nbr27, L.order[0], add=add, w_t=wt, tap_plan=L.tap_plan)
    d0 = ops.conv_dgrad(x, w, L.nbr27, L.order[0], add=add, w_t=wt, lvl=L)
    d1 = ops.conv_dgrad(x, w, L.nbr27, L.order[0], add=add, w_t=wt, lvl=L, tap_plan=L.tap_plan)
    # fp64 reference of the forward
    nb = L.nbr27.long(); w64 = w.double().reshape(C, 27, C)
    yr = bias.double()[None, :] + add.double()
    for t in range(27):
        m = nb[t] >= 0
        yr[m] += x.double()[nb[t][m]] @ w64[:, t, :].T
    print(li, C, "fwd tap-vs-pairs %.2e  tap-vs-f64 %.2e pairs-vs-f64 %.2e   dgrad tap-vs-pairs %.2e" % (
        float((y1 - y0).abs().max() / y0.abs().max()), float((y1.double() - yr).abs().max() / yr.abs().max()),
        float((y0.double() - yr).abs().max() / yr.abs().max()), float((d1 - d0).abs().max() / d0.abs().max())))
    for name, fn in (("pairs fwd", lambda: ops.conv_fwd(x, w, bias, L.nbr27, L.order[0], w_t=wt)), ("tap fwd", lambda: ops.conv_fwd(x, w, bias, L.nbr27, L.order[0], w_t=wt, tap_plan=L.tap_plan)),
                     ("pairs dgrad", lambda: ops.conv_dgrad(x, w, L.nbr27, L.order[0], w_t=wt, lvl=L)), ("tap dgrad", lambda: ops.conv_dgrad(x, w, L.nbr27, L.order[0], w_t=wt, lvl=L, tap_plan=L.tap_plan))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); e1.synchronize()
        print("   %-12s %7.1f us" % (name, e0.elapsed_time(e1) * 100))

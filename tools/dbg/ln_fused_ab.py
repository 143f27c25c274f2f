"""Stand-alone A/B of lotus_linear_dgrad_ln (LayerNorm backward as the epilogue of the input gradient) against the two launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import robot_3dlotus_amd
from robot_3dlotus_amd import ops

def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    return best

for M, N, C in [(65536, 512, 128), (65536, 384, 128), (65536, 128, 128), (65536, 256, 64), (65536, 192, 64), (65536, 64, 64), (23894, 512, 128), (23894, 384, 128), (23894, 128, 128)]:
    x = torch.randn(M, C, device="cuda"); w = torch.randn(N, C, device="cuda") * 0.05; dy = torch.randn(M, N, device="cuda")
    g = torch.rand(C, device="cuda") + 0.5; b = torch.zeros(C, device="cuda"); add = torch.randn(M, C, device="cuda")
    _, mean, rstd = ops.ln_fwd(x, g, b)
    fused = t(lambda: ops.linear_dgrad_ln(dy, w, x, mean, rstd, g, add=add, drop=(0.1, 5)))
    class H: pass
    def two():
        dn = ops.linear_dgrad(dy, w)
        hand = ops.Handoff(); hand.arm(0.1, 5)
        ops.ln_bwd(dn, x, mean, rstd, g, add=add, hand=hand)
    sep = t(two)
    dg = t(lambda: ops.linear_dgrad(dy, w))
    print(f"M {M} N {N} C {C}: fused {fused:.1f} us   dgrad + ln_bwd {sep:.1f} us   (dgrad alone {dg:.1f})")

"""Diagnostic: time chosen (kind, M, N, K) GEMM shapes; env LOTUS_GEMM_TILE / LOTUS_GEMM_NZ force tile / split."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import robot_3dlotus_amd
from robot_3dlotus_amd import ops

shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (M, N, K) in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; dy = torch.randn(M, N, device="cuda")
    out = []
    for kind, fn in (("fwd", lambda: ops.linear_fwd(x, w, None)), ("dgrad", lambda: ops.linear_dgrad(dy, w)), ("wgrad", lambda: ops.linear_wgrad(dy, x))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out.append(f"{kind} {ms*1e3:7.1f}us {2e-9*M*N*K/ms:6.1f}TF")
    print(f"M={M:6d} N={N:5d} K={K:5d} | " + " | ".join(out), flush=True)

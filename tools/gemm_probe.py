"""Diagnostic: achieved TFLOP/s of lotus_linear_{fwd,dgrad,wgrad} over a list of shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import robot_3dlotus_amd
from robot_3dlotus_amd import ops

shapes = [(8192, 8192, 8192), (4096, 4096, 4096), (65536, 128, 128), (65536, 512, 128), (65536, 128, 512), (65536, 256, 64),
          (65536, 64, 256), (23894, 512, 128), (6077, 1024, 256), (6077, 256, 1024), (1450, 2048, 512), (1450, 512, 2048),
          (361, 3072, 768), (361, 768, 3072)]
for (M, N, K) in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; dy = torch.randn(M, N, device="cuda")
    for kind, fn in (("fwd", lambda: ops.linear_fwd(x, w, None)), ("dgrad", lambda: ops.linear_dgrad(dy, w)), ("wgrad", lambda: ops.linear_wgrad(dy, x))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 if M * N * K > 1e11 else 10
        e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{kind:6s} M={M:6d} N={N:5d} K={K:5d}  {ms*1e3:9.1f} us  {2e-9*M*N*K/ms:7.1f} TF", flush=True)

"""Diagnostic: run ONE dense-layer product many times (for rocprofv3 --pmc passes).
   python tools/dbg/one_gemm.py wgrad 65536 128 128 [bf16|fp32] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import robot_3dlotus_amd  # noqa: F401
from robot_3dlotus_amd import ops
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
mode = sys.argv[5] if len(sys.argv) > 5 else "bf16"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 30
dev = torch.device("cuda", 0)
ops.enable_side_stream(False)
if mode == "bf16":
    ops.set_gemm_precision("bf16")
ctx = ops.storage(torch.bfloat16) if mode == "bf16" else ops.storage(None)
with ctx:
    cast = (lambda t: t.bfloat16()) if mode == "bf16" else (lambda t: t)
    x = cast(torch.randn(M, K, device=dev)); w = torch.randn(N, K, device=dev) * 0.02; dy = cast(torch.randn(M, N, device=dev))
    fn = {"fwd": lambda: ops.linear_fwd(x, w, None), "dgrad": lambda: ops.linear_dgrad(dy, w), "wgrad": lambda: ops.linear_wgrad(dy, x)}[kind]
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); e1.synchronize()
    print(kind, M, N, K, mode, "%.1f us" % (e0.elapsed_time(e1) * 50))

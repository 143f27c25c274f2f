"""Diagnostic: dense layers of the v1 step with bf16 activation storage (lotus_b16_* twins, bf16 MFMA): us per launch and
the fraction of the byte roof (A + C in bf16, weights fp32, at 6.3 TB/s) / the bf16 MFMA roof.   python tools/gemm_b16_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [(65536, 512, 128), (65536, 128, 512), (65536, 128, 128), (65536, 384, 128), (65536, 256, 64), (65536, 64, 256),
          (23894, 512, 128), (23894, 128, 512), (23894, 128, 128), (6077, 1024, 256), (6077, 256, 1024), (6077, 256, 256),
          (1450, 2048, 512), (1450, 512, 2048), (1450, 512, 512), (361, 3072, 768), (361, 768, 768)]
ops.set_gemm_precision("bf16")
with ops.storage(torch.bfloat16):
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = torch.randn(N, K, device=dev) * 0.02
        dy = torch.randn(M, N, device=dev).bfloat16()
        row = [f"M={M:6d} N={N:5d} K={K:5d}"]
        for kind, fn in (("fwd", lambda: ops.linear_fwd(x, w, None)), ("dgrad", lambda: ops.linear_dgrad(dy, w)),
                         ("wgrad", lambda: ops.linear_wgrad(dy, x))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); e1.synchronize()
            us = e0.elapsed_time(e1) * 100
            byt = 2.0 * (M * K + M * N) + 4.0 * N * K
            row.append(f"{kind} {us:6.1f} us ({byt / 6.3e6 / us:4.2f} of bytes, {2.0 * M * N * K / 2.5e9 / us:4.2f} of MFMA)")
        print("  ".join(row), flush=True)

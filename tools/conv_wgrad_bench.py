"""Diagnostic: the 3^3 sparse-convolution weight gradient per level of a 16 x 4096 batch, fp32 / bf16 operands / bf16 storage.
python tools/conv_wgrad_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import ops, synth  # noqa: E402
from robot_3dlotus_amd.frontend import FrontEnd  # noqa: E402

dev = torch.device("cuda", 0)
b = synth.synth_batch(int(os.environ.get("CLOUDS", "16")), 4096, seed=0)
levels = FrontEnd(5).build(b["pc_fts"].to(dev), b["npoints_in_batch"], b["txt_lens"], [[0, 1, 2, 3]] * 5)
for li, (L, C) in enumerate(zip(levels, (64, 128, 256, 512, 768))):
    x, dy = torch.randn(L.n, C, device=dev), torch.randn(L.n, C, device=dev)
    row = [f"level {li} n={L.n:6d} C={C:4d}"]
    for mode in ("fp32", "bf16", "bf16 storage"):
        ops.set_gemm_precision("bf16" if mode != "fp32" else "fp32")
        with ops.storage(torch.bfloat16 if mode == "bf16 storage" else None):
            xx, dd = (x.bfloat16(), dy.bfloat16()) if mode == "bf16 storage" else (x, dy)
            for _ in range(3):
                ops.conv_wgrad(dd, xx, (C, 3, 3, 3, C), L.nbr27)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv_wgrad(dd, xx, (C, 3, 3, 3, C), L.nbr27)
            e1.record(); e1.synchronize()
        row.append(f"{mode} {e0.elapsed_time(e1) * 100:7.1f} us")
    ops.set_gemm_precision("fp32")
    print("  ".join(row), flush=True)

"""Diagnostic: the 3^3 submanifold convolution (pair-compacted kernel) at every level of one synthetic v1 batch (16 clouds x
4096 points), forward mode, encoder and decoder widths, fp32 / bf16 operands.   python tools/conv_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import ops, synth  # noqa: E402
from robot_3dlotus_amd.frontend import FrontEnd  # noqa: E402

dev = torch.device("cuda", 0)
b = synth.synth_batch(int(os.environ.get("CLOUDS", "16")), int(os.environ.get("NPOINTS", "4096")), seed=0)
levels = FrontEnd(5).build(b["pc_fts"].to(dev), b["npoints_in_batch"], b["txt_lens"], [[0, 1, 2, 3]] * 5)
ENC, DEC = (64, 128, 256, 512, 768), (64, 64, 128, 256)   # v1 widths
NATURAL = os.environ.get("ROWIDX") == "0"  # row tiles in index order instead of curve order (diagnostic)
for li, L in enumerate(levels):
    for C in sorted({ENC[li]} | ({DEC[li]} if li < 4 else set())):
        x = torch.randn(L.n, C, device=dev)
        w = torch.randn(C, 3, 3, 3, C, device=dev) / (C * 9) ** 0.5
        row = [f"level {li} n={L.n:6d} C={C:4d}"]
        for mode in (("fp32",) if os.environ.get("FP32_ONLY") else ("fp32", "bf16", "bf16 storage")):
            ops.set_gemm_precision("bf16" if mode != "fp32" else "fp32")
            with ops.storage(torch.bfloat16 if mode == "bf16 storage" else None):
                xx = x.bfloat16() if mode == "bf16 storage" else x
                wt = ops.conv_weight_t(w)
                for _ in range(3):
                    ops.conv_fwd(xx, w, None, L.nbr27, None if NATURAL else L.order[0], w_t=wt)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.conv_fwd(xx, w, None, L.nbr27, None if NATURAL else L.order[0], w_t=wt)
                e1.record(); e1.synchronize()
            row.append(f"{mode} {e0.elapsed_time(e1) * 100:7.1f} us")
        ops.set_gemm_precision("fp32")
        pairs = int((L.nbr27 >= 0).sum())
        # fill of the 32-pair MFMA groups of the pair-compacted kernel: per (row tile of BM points in curve order, tap)
        fills = []
        for BM, G in ((64, 32), (128, 32), (64, 16)):
            nb = L.nbr27 if NATURAL else L.nbr27[:, L.order[0].long()]
            pad = (-L.n) % BM
            act = torch.nn.functional.pad(nb >= 0, (0, pad)).reshape(27, -1, BM).sum(-1)
            fills.append(f"BM{BM}/G{G} {float(act.sum()) / float(((act + G - 1) // G * G).sum()):.2f}")
        row.append("group fill " + " ".join(fills))
        # output-stationary formulation: useful fraction of the products when a wave of G rows skips the taps none of its rows has
        osf = []
        for G in (32, 16):
            nb = L.nbr27 if NATURAL else L.nbr27[:, L.order[0].long()]
            act = torch.nn.functional.pad(nb >= 0, (0, (-L.n) % G)).reshape(27, -1, G)
            osf.append(f"G{G} {float(act.sum()) / float(act.any(-1).sum() * G):.2f} ({float(act.any(-1).sum()) / act.shape[1]:.1f} taps)")
        row.append("OS useful " + " ".join(osf))
        row.append(f"pairs/row {pairs / L.n:.1f}  useful GFLOP {2 * pairs * C * C / 1e9:.2f}")
        print("  ".join(row), flush=True)

"""Stand-alone timing of the HBM-bound kernels of the step at the level sizes of a 16 x 4096 batch, against the bytes each
one has to move (one launch per timing, idle GPU): which of them are far from the HBM roof on their own, as opposed to slowed
down by the weight-gradient stream inside the step.

    python tools/membound_bench.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import ops  # noqa: E402

LEVELS = [(65536, 64), (65536, 128), (23894, 128), (6077, 256), (1450, 512), (361, 768)]
dev = torch.device("cuda", 0)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def row(name, M, C, us, nbytes):
    print(f"{name:28s} M={M:6d} C={C:4d}  {us:8.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / us / 1e6:7.2f} TB/s  ({nbytes / us / 1e6 / 8 * 100:4.1f} % of 8 TB/s)")


for M, C in LEVELS:
    g = torch.Generator(device="cuda").manual_seed(M + C)
    x, dy, add = (torch.randn(M, C, device=dev, generator=g) for _ in range(3))
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    F = 4 * M * C
    y, mean, rstd = ops.ln_fwd(x, gam, bet, res=add)
    row("ln_fwd(+res)", M, C, timeit(lambda: ops.ln_fwd(x, gam, bet, res=add)), 3 * F)
    row("ln_fwd", M, C, timeit(lambda: ops.ln_fwd(x, gam, bet)), 2 * F)
    row("ln_bwd(+add)", M, C, timeit(lambda: ops.ln_bwd(dy, x, mean, rstd, gam, add=add)), 4 * F)
    h = ops.Handoff()
    h.arm(0.1, 1234)

    def lnb():
        h.arm(0.1, 1234)
        return ops.ln_bwd(dy, x, mean, rstd, gam, add=add, hand=h)
    row("ln_bwd(+add,+dz)", M, C, timeit(lnb), 5 * F)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    yb, mu, istd = ops.bn_fwd(x, gam, bet, rm, rv, True, ops.ACT_GELU)
    row("bn_fwd (stats+apply)", M, C, timeit(lambda: ops.bn_fwd(x, gam, bet, rm, rv, True, ops.ACT_GELU)), 3 * F)
    sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
    row("  bn stats only", M, C, timeit(lambda: ops._bn_stats(x, sums)), F)
    row("bn_bwd (stats+apply)", M, C, timeit(lambda: ops.bn_bwd(dy, x, mu, istd, gam, bet, True, ops.ACT_GELU)), 5 * F)
    row("dropout", M, C, timeit(lambda: ops.dropout(x, 0.1, 77)), 2 * F)
    row("add", M, C, timeit(lambda: ops.add(x, dy)), 3 * F)
    print()

"""Diagnostic (GPU box): lotus_stream_probe between the package's step streams and a few extra ones, under the current
GPU_MAX_HW_QUEUES — which pairs of HIP streams can hold each other back (share a hardware queue)?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import robot_3dlotus_amd  # noqa: F401
from robot_3dlotus_amd import _capi, parallel

parallel._uniform_stream_priority()
T = parallel.training_stream()
names = ["train", "side", "comm", "fe"]
streams = [_capi.step_stream(n) for n in names]
for i in range(4):
    names.append("x%d" % i)
    streams.append(torch.cuda.Stream(priority=-1))
names.append("null")
streams.append(torch.cuda.default_stream())
for s in streams:
    with torch.cuda.stream(s):
        torch.zeros(1, device="cuda")
torch.cuda.synchronize()
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
print("%-6s" % "" + " ".join("%-5s" % n for n in names) + "   (row = parked, column = other; 1 = independent)")
t0 = time.perf_counter()
for a, na in zip(streams, names):
    row = []
    for b in streams:
        row.append("-" if a is b else str(_capi.query("lotus_stream_probe", a.cuda_stream, b.cuda_stream, 30)))
    print("%-6s" % na + " ".join("%-5s" % r for r in row))
print("%.2f s" % (time.perf_counter() - t0))

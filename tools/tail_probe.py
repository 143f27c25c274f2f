"""Diagnostic: where does a training step end?  HIP events at the end of the critical backward stream and at the end of
the weight-gradient stream (no profiler attached)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import robot_3dlotus_amd
from robot_3dlotus_amd import config as lcfg, ops, synth
from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
import bench
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = SimplePolicyPTV3CA(lcfg.preset("v1")).to(dev).train()
batch = bench.dev_batch(synth.synth_batch(16, 4096, seed=0), dev)
params = list(model.parameters())
ops.set_wgrad_join("end")
hi = torch.cuda.Stream(priority=-1)
torch.cuda.set_stream(hi)
marks = []
orig = ops._end_of_backward
def probe():
    ec = torch.cuda.Event(enable_timing=True); ec.record(torch.cuda.current_stream())
    es = torch.cuda.Event(enable_timing=True)
    if ops.SIDE is not None: es.record(ops.SIDE)
    else: es.record(torch.cuda.current_stream())
    marks[-1].extend([ec, es])
    orig()
ops._end_of_backward = probe
def step():
    e0 = torch.cuda.Event(enable_timing=True); e0.record(torch.cuda.current_stream())
    marks.append([e0])
    for p in params: p.grad = None
    _, losses = model(batch, compute_loss=True, compute_final_action=False)
    ef = torch.cuda.Event(enable_timing=True); ef.record(torch.cuda.current_stream()); marks[-1].append(ef)
    model.prefetch(batch)
    losses["total"].backward()
for _ in range(10): step()
torch.cuda.synchronize(); marks.clear()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("wall %.2f ms/step" % ((time.perf_counter() - t0) * 1e3 / 20))
import statistics as st
fw = [m[0].elapsed_time(m[1]) for m in marks]; cb = [m[0].elapsed_time(m[2]) for m in marks]; sb = [m[0].elapsed_time(m[3]) for m in marks]
nxt = [marks[i][0].elapsed_time(marks[i + 1][0]) for i in range(len(marks) - 1)]
print("from step start: forward done %.2f, critical backward done %.2f, weight-gradient stream done %.2f, next step starts %.2f (ms, medians)"
      % (st.median(fw), st.median(cb), st.median(sb), st.median(nxt)))

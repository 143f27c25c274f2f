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
caught = []
host = []
fdone = []
starts = []
def ec_done():
    return marks[-1][2].query()  # has the GPU already passed the end of the critical backward when the host leaves backward()?
orig = ops._end_of_backward
def probe():
    ec = torch.cuda.Event(enable_timing=True); ec.record(torch.cuda.current_stream())
    es = torch.cuda.Event(enable_timing=True)
    if ops.SIDE is not None: es.record(ops.SIDE)
    else: es.record(torch.cuda.current_stream())
    marks[-1].extend([ec, es])
    orig()
    ej = torch.cuda.Event(enable_timing=True); ej.record(torch.cuda.current_stream()); marks[-1].append(ej)  # behind the join
ops._end_of_backward = probe
def step():
    th = [time.perf_counter()]
    e0 = torch.cuda.Event(enable_timing=True); e0.record(torch.cuda.current_stream())
    marks.append([e0])
    e0b = torch.cuda.Event(enable_timing=True); e0b.record(torch.cuda.current_stream()); starts.append((e0, e0b))
    for p in params: p.grad = None
    _, losses = model(batch, compute_loss=True, compute_final_action=False)
    th.append(time.perf_counter())
    ef = torch.cuda.Event(enable_timing=True); ef.record(torch.cuda.current_stream()); marks[-1].append(ef)
    if os.environ.get("NO_PREFETCH") != "1": model.prefetch(batch)
    losses["total"].backward()
    eb = torch.cuda.Event(enable_timing=True); eb.record(torch.cuda.current_stream()); marks[-1].append(eb)  # backward() returned
    caught.append(ec_done())
    fdone.append(marks[-1][1].query())
    th.append(time.perf_counter()); host.append(th)
    eb2 = torch.cuda.Event(enable_timing=True); eb2.record(torch.cuda.current_stream()); marks[-1].append(eb2)  # a second marker right behind
for _ in range(10): step()
torch.cuda.synchronize(); marks.clear()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("wall %.2f ms/step" % ((time.perf_counter() - t0) * 1e3 / 20))
import statistics as st
fw = [m[0].elapsed_time(m[1]) for m in marks]; cb = [m[0].elapsed_time(m[2]) for m in marks]; sb = [m[0].elapsed_time(m[3]) for m in marks]
nxt = [marks[i][0].elapsed_time(marks[i + 1][0]) for i in range(len(marks) - 1)]
print("host leaves backward() AFTER the GPU finished the critical backward in %d of %d steps" % (sum(caught[-20:]), 20))
print("second marker behind backward(): %.2f" % st.median([m[0].elapsed_time(m[6]) for m in marks]))
print("... and the GPU had finished the FORWARD of that step in %d of 20" % sum(fdone[-20:]))
print("two markers back to back at the start of a step: %.3f ms apart; last marker of step i -> first marker of step i + 1: %.3f ms" % (st.median([a.elapsed_time(b) for a, b in starts[-20:]]), st.median([marks[i][6].elapsed_time(marks[i + 1][0]) for i in range(len(marks) - 1)])))
hh = host[-20:]
print("host: forward() %.2f ms, prefetch + backward() %.2f ms, between steps %.2f ms (medians; the step period is the wall time above)" % (st.median([h[1] - h[0] for h in hh]) * 1e3, st.median([h[2] - h[1] for h in hh]) * 1e3, st.median([b[0] - a[2] for a, b in zip(hh, hh[1:])]) * 1e3))
print("behind the join %.2f, backward() returned %.2f" % (st.median([m[0].elapsed_time(m[4]) for m in marks]), st.median([m[0].elapsed_time(m[5]) for m in marks])))
print("from step start: forward done %.2f, critical backward done %.2f, weight-gradient stream done %.2f, next step starts %.2f (ms, medians)"
      % (st.median(fw), st.median(cb), st.median(sb), st.median(nxt)))

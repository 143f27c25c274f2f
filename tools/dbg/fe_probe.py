"""Diagnostic: when does the prefetched front-end of step k + 1 finish on the GPU, relative to step k's backward pass, in the
one-rank RCCL rehearsal of the data-parallel step?  HIP events, no profiler.  LOTUS_FORCE_COLLECTIVES=1 python tools/dbg/fe_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import robot_3dlotus_amd
from robot_3dlotus_amd import config as lcfg, ops, parallel, synth
from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
import bench
dev = torch.device("cuda", 0)
hi = torch.cuda.Stream(priority=-1) if os.environ.get("LOTUS_HIPRIO", "1") == "1" else torch.cuda.current_stream()
parallel.init_distributed()
torch.manual_seed(0)
model = SimplePolicyPTV3CA(lcfg.preset("v1")).to(dev).train()
reducer = None
if os.environ.get("LOTUS_FORCE_COLLECTIVES") == "1":
    reducer = parallel.GradReducer(model, bucket_mb=32.0)
    parallel.enable_sync_batchnorm()
batch = bench.dev_batch(synth.synth_batch(16, 4096, seed=0), dev)
params = list(model.parameters())
ops.set_wgrad_join("end")
torch.cuda.synchronize()
torch.cuda.set_stream(hi)
marks, hostw = [], []
def ev(stream=None):
    e = torch.cuda.Event(enable_timing=True); e.record(stream if stream is not None else torch.cuda.current_stream()); return e
def step():
    e0 = ev()
    if reducer is not None: reducer.zero_grad()
    else:
        for p in params: p.grad = None
    t0 = time.perf_counter()
    _, losses = model(batch, compute_loss=True, compute_final_action=False)
    t1 = time.perf_counter()
    ef = ev()
    model.prefetch(batch)
    efe = ev(model.ptv3_model._fe_stream)
    losses["total"].backward()
    eb = ev()
    if reducer is not None: reducer.finish()
    ee = ev()
    marks.append((e0, ef, efe, eb, ee)); hostw.append(t1 - t0)
for _ in range(10): step()
torch.cuda.synchronize(); marks.clear(); hostw.clear()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("wall %.2f ms/step, host forward %.2f ms" % ((time.perf_counter() - t0) * 1e3 / 20, 1e3 * sum(hostw) / len(hostw)))
import statistics as st
med = lambda f: st.median(f(m) for m in marks)
print("GPU ms from step start: forward end %.2f, front-end of the next batch done %.2f, backward end %.2f, finish end %.2f" % (
    med(lambda m: m[0].elapsed_time(m[1])), med(lambda m: m[0].elapsed_time(m[2])), med(lambda m: m[0].elapsed_time(m[3])), med(lambda m: m[0].elapsed_time(m[4]))))
print("next step starts %.2f ms after this one" % st.median(marks[i][0].elapsed_time(marks[i + 1][0]) for i in range(len(marks) - 1)))

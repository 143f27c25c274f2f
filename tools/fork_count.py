"""Diagnostic: event-record forks issued from the Python path per training step, by call site."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import robot_3dlotus_amd
from robot_3dlotus_amd import config as lcfg, ops, synth, _capi
from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
import bench
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = SimplePolicyPTV3CA(lcfg.preset("v1")).to(dev).train()
batch = bench.dev_batch(synth.synth_batch(16, 4096, seed=0), dev)
ops.set_wgrad_join("end")
sites = collections.Counter()
real = _capi.call_raw
def call_raw(name, *a):
    if name == "lotus_streamlink_wait":
        st = traceback.extract_stack(limit=8)
        sites[" < ".join(f"{f.name}:{f.lineno}" for f in reversed(st[:-1]) if "ops.py" in f.filename or "ptv3" in f.filename or "policy" in f.filename)[:160]] += 1
    return real(name, *a)
def step():
    for p in model.parameters(): p.grad = None
    _, losses = model(batch, compute_loss=True, compute_final_action=False)
    model.prefetch(batch)
    losses["total"].backward()
for _ in range(3): step()
torch.cuda.synchronize()
_capi.call_raw = call_raw
ops._capi.call_raw = call_raw
step(); torch.cuda.synchronize()
print("forks per step from Python:", sum(sites.values()))
for k, v in sites.most_common(30): print(v, k)

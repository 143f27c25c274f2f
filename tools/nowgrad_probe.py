"""Diagnostic: how much does the weight-gradient stream cost the critical stream?  Runs the training step (per-launch
Python path) three ways: as is; with the dense / sparse-conv weight-gradient launches skipped (gradients left
uninitialised: timing only); and with the side stream disabled (everything serial on one stream)."""
import os, sys, time
os.environ["LOTUS_PY_BLOCKS"] = "1"
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
hi = torch.cuda.Stream(priority=-1)
torch.cuda.set_stream(hi)
def step():
    for p in params: p.grad = None
    _, losses = model(batch, compute_loss=True, compute_final_action=False)
    ef = torch.cuda.Event(enable_timing=True); ef.record(torch.cuda.current_stream())
    model.prefetch(batch)
    losses["total"].backward()
    return ef
def run(tag):
    for _ in range(8): step()
    torch.cuda.synchronize()
    fw = []
    t0 = time.perf_counter()
    for _ in range(20):
        e0 = torch.cuda.Event(enable_timing=True); e0.record(torch.cuda.current_stream())
        fw.append((e0, step()))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 20
    f = sorted(a.elapsed_time(b) for a, b in fw)[10]
    print("%-28s %.2f ms/step  (forward %.2f, backward %.2f)" % (tag, ms, f, ms - f), flush=True)
run("as is")
real_call = ops.call
SKIP = {"lotus_linear_wgrad", "lotus_subm_conv_wgrad", "lotus_layernorm_bwd_params"}
def call(name, *a):
    if name in SKIP: return 0
    return real_call(name, *a)
ops.call = call
run("no weight-gradient launches")
ops.call = real_call
ops.enable_side_stream(False)
run("side stream off")

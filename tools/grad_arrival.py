"""Diagnostic: in which order do parameter gradients arrive during backward, and when can each GradReducer bucket
flush?  (Bucket order should follow arrival order or a late gradient holds back a whole bucket's all-reduce.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import robot_3dlotus_amd  # noqa: F401
from robot_3dlotus_amd import config as lcfg, parallel, synth
from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
import bench

dev = torch.device("cuda", 0)
model = SimplePolicyPTV3CA(lcfg.preset("v1")).to(dev).train()
names = {p: n for n, p in model.named_parameters()}
red = parallel.GradReducer(model, bucket_mb=32.0)
arrival = []
for p in red.params:
    p.register_post_accumulate_grad_hook(lambda q: arrival.append(q))
batch = bench.dev_batch(synth.synth_batch(4, 2048, seed=0), dev)
for it in range(2):   # the reducer learns the arrival order in the first pass and re-lays its buckets
    arrival.clear()
    red.zero_grad()
    _, losses = model(batch, compute_loss=True, compute_final_action=False)
    losses["total"].backward()
    red.finish()
    torch.cuda.synchronize()
pos = {p: i for i, p in enumerate(arrival)}
n = len(arrival)
for b, ps in enumerate(red._bparams):
    last = max(ps, key=lambda p: pos[p])
    numel = sum(p.numel() for p in ps)
    print(f"bucket {b}: {len(ps):3d} tensors {numel * 4 / 1e6:6.1f} MB  complete after arrival {pos[last] + 1}/{n}  (last: {names[last]})")

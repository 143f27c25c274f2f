"""Diagnostic: cProfile of the host side of training steps with a batch so small that the GPU is idle (1 cloud x 256 points):
where the ~10 ms of host time per step go.   python tools/host_profile.py [peract]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import config as lcfg, ops, synth  # noqa: E402
from robot_3dlotus_amd.policy import SimplePolicyPTV3CA  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = SimplePolicyPTV3CA(lcfg.preset("v1")).to(dev).train()
if len(sys.argv) > 1 and sys.argv[1] == "peract":
    model.act_storage = "bf16"
batch = bench.dev_batch(synth.synth_batch(1, 256, seed=0), dev)
params = list(model.parameters())
ops.set_wgrad_join("end")


def step():
    for p in params:
        p.grad = None
    _, losses = model(batch, compute_loss=True, compute_final_action=False)
    model.prefetch(batch)
    losses["total"].backward()


for _ in range(8):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

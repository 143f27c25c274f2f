"""Diagnostic: cProfile of the host side of the v1 training step (enqueue only; the GPU runs behind)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import robot_3dlotus_amd
from robot_3dlotus_amd import config as lcfg, synth
from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
model = SimplePolicyPTV3CA(lcfg.preset("v1")).to(dev).train()
batch = bench.dev_batch(synth.synth_batch(16, 4096, seed=0), dev)
params = list(model.parameters())
def step():
    for p in params: p.grad = None
    _, losses = model(batch, compute_loss=True, compute_final_action=False)
    model.prefetch(batch)
    losses["total"].backward()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/10:.2f} ms/step, with drain {1e3*(t2-t0)/10:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)

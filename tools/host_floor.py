"""Host floor of the v1 training step: enqueue time per step with a batch so small that the GPU is idle most of the time
(1 cloud x 256 points), split by forward / backward and by autograd Function body.  LOTUS_PY_BLOCKS=1 selects the
per-launch host path for comparison.

    python tools/host_floor.py
"""
import collections
import os
import sys
import time

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
batch = bench.dev_batch(synth.synth_batch(1, 256, seed=0), dev)
params = list(model.parameters())
ops.set_wgrad_join("end")
T, N = collections.defaultdict(float), collections.Counter()


def wrap(cls):
    f0, b0 = cls.forward, cls.backward

    def fw(ctx, *a):
        t = time.perf_counter(); r = f0(ctx, *a); T[cls.__name__ + ".fwd"] += time.perf_counter() - t; N[cls.__name__ + ".fwd"] += 1; return r

    def bw(ctx, *a):
        t = time.perf_counter(); r = b0(ctx, *a); T[cls.__name__ + ".bwd"] += time.perf_counter() - t; N[cls.__name__ + ".bwd"] += 1; return r

    cls.forward, cls.backward = staticmethod(fw), staticmethod(bw)


for c in (ops.CpeFn, ops.FfnFn, ops.SelfAttnFn, ops.CrossAttnFn, ops.StemFn, ops.PoolFn, ops.UnpoolFn, ops.HeadLossFn, ops.LinearFn):
    wrap(c)
tf = tb = 0.0


def step():
    global tf, tb
    t0 = time.perf_counter()
    for p in params:
        p.grad = None
    _, losses = model(batch, compute_loss=True, compute_final_action=False)
    model.prefetch(batch)
    t1 = time.perf_counter()
    losses["total"].backward()
    t2 = time.perf_counter()
    tf += t1 - t0
    tb += t2 - t1


for _ in range(8):
    step()
torch.cuda.synchronize()
T.clear(); N.clear(); tf = tb = 0.0
n = 30
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"composites={'on' if ops.composites_enabled() else 'off'}: host {1e3 * (t1 - t0) / n:.2f} ms/step (fwd {1e3 * tf / n:.2f} + bwd {1e3 * tb / n:.2f})")
tot = 0.0
for k in sorted(T, key=lambda k: -T[k]):
    print(f"  {k:18s} {1e3 * T[k] / n:7.3f} ms/step  {N[k] / n:5.1f} calls  {1e6 * T[k] / N[k]:7.1f} us/call")
    tot += T[k]
print(f"  inside Function bodies: {1e3 * tot / n:.2f} ms/step")

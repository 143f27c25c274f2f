"""Diagnostic: the policy with bf16 activation storage against its own fp32 run (same weights, dropout off), per-launch and
composite paths, twice each.  HIP_LAUNCH_BLOCKING=1 reports a device fault at the Python line of the launch that caused it.
    python tools/bf16_probe.py [tiny|peract]"""
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import config as lcfg, ops, synth  # noqa: E402
from robot_3dlotus_amd.policy import SimplePolicyPTV3CA  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = lcfg.preset(which)
torch.manual_seed(0)
m = SimplePolicyPTV3CA(cfg).cuda().train()
m.ptv3_model.proj_drop = m.ptv3_model.attn_drop = 0.0
m.act_proj_head.dropout = 0.0
m.ptv3_model.order_perms = [[1, 3, 0, 2], [0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1], [1, 0, 2, 3]][:m.ptv3_model.num_stages]
if os.environ.get("LOTUS_SIDE_STREAM") == "0":
    ops.enable_side_stream(False)
B, n = (2, 512) if which == "tiny" else (2, 4096)
batch = synth.synth_batch(B, n, seed=1)
dev = {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if k == "disc_pos_probs" else v)) for k, v in batch.items()}


def run(storage, comp):
    m.act_storage = storage
    ops.set_composites(comp)
    for p in m.parameters():
        p.grad = None
    _, losses = m(dict(dev), compute_loss=True, compute_final_action=False)
    torch.cuda.synchronize()
    losses["total"].backward()
    torch.cuda.synchronize()
    g = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    return float(losses["total"]), g, {n_: p.grad.clone() for n_, p in m.named_parameters()}


ref_l, ref_g, ref_d = run(None, True)
print("fp32 composite: loss", ref_l, "|g|", float(ref_g.norm()), flush=True)
for storage, comp in ((None, False), ("bf16", False), ("bf16", False), ("bf16", True), ("bf16", True)):
    l, g, d = run(storage, comp)
    rel = float((g - ref_g).norm() / ref_g.norm())
    worst = sorted(((float((d[k] - ref_d[k]).norm() / (ref_d[k].norm() + 1e-3 * ref_g.norm())), k) for k in d), reverse=True)[:4]
    print(storage, "composite" if comp else "per-launch", "loss", l, "|g|", float(g.norm()), "rel", rel, "finite", bool(torch.isfinite(g).all()),
          worst, flush=True)

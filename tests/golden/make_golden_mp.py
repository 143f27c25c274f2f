"""Golden fixtures of the 3D-LOTUS++ motion planner (BASELINE configs[3]) from the *imported reference*
(build container only):

    python tests/golden/make_golden_mp.py         # writes tests/golden/mp_*.npz

Same recipe as make_golden.py (stand-in native deps, half() neutralised, dropout-free, seeds instead of tensors);
the integer front-end tables are not stored again — the motion planner shares them with the policy.  Each fixture
holds the recorded shuffle permutations, the head outputs xt [T,3,N,nb] / xr [B,T,72,3] / xo / xstop, the five
losses, every parameter-gradient norm + leading entries (parameters the reference leaves without gradient are
listed under `nograd`), and the BatchNorm running statistics after the step.
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import ref_harness as rh  # noqa: E402
from make_golden import GRAD_KEYS_SAMPLE, zero_dropouts  # noqa: E402
from weights_util import seeded_state_dict  # noqa: E402
import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import synth  # noqa: E402

CASES = {
    # name: (variant, batch, npoints, ragged, data_seed, weight_seed, weight_variant, train)
    "mp_tiny_scaled_train": ("mp_tiny", 2, 512, True, 31, 5, "scaled", True),
    "mp_init_train": ("mp", 2, 768, True, 32, 6, "init", True),
    "mp_scaled_eval": ("mp", 2, 768, True, 33, 7, "scaled", False),
    # the YAML's own use_ee_pose = True: one RobotPoseEmbedding token appended to every cloud's instruction tokens
    "mp_tinyctx_scaled_train": ("mp_tinyctx", 2, 512, True, 34, 8, "scaled", True),
}


def run_case(name, spec):
    variant, B, n, ragged, dseed, wseed, wvar, train = spec
    torch.manual_seed(0)
    ref, cfg = rh.build_reference_mp(variant)
    sd = seeded_state_dict(ref.state_dict(), wseed, wvar)
    ref.load_state_dict(sd, strict=True)
    zero_dropouts(ref)
    ref.train(train)
    batch = synth.synth_batch_mp(B, n, ragged=ragged, seed=dseed)
    head = {}
    hh = ref.act_proj_head.register_forward_hook(
        lambda mod, i, o: head.update(xt=o[0], xr=o[1], xo=o[2], xstop=o[3]))
    feats = []
    fh = [m.register_forward_hook(lambda mod, i, o: feats.append(o.feat.detach().clone()))
          for nme, m in ref.ptv3_model.named_modules() if nme.endswith("ca_block0")]
    perms = []
    with rh.neutralise_half(), rh.record_randperm(perms):
        torch.manual_seed(100 + dseed)
        losses = rh.reference_forward_mp(ref, copy.deepcopy(batch))
    for p in ref.parameters():
        p.grad = None
    losses["total"].backward()
    out = {"meta_variant": variant, "meta_B": B, "meta_n": n, "meta_ragged": ragged, "meta_dseed": dseed,
           "meta_wseed": wseed, "meta_wvar": wvar, "meta_train": train,
           "perms": torch.stack(perms).numpy().astype(np.int64),
           "npoints_in_batch": np.array(batch["npoints_in_batch"]),
           "input_checksum": np.float64(batch["pc_fts"].double().sum().item()
                                        + batch["pc_labels"].double().sum().item()),
           "weight_checksum": np.float64(sum(v.double().sum().item() for v in sd.values()))}
    out["feat_last_norm"] = np.float64(feats[-1].double().norm().item())
    for k in ("xt", "xr", "xo", "xstop"):
        out[k] = head[k].detach().numpy()
    for k, v in losses.items():
        out["loss_" + k] = np.float32(v.item())
    nograd = []
    for nme, p in ref.named_parameters():
        if p.grad is None:
            nograd.append(nme)
            continue
        g = p.grad.detach()
        out["gnorm/" + nme] = np.float64(g.double().norm().item())
        out["ghead/" + nme] = g.flatten()[:GRAD_KEYS_SAMPLE].numpy()
    out["nograd"] = np.array(nograd)
    if train:
        for nme, b in ref.named_buffers():
            if "running" in nme:
                out["buf/" + nme] = b.detach().numpy()
    for h in fh + [hh]:
        h.remove()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB  losses="
          f"{ {k: round(float(v), 5) for k, v in losses.items()} }  |xt|max={head['xt'].abs().max():.3f} "
          f"nograd={nograd} perms={out['perms'].tolist()}")


if __name__ == "__main__":
    only = sys.argv[1:] or list(CASES)
    for nme in only:
        run_case(nme, CASES[nme])

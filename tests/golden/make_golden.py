"""Generate the golden parity fixtures from the *imported reference* (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference (/root/reference) is imported through tests/golden/ref_harness.py (stand-ins for the
un-installed spconv / flash_attn / torch_scatter / timm / addict / easydict / yacs; half() is
neutralised => "O-fp32-ideal" flavour of SURVEY.md §8c).  Fixtures hold DATA only: the seeds and
sizes that rebuild the inputs and weights bit-identically on any box (tests/weights_util.py,
robot-3dlotus_amd/synth.py), the recorded shuffle permutations, the integer tables of every level
captured from the reference's own `Point` objects, and the expected float outputs.

Cases (name -> config, clouds, weights, mode):
  tiny_init_eval / tiny_scaled_train      BASELINE configs[0]: 1 cloud x 512 pts, 2-stage model
  v1_init_train / v1_scaled_train / v1_scaled_eval   v1 model, 2 clouds (1024, ragged)
An "fp16-faithful" record (attention operands rounded through fp16 as the reference's GPU path
does, model.py:544 / model_ca.py:63) is stored next to each fp32-ideal output to document the gap.
"""
import copy
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import ref_harness as rh  # noqa: E402
from weights_util import seeded_state_dict  # noqa: E402
import robot_3dlotus_amd  # noqa: E402,F401
from robot_3dlotus_amd import synth  # noqa: E402

CASES = {
    # name: (variant, batch, npoints, ragged, data_seed, weight_seed, weight_variant, train)
    "tiny_init_eval": ("tiny", 1, 512, False, 11, 0, "init", False),
    "tiny_scaled_train": ("tiny", 1, 512, False, 12, 1, "scaled", True),
    "v1_init_train": ("v1", 2, 1024, True, 13, 2, "init", True),
    "v1_scaled_train": ("v1", 2, 1024, True, 14, 3, "scaled", True),
    "v1_scaled_eval": ("v1", 2, 1024, True, 15, 3, "scaled", False),
    # stages deeper than one Block (enc_depths [2, 5], dec_depths [2] on the tiny width: order_index = i % 4 wraps); the eval
    # case carries the YAML's drop_path 0.1 (DropPath is the identity in eval mode, model.py:655-657)
    "tinydeep_scaled_train": ("tinydeep", 2, 700, True, 16, 4, "scaled", True),
    "tinydeep_scaled_eval": ("tinydeep", 2, 700, True, 17, 4, "scaled", False, 0.1),
    # use_ee_pose + use_step_id: one pose token and one step token appended to every cloud's instruction tokens
    "tinyctx_scaled_train": ("tinyctx", 3, 600, True, 18, 5, "scaled", True),
}
GRAD_KEYS_SAMPLE = 48  # leading entries of every gradient kept besides its norm


def zero_dropouts(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(getattr(m, "attn_drop", None), float):
            m.attn_drop = 0.0


def run_case(name, spec, out_dir=HERE):
    variant, B, n, ragged, dseed, wseed, wvar, train = spec[:8]
    drop_path = spec[8] if len(spec) > 8 else 0.0
    torch.manual_seed(0)
    ref, cfg = rh.build_reference_policy(variant, drop_path=drop_path)
    sd = seeded_state_dict(ref.state_dict(), wseed, wvar)
    ref.load_state_dict(sd, strict=True)
    zero_dropouts(ref)
    ref.train(train)
    batch = synth.synth_batch(B, n, ragged=ragged, seed=dseed)

    levels = []

    def hook(mod, inp, outp):
        p = outp
        lv = dict(grid=p.grid_coord, batch=p.batch, code=p.serialized_code, order=p.serialized_order,
                  inverse=p.serialized_inverse, pad=p.pad, unpad=p.unpad, cu_seqlens=p.cu_seqlens_key,
                  depth=torch.tensor(p.serialized_depth), feat=None)
        if "pooling_inverse" in p.keys():
            lv["cluster"] = p.pooling_inverse
        nb = p.sparse_conv_feat.indice_dict
        for (key, k), t in nb.items():
            lv[f"nbr{k ** 3}"] = t.int()
        levels.append({k: v.detach().clone() for k, v in lv.items() if v is not None})

    hooks = [m.register_forward_hook(hook) for nme, m in ref.ptv3_model.enc.named_modules()
             if nme.endswith("block0") and "ca_" not in nme]
    feats = []
    fhooks = [m.register_forward_hook(lambda mod, i, o: feats.append(o.feat.detach().clone()))
              for nme, m in ref.ptv3_model.named_modules() if re.search(r"ca_block\d+$", nme)]
    head = {}
    hh = ref.act_proj_head.register_forward_hook(lambda mod, i, o: head.update(xt=o[0], xr=o[1], xo=o[2]))

    perms = []
    with rh.neutralise_half(), rh.record_randperm(perms):
        torch.manual_seed(100 + dseed)
        losses = rh.reference_forward(ref, copy.deepcopy(batch), full=(variant == "v1"))
    for p in ref.parameters():
        p.grad = None
    losses["total"].backward()
    out = {"meta_variant": variant, "meta_B": B, "meta_n": n, "meta_ragged": ragged, "meta_dseed": dseed,
           "meta_wseed": wseed, "meta_wvar": wvar, "meta_train": train, "meta_drop_path": np.float64(drop_path),
           "perms": torch.stack(perms).numpy().astype(np.int64),
           "npoints_in_batch": np.array(batch["npoints_in_batch"]),
           "input_checksum": np.float64(batch["pc_fts"].double().sum().item()),
           "weight_checksum": np.float64(sum(v.double().sum().item() for v in sd.values()))}
    for s, lv in enumerate(levels):
        for k, v in lv.items():
            out[f"L{s}_{k}"] = v.numpy()
    for i, f in enumerate(feats):
        out[f"feat{i}_norm"] = np.float64(f.double().norm().item())
        out[f"feat{i}_head"] = f[:16, :16].numpy()
    out["feat_last"] = feats[-1].numpy()
    for k in ("xt", "xr", "xo"):
        out[k] = head[k].detach().numpy()
    for k, v in losses.items():
        out["loss_" + k] = np.float32(v.item())
    for nme, p in ref.named_parameters():
        g = p.grad.detach()
        out["gnorm/" + nme] = np.float64(g.double().norm().item())
        out["ghead/" + nme] = g.flatten()[:GRAD_KEYS_SAMPLE].numpy()
    if train:
        for nme, b in ref.named_buffers():
            if "running" in nme:
                out["buf/" + nme] = b.detach().numpy()

    # fp16-faithful flavour (documents the reference's GPU numerics; not the parity target)
    for h in hooks + fhooks + [hh]:
        h.remove()
    ref.load_state_dict(sd, strict=True)
    head16 = {}
    hh = ref.act_proj_head.register_forward_hook(lambda mod, i, o: head16.update(xt=o[0], xr=o[1], xo=o[2]))
    with torch.no_grad():
        torch.manual_seed(100 + dseed)
        rh.reference_forward(ref, copy.deepcopy(batch), full=(variant == "v1"))
    hh.remove()
    for k in ("xt", "xr", "xo"):
        out["fp16_gap_" + k] = np.float32((head16[k] - head[k]).abs().max().item())
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **out)
    gaps = {k: float(out["fp16_gap_" + k]) for k in ("xt", "xr", "xo")}
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB  losses="
          f"{ {k: round(float(v), 5) for k, v in losses.items()} }  |xt|max={head['xt'].abs().max():.3f} "
          f"|xr|max={head['xr'].abs().max():.3f}  fp16 gap={gaps}  perms={out['perms'].tolist()}")


if __name__ == "__main__":
    only = sys.argv[1:] or list(CASES)
    for nme in only:
        run_case(nme, CASES[nme])

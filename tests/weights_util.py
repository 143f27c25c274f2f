"""Deterministic, platform-independent parameter sets for parity fixtures (test infrastructure).

Weights are drawn with numpy's PCG64 from a seed, key by key in sorted order, so that the
fixture generator (build container, reference model) and the GPU-box tests (HIP model, oracle)
rebuild bit-identical state dicts without shipping the tensors.

variant 'init'   : the reference's init regime (base.py:36-49): Linear ~ N(0, 0.02) (trunc-normal
                   at +-2 sigma is indistinguishable at this std), zero biases, LN = (1, 0),
                   BN = (1, 0, mean 0, var 1), conv ~ U(+-1/sqrt(fan_in)).
variant 'scaled' : SURVEY.md Trap 2 — weights x3, non-zero biases, non-trivial LN/BN affine and
                   running statistics, so softmax / qk-norm / GELU / BN leave their linear regime.
"""
import numpy as np
import torch


def seeded_state_dict(template, seed=0, variant="init"):
    """template: dict name -> tensor (shapes/dtypes are read; values ignored)."""
    rng = np.random.default_rng(seed)
    scaled = variant == "scaled"
    out = {}
    for name in sorted(template.keys()):
        ref = template[name]
        shape = tuple(ref.shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros(shape, dtype=torch.long)
            continue
        leaf = name.rsplit(".", 1)[-1]
        is_norm = (ref.ndim == 1 and leaf == "weight")
        if leaf == "running_mean":
            v = rng.normal(0, 0.1, shape) if scaled else np.zeros(shape)
        elif leaf == "running_var":
            v = rng.uniform(0.5, 1.5, shape) if scaled else np.ones(shape)
        elif is_norm:
            v = 1 + rng.normal(0, 0.1, shape) if scaled else np.ones(shape)
        elif leaf == "bias":
            v = rng.normal(0, 0.1, shape) if scaled else np.zeros(shape)
        elif ref.ndim == 5:  # sparse conv (Cout,k,k,k,Cin)
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            b = (3.0 if scaled else 1.0) / np.sqrt(fan_in)
            v = rng.uniform(-b, b, shape)
        else:
            v = rng.normal(0, 0.06 if scaled else 0.02, shape)
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape))
    return out

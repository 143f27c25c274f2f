"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports exactly
the symbols include/lotus_hip.h declares; the product refuses to run without a HIP device."""
import os
import re
import subprocess

import pytest
import torch

import robot_3dlotus_amd  # noqa: F401
from robot_3dlotus_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge

    return ge.build()


def test_library_exports_every_declared_symbol(built):
    protos = _capi.parse_header()
    assert len(protos) >= 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", built], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("lotus_")}
    assert set(protos) == exported, (set(protos) ^ exported)
    L = _capi.lib()
    assert L.fn["lotus_abi_version"]() == _capi.ABI_VERSION == 3
    assert L.last_error() == ""


def test_header_lists_every_environment_switch_of_the_library(built):
    """The C-ABI's only process-wide inputs are the environment switches include/lotus_hip.h lists: the
    LOTUS_* strings the library contains are exactly that list, and no exported symbol is missing from the headers."""
    import re
    import subprocess

    blob = open(os.path.join(ROOT, "robot-3dlotus_amd", "csrc", "liblotus_hip.so"), "rb").read()
    in_lib = set(m.decode() for m in re.findall(rb"LOTUS_[A-Z0-9_]{2,}", blob))
    in_lib = {s for s in in_lib if not s.startswith("LOTUS_E_") and s not in ("LOTUS_ACT_BF16",)}
    head = open(os.path.join(ROOT, "include", "lotus_hip.h")).read()
    listed = set(re.findall(r"^ \*\s+(LOTUS_[A-Z0-9_]+)=", head, re.M))
    assert listed == in_lib, (sorted(listed - in_lib), sorted(in_lib - listed))
    assert len(listed) <= 15
    syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "robot-3dlotus_amd", "csrc", "liblotus_hip.so")],
                          capture_output=True, text=True).stdout
    assert "lotus_tls_stop_event" not in syms
    # no unmangled export besides the C-ABI itself (a __global__ inside an extern "C" block exports its launch stub: VERDICT r5)
    stray = [l.split()[-1] for l in syms.splitlines() if " T " in l and not l.split()[-1].startswith(("lotus_", "_Z", "_init", "_fini"))]
    assert not stray, stray


def test_python_side_switches_are_documented():
    """The LOTUS_* environment variables the Python host (and bench.py) reads are listed in INTEGRATION.md;
    the library's own twelve are in include/lotus_hip.h."""
    import glob
    import re

    used = set()
    for f in glob.glob(os.path.join(ROOT, "robot-3dlotus_amd", "*.py")) + [os.path.join(ROOT, "bench.py")]:
        used |= set(re.findall(r"[\"'](LOTUS_[A-Z0-9_]+)[\"']", open(f).read()))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(v for v in used if v not in doc)
    assert not missing, missing


def test_trampoline_module_covers_the_header(built):
    """csrc/_lotus_fastcall.so (generated from the header) exposes one METH_FASTCALL function per prototype, argument
    conversion included: pointers accept int | None | objects with data_ptr(), wrong arity raises TypeError."""
    F = _capi.fastcall()
    assert F is not None
    protos = _capi.parse_header()
    assert all(callable(getattr(F, n, None)) for n in protos), [n for n in protos if not hasattr(F, n)]
    assert F.lotus_abi_version() == _capi.ABI_VERSION and F.lotus_last_error() == ""
    assert F.lotus_linear_wgrad_workspace(65536, 256, 64) == _capi.lib().fn["lotus_linear_wgrad_workspace"](65536, 256, 64)
    with pytest.raises(TypeError):
        F.lotus_add(None, None)
    t = torch.zeros(8)
    assert F.lotus_add(t, t, None, 8, 0) < 0 and "lotus_add" in F.lotus_last_error()      # argument check, no launch


def test_header_cites_reference_for_every_entry():
    src = open(_capi.HEADER_PATH).read()
    assert "model.py" in src and "model_ca.py" in src and "simple_policy_ptv3.py" in src
    assert src.count("extern \"C\"") == 1


def test_workspace_queries_run_without_gpu(built):
    assert _capi.query("lotus_linear_wgrad_workspace", 65536, 256, 64) > 0
    assert _capi.query("lotus_fe_sort_workspace", 65536) > 4 * 65536 * 12
    assert _capi.query("lotus_subm_conv_wgrad_workspace", 65536, 27, 64, 64) >= 32 * 64 * 27 * 64 * 4
    assert _capi.query("lotus_attention_bwd_workspace", 512, 2) == 512 * 2 * 4 * 32 * 4


def test_no_kernel_source_uses_compat_layers():
    csrc = os.path.join(os.path.dirname(_capi.LIB_PATH))
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".cpp")):
            s = open(os.path.join(csrc, f)).read()
            assert "__HIP_PLATFORM" not in s and "cuda_runtime" not in s and "triton" not in s.lower(), f


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_product_fails_loudly_without_gpu():
    from robot_3dlotus_amd import config as lcfg, synth
    from robot_3dlotus_amd.policy import SimplePolicyPTV3CA

    m = SimplePolicyPTV3CA(lcfg.preset("tiny"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(synth.synth_batch(1, 128, seed=0), compute_loss=True)


def test_product_does_not_import_the_oracle():
    pkg = os.path.dirname(_capi.LIB_PATH)
    pkg = os.path.dirname(pkg)
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            s = open(os.path.join(pkg, f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", s, flags=re.M), f


def test_abi_has_no_process_wide_precision_state(built):
    """no global mutable state in the library — operand precision is an argument of every dense / conv /
    attention entry point (it used to be lotus_set_gemm_precision + a static)."""
    protos = _capi.parse_header()
    assert "lotus_set_gemm_precision" not in protos and "lotus_get_gemm_precision" not in protos
    for name in ("lotus_linear_fwd", "lotus_linear_dgrad", "lotus_linear_wgrad", "lotus_conv_weight_transpose",
                 "lotus_subm_conv", "lotus_attention_fwd", "lotus_attention_bwd"):
        assert "precision" in protos[name][2], name


_CTYPES_SMOKE = r"""
import os, sys
sys.path[:0] = [{root!r}, os.path.join({root!r}, "tests")]
import numpy as np, torch
import golden_util as gu
from oracle.model import Oracle
import robot_3dlotus_amd
from robot_3dlotus_amd import _capi, config as lcfg, synth
from robot_3dlotus_amd.policy import SimplePolicyPTV3CA
from weights_util import seeded_state_dict
assert _capi.fastcall() is None, "LOTUS_NO_FASTCALL=1 must select the ctypes binding"
cfg = lcfg.preset("tiny")
sd = seeded_state_dict(gu.state_template(cfg), 5, "scaled")
batch = synth.synth_batch(2, 512, ragged=True, seed=42)
perms = [[2, 0, 1, 3], [1, 3, 0, 2]]
ref = Oracle({{k: v.clone() for k, v in sd.items()}}, lcfg.plain(cfg), training=True).forward(batch, perms)
m = SimplePolicyPTV3CA(cfg); m.load_state_dict(sd); m = m.cuda().train()
m.ptv3_model.proj_drop = m.ptv3_model.attn_drop = 0.0; m.act_proj_head.dropout = 0.0
m.ptv3_model.order_perms = perms
dev = {{k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if k == "disc_pos_probs" else v)) for k, v in batch.items()}}
_, losses = m(dev, compute_loss=True, compute_final_action=False)
losses["total"].backward(); torch.cuda.synchronize()
err = float(np.abs(m.last_pred[0].detach().cpu().numpy() - ref["xt"].numpy()).max())
assert err <= 1e-4 * max(1.0, float(np.abs(ref["xt"].numpy()).max())), err
assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
print("ctypes-path ok", err)
"""


@pytest.mark.gpu
def test_documented_ctypes_binding_runs_the_model_on_the_gpu(built):
    """INTEGRATION.md documents ctypes as THE FFI; every other GPU test goes through the generated trampolines.  Run one
    forward + backward of the policy with LOTUS_NO_FASTCALL=1 (pure ctypes) against the oracle."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CTYPES_SMOKE.format(root=root)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, LOTUS_NO_FASTCALL="1"))
    assert r.returncode == 0 and "ctypes-path ok" in r.stdout, r.stderr[-3000:]


def test_objects_do_not_depend_on_the_build_directory(tmp_path):
    """a library rebuilt from the same sources in another checkout must hash like the one the committed
    counter profiles were taken on.  hipcc derives its compilation-unit id from the absolute input path unless told otherwise;
    build.py therefore compiles with relative names and a fixed -cuid.  One small translation unit, two directories."""
    import hashlib
    import shutil
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "robot-3dlotus_amd", "csrc")
    sys.path.insert(0, csrc)
    import build as lbuild

    if not os.path.exists(lbuild.HIPCC):
        pytest.skip("hipcc not installed")
    hashes = []
    for name in ("a", "deeper/nested/b"):
        d = tmp_path / name / "robot-3dlotus_amd" / "csrc"
        d.mkdir(parents=True)
        inc = tmp_path / name / "include"
        inc.mkdir()
        for f in ("optim.hip", "common.h"):
            shutil.copy(os.path.join(csrc, f), d / f)
        cmd = [lbuild.HIPCC] + lbuild.FLAGS + ["-cuid=lotus-optim", f"-ffile-prefix-map={d}=.", "-x", "hip", "-c", "optim.hip", "-o", "optim.o"]
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        hashes.append(hashlib.sha256((d / "optim.o").read_bytes()).hexdigest())
    assert hashes[0] == hashes[1]


@pytest.mark.skipif(torch.cuda.is_available(), reason="launches with host pointers: only meaningful (and safe) without a device")
def test_launching_entry_point_returns_an_error_without_a_device(built):
    """The whole host path of an entry point up to the kernel launch — argument checks, tile choice, the thread-local
    stop-event of LOTUS_LAUNCH — runs on a box without a GPU and must end in LOTUS_E_LAUNCH, not in a crash (round 5: a hidden
    `thread_local` resolved its weak init function to the load address and every launch jumped there)."""
    import ctypes

    import numpy as np
    L = _capi.lib()
    a, w, y = (np.zeros(n, dtype=np.float32) for n in (100 * 64, 32 * 64, 100 * 32))
    rc = L.fn["lotus_linear_fwd"](a.ctypes.data, w.ctypes.data, None, None, y.ctypes.data, None, 100, 32, 64, 0, 0.0, 0, 0, None, 0, None, None)
    assert rc == -2 and b"launch failed" in L.fn["lotus_last_error"]()
    a2, w2, y2 = (np.zeros(n, dtype=np.float32) for n in (20000 * 64, 128 * 64, 20000 * 128))   # the LDS-DMA kernels' host path
    rc = L.fn["lotus_linear_fwd"](a2.ctypes.data, w2.ctypes.data, None, None, y2.ctypes.data, None, 20000, 128, 64, 0, 0.0, 0, 0, None, 0, None, None)
    assert rc == -2

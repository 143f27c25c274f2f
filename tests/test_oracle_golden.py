"""Pin the oracle (oracle/) against the golden fixtures captured from the imported reference."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import front_end as fe
from oracle.model import Oracle
from robot_3dlotus_amd import config as lcfg


@pytest.mark.parametrize("case", gu.CASES)
def test_oracle_matches_reference_fixture(case):
    fx, cfg, batch, sd = gu.load_case(case, gu.state_template)
    train = bool(fx["meta_train"])
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    orc = Oracle(sdg, lcfg.plain(cfg), training=train)
    out = orc.forward(batch, list(fx["perms"]))
    # integer tables: bit-exact
    for s, lv in enumerate(out["levels"]):
        for k in ("grid", "batch", "code", "order", "inverse", "pad", "unpad", "cu_seqlens", "nbr27"):
            np.testing.assert_array_equal(np.asarray(lv[k]).astype(np.int64), fx[f"L{s}_{k}"].astype(np.int64),
                                          err_msg=f"{case} L{s} {k}")
        assert lv["depth"] == int(fx[f"L{s}_depth"])
        if s > 0:
            np.testing.assert_array_equal(lv["cluster"], fx[f"L{s}_cluster"])
    np.testing.assert_array_equal(out["levels"][0]["nbr125"], fx["L0_nbr125"])
    # floats: same arithmetic on the same CPU => tight
    for k in ("xt", "xr", "xo"):
        ref = fx[k]
        tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(out[k].detach().numpy(), ref, atol=tol, rtol=0, err_msg=f"{case} {k}")
    for k, v in out["losses"].items():
        assert abs(v.item() - float(fx["loss_" + k])) < 2e-5 * max(1.0, abs(float(fx["loss_" + k])))
    out["losses"]["total"].backward()
    gmax = max(float(fx[k]) for k in fx if k.startswith("gnorm/"))
    for k in fx:
        if k.startswith("gnorm/"):  # mathematically-zero grads (k_norm.bias) are pure rounding noise
            name = k[6:]
            g = sdg[name].grad
            ref = float(fx[k])
            assert abs(g.double().norm().item() - ref) <= 1e-4 * ref + 1e-6 * gmax, name
            np.testing.assert_allclose(g.flatten()[:48].numpy(), fx["ghead/" + name],
                                       atol=1e-4 * float(np.abs(fx["ghead/" + name]).max()) + 1e-7 * gmax, rtol=0)
    if train:
        for k, v in orc.new_running.items():
            np.testing.assert_allclose(v.numpy(), fx["buf/" + k], atol=2e-3, rtol=2e-3)


def test_hilbert_parent_property():
    """Grid pooling relies on every curve being hierarchical: code >> 3 is a function of the
    parent cell (PointTransformerV3/model.py:726-740)."""
    rng = np.random.default_rng(0)
    g = rng.integers(0, 256, size=(4000, 3)).astype(np.int32)
    b = np.zeros(4000, dtype=np.int64)
    for o in fe.ORDERS:
        child = fe.encode(g, b, 8, o) >> 3
        parent = fe.encode(g >> 1, b, 7, o)
        np.testing.assert_array_equal(child, parent)


def test_padding_tables_known_answer():
    """SURVEY.md §8 a12 known-answer probe: counts [6,3,10], K=4."""
    pad, unpad, cu = fe.padding_tables([6, 3, 10], 4)
    assert pad.tolist() == [0, 1, 2, 3, 4, 5, 2, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 15, 16]
    assert cu.tolist() == [0, 4, 8, 11, 15, 19, 23]
    assert unpad.tolist() == [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20]


@pytest.mark.parametrize("case", gu.MP_CASES)
def test_oracle_mp_matches_reference_fixture(case):
    """Motion-planner restatement (Oracle.forward_mp) against the fixtures of the imported MotionPlannerPTV3CA."""
    fx, cfg, batch, sd = gu.load_case_mp(case)
    train = bool(fx["meta_train"])
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    orc = Oracle(sdg, lcfg.plain(cfg), training=train)
    out = orc.forward_mp(batch, list(fx["perms"]))
    for k in ("xt", "xr", "xo", "xstop"):
        ref = fx[k]
        tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(out[k].detach().numpy(), ref, atol=tol, rtol=0, err_msg=f"{case} {k}")
    assert set(out["losses"]) == {"pos", "rot", "open", "stop", "total"}
    for k, v in out["losses"].items():
        assert abs(v.item() - float(fx["loss_" + k])) < 2e-5 * max(1.0, abs(float(fx["loss_" + k])))
    out["losses"]["total"].backward()
    gmax = max(float(fx[k]) for k in fx if k.startswith("gnorm/"))
    for name in fx["nograd"].tolist():            # txt_attn_fc is built but unused by the CA variant
        assert sdg[name].grad is None
    for k in fx:
        if k.startswith("gnorm/"):
            name = k[6:]
            g = sdg[name].grad
            ref = float(fx[k])
            assert abs(g.double().norm().item() - ref) <= 1e-4 * ref + 1e-6 * gmax, name
            np.testing.assert_allclose(g.flatten()[:48].numpy(), fx["ghead/" + name],
                                       atol=1e-4 * float(np.abs(fx["ghead/" + name]).max()) + 1e-7 * gmax, rtol=0)
    if train:
        for k, v in orc.new_running.items():
            np.testing.assert_allclose(v.numpy(), fx["buf/" + k], atol=2e-3, rtol=2e-3)

"""Pin the CPU oracle (oracle/moe_oracle.py) to the REAL reference's outputs.

The golden .npz files were produced by oracle/make_goldens.py, which imports
/root/reference unmodified.  The reference ships no tests of its
own for this path, so these fixtures are the pin."""
import numpy as np
import pytest
import torch

from oracle import moe_oracle as orc

GATE_CASES = ["rand_10x4", "rand_64x32", "rand_1896x32", "rand_1000x32_sharp", "rand_777x8_cf125",
              "rand_300x32_eval2", "rand_5x32_mincap", "rand_3x2", "rand_130x3", "ties_allequal",
              "ties_allequal_nonoise", "ties_pairwise", "overflow_10x4_cf05", "overflow_200x32",
              "underflow_64x8", "large_128x32"]


def sparse_pairs(slot1, w1, slot2, w2):
    """-> idx [S,2] ascending with -1 pad last, w [S,2] (golden layout)."""
    S = slot1.shape[0]
    idx = np.full((S, 2), -1, dtype=np.int64)
    w = np.zeros((S, 2), dtype=np.float32)
    for s in range(S):
        items = sorted((int(sl), float(ww)) for sl, ww in ((slot1[s], w1[s]), (slot2[s], w2[s])) if sl >= 0)
        for k, (j, ww) in enumerate(items):
            idx[s, k], w[s, k] = j, ww
    return idx, w


def test_noise_restatement_bitexact():
    z = np.load("tests/golden/noise_seed77.npz")
    torch.manual_seed(77)
    assert np.array_equal(orc.gumbel_noise((257, 32)).numpy(), z["noise"])


def test_case_list_complete(gate_cases):
    assert sorted(gate_cases) == sorted(GATE_CASES)


@pytest.mark.parametrize("name", GATE_CASES)
def test_index_gate_matches_reference(gate_cases, name):
    c = gate_cases[name]
    logits, noise = torch.from_numpy(c["logits"]), torch.from_numpy(c["noise"])
    r = orc.index_top2gating(logits, noise, float(c["cf"]), int(c["min_cap"]))
    assert r.C == int(c["C"])
    assert np.array_equal(r.exp_counts.numpy(), c["exp_counts"])
    idx, w = sparse_pairs(r.slot1.numpy(), r.w1.numpy(), r.slot2.numpy(), r.w2.numpy())
    assert np.array_equal(idx, c["idx"]), "routing slots differ from reference"
    assert np.array_equal(w, c["w"]), "combine weights differ from reference (expected bit-equal on CPU)"
    np.testing.assert_allclose(r.l_aux.numpy(), c["l_aux"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("name", [n for n in GATE_CASES if n != "rand_1896x32"])
def test_dense_gate_matches_reference(gate_cases, name):
    c = gate_cases[name]
    logits, noise = torch.from_numpy(c["logits"]), torch.from_numpy(c["noise"])
    l_aux, cw, mask, cnt = orc.dense_top2gating(logits, noise, float(c["cf"]), int(c["min_cap"]))
    assert cw.shape[2] == int(c["C"])
    assert np.array_equal(cnt.numpy(), c["exp_counts"])
    S = logits.shape[0]
    flat = cw.reshape(S, -1)
    dense_ref = np.zeros(flat.shape, dtype=np.float32)
    for s in range(S):
        for k in range(2):
            if c["idx"][s, k] >= 0:
                dense_ref[s, c["idx"][s, k]] = c["w"][s, k]
    assert np.array_equal(flat.numpy(), dense_ref)
    assert torch.equal(mask, cw != 0)
    np.testing.assert_allclose(l_aux.numpy(), c["l_aux"], rtol=1e-6)


def _params(z):
    return {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p/")}


LAYER_KEYS = ["vl", "vl_big", "vision_big", "text_big", "plain_big"]


@pytest.mark.parametrize("key", LAYER_KEYS)
@pytest.mark.parametrize("form", ["dense", "index"])
def test_layer_forward_backward_matches_reference(layer_small, key, form):
    z = layer_small
    p = {k: v.clone().requires_grad_(True) for k, v in _params(z).items()}
    mode = key.split("_")[0]
    n_vis = None if mode == "plain" else int(z["n_vis"])
    if mode == "plain":
        p.pop("vision_embed"), p.pop("text_embed")
        mode = "vl"
    h = torch.from_numpy(z[f"{key}/h"]).requires_grad_(True)
    noise, dout = torch.from_numpy(z[f"{key}/noise"]), torch.from_numpy(z[f"{key}/dout"])
    fwd = orc.dense_moe_forward if form == "dense" else orc.index_moe_forward
    res = fwd(h, p, noise, mode=mode, n_vis=n_vis)
    out, l_aux, cnt = res[0], res[1], res[2]
    assert np.array_equal(cnt.numpy(), z[f"{key}/exp_counts"])
    np.testing.assert_allclose(out.detach().numpy(), z[f"{key}/out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(l_aux.detach().numpy(), z[f"{key}/l_aux"], rtol=1e-6)
    ((out * dout).sum() + float(z["aux_w"]) * l_aux).backward()
    np.testing.assert_allclose(h.grad.numpy(), z[f"{key}/dh"], rtol=1e-4, atol=2e-6)
    for name, gk in (("wg", "dwg"), ("w1", "dw1"), ("b1", "db1"), ("w2", "dw2"), ("b2", "db2"),
                     ("vision_embed", "dvision_embed"), ("text_embed", "dtext_embed")):
        if name not in p:
            continue
        got = p[name].grad.numpy() if p[name].grad is not None else np.zeros_like(z[f"{key}/{gk}"])
        np.testing.assert_allclose(got, z[f"{key}/{gk}"], rtol=1e-4, atol=5e-6, err_msg=name)


def test_eval_capacity_factor(layer_small):
    z = layer_small
    p = _params(z)
    h, noise = torch.from_numpy(z["eval2/h"]), torch.from_numpy(z["eval2/noise"])
    out, l_aux, cnt, r = orc.index_moe_forward(h, p, noise, mode="vl", n_vis=int(z["n_vis"]), capacity_factor=2.0)
    assert np.array_equal(cnt.numpy(), z["eval2/exp_counts"])
    np.testing.assert_allclose(out.numpy(), z["eval2/out"], rtol=1e-5, atol=1e-6)


def test_cfg1_full_shape_index_form(layer_cfg1):
    """EVE-Base config-1 shape (B=8, 197+40 tokens, fp32): index-form oracle vs reference."""
    z = layer_cfg1
    M, I, E, B, n_vis, T = 768, 3072, 32, 8, 197, 237
    p = orc.make_params(M, I, E, seed=int(z["param_seed"]))
    h = orc.synth_hidden(B, T, M, seed=int(z["hidden_seed"]))
    g = torch.Generator().manual_seed(int(z["noise_seed"]))
    noise = orc.gumbel_noise((B * T, E), generator=g)
    with torch.no_grad():
        out, l_aux, cnt, r = orc.index_moe_forward(h, p, noise, mode="vl", n_vis=n_vis)
    assert np.array_equal(cnt.numpy(), z["exp_counts"])
    o = out.reshape(B * T, M)
    np.testing.assert_allclose(o[torch.from_numpy(z["tok_index"])].numpy(), z["out_slice"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(o.double().sum(1).numpy(), z["out_rowsum"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(l_aux.numpy(), z["l_aux"], rtol=1e-6)

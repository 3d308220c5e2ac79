"""CPU: the oracle's restatement of the hard modality split against goldens captured from the
reference's own MultiwayTransformerEnocderLayer (oracle/make_goldens_split.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import moe_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "split_ffn_small.npz")
CASES = ["vl", "vl_long", "vision", "text"]


def load():
    z = np.load(GOLD)
    p = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p/")}
    return z, p


@pytest.mark.parametrize("case", CASES)
def test_oracle_split_ffn_matches_reference_layer(case):
    z, p = load()
    p = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    h = torch.from_numpy(z[f"{case}/h"]).requires_grad_(True)
    out = orc.split_ffn_forward(h, p, int(z["n_text"]), str(z[f"{case}/mode"]))
    np.testing.assert_allclose(out.detach().numpy(), z[f"{case}/out"], rtol=1e-5, atol=1e-6)
    (out * torch.from_numpy(z[f"{case}/dout"])).sum().backward()
    np.testing.assert_allclose(h.grad.numpy(), z[f"{case}/dh"], rtol=1e-4, atol=1e-6)
    for k in ("ln_w", "ln_b", "w1", "b1", "w2", "b2"):
        got = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
        np.testing.assert_allclose(got.numpy(), z[f"{case}/d{k}"], rtol=1e-4, atol=2e-6, err_msg=k)

"""Golden vectors for the hard modality split of the non-MoE layers, from the REAL reference layer.

TEST INFRASTRUCTURE ONLY.  Builds the reference's `MultiwayTransformerEnocderLayer` (imported unmodified from
/root/reference with the timm stub of make_goldens.py) in its split-FFN configuration, replaces its attention by a
module that returns zeros (so the layer's own forward hands the FFN branch the layer input itself) and captures
the FFN branch's output at the layer's own `drop_path` call (`:408`, gamma_2 = 1.0, DropPath = Identity) -- every
number in the fixture is produced by the reference's forward/backward code.  Writes tests/golden/split_ffn_small.npz.

    python oracle/make_goldens_split.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import make_goldens as mg  # noqa: E402
from oracle import moe_oracle as orc  # noqa: E402


class ZeroAttn(torch.nn.Module):
    def forward(self, x, mask=None, relative_position_bias=None):
        return torch.zeros_like(x)


def main():
    sm, layer_mod, experts, mt = mg.import_reference()
    M, I, n_text = 32, 64, 4
    p = orc.make_split_params(M, I, seed=21)
    out = {"M": np.int64(M), "I": np.int64(I), "n_text": np.int64(n_text)}
    for k, v in p.items():
        out["p/" + k] = v.numpy()
    g = torch.Generator().manual_seed(22)
    for name, mode, B, T in (("vl", "vl", 3, 11), ("vl_long", "vl", 2, 45), ("vision", "vision", 2, 7), ("text", "text", 3, 4)):
        lay = mt.MultiwayTransformerEnocderLayer(M, "quick_gelu", 4, I, (T - n_text, n_text), 0.0, layer_scale_init_values=None,
                                                use_ffn=False, ffn_type="none", ffn_param=None, single_ffn={"type": "split"})
        lay.attn = ZeroAttn()
        with torch.no_grad():
            for i, (norm, mlp) in enumerate(((lay.norm2_imag, lay.mlp_imag), (lay.norm2_text, lay.mlp_text))):
                norm.weight.copy_(p["ln_w"][i]); norm.bias.copy_(p["ln_b"][i])
                mlp.fc1.weight.copy_(p["w1"][i]); mlp.fc1.bias.copy_(p["b1"][i])
                mlp.fc2.weight.copy_(p["w2"][i]); mlp.fc2.bias.copy_(p["b2"][i])
        captured = []
        lay.drop_path.register_forward_hook(lambda m, inp, res: captured.append(inp[0]))
        h = torch.randn(B, T, M, generator=g).requires_grad_(True)
        dout = torch.randn(B, T, M, generator=g)
        lay(h, None, None, mode=mode)
        ffn_out = captured[1]  # [0] is the attention branch's drop_path call (:344), [1] the FFN branch's (:408)
        (ffn_out * dout).sum().backward()
        # the oracle restatement must agree with the reference to fp32 rounding
        torch.testing.assert_close(orc.split_ffn_forward(h.detach(), p, n_text, mode), ffn_out.detach(), rtol=1e-6, atol=1e-6)
        z = lambda q: (q.grad if q.grad is not None else torch.zeros_like(q)).numpy()
        out.update({f"{name}/mode": np.array(mode), f"{name}/h": h.detach().numpy(), f"{name}/dout": dout.numpy(),
                    f"{name}/out": ffn_out.detach().numpy(), f"{name}/dh": h.grad.numpy(),
                    f"{name}/dln_w": np.stack([z(lay.norm2_imag.weight), z(lay.norm2_text.weight)]),
                    f"{name}/dln_b": np.stack([z(lay.norm2_imag.bias), z(lay.norm2_text.bias)]),
                    f"{name}/dw1": np.stack([z(lay.mlp_imag.fc1.weight), z(lay.mlp_text.fc1.weight)]),
                    f"{name}/db1": np.stack([z(lay.mlp_imag.fc1.bias), z(lay.mlp_text.fc1.bias)]),
                    f"{name}/dw2": np.stack([z(lay.mlp_imag.fc2.weight), z(lay.mlp_text.fc2.weight)]),
                    f"{name}/db2": np.stack([z(lay.mlp_imag.fc2.bias), z(lay.mlp_text.fc2.bias)])})
        print(name, "done", tuple(ffn_out.shape))
    np.savez_compressed(os.path.join(mg.OUT, "split_ffn_small.npz"), **out)


if __name__ == "__main__":
    main()

"""Golden vectors for the whole multiway encoder layer, from the REAL reference layer.

TEST INFRASTRUCTURE ONLY.  Builds the reference's `MultiwayTransformerEnocderLayer` (imported unmodified from
/root/reference with the timm stub of make_goldens.py) in fp32 on the CPU -- its own attention, LayerNorms, layer
scale, MoE branch and hard-split branch -- runs forward + backward of `sum(out * dout) + aux_w * aux_loss` and
saves the state dict, the inputs (hidden state, key-padding mask, injected Gumbel noise) and every result (output,
input gradient, all parameter gradients by state-dict name).  Writes tests/golden/encoder_layer_small.npz.

    python oracle/make_goldens_encoder.py
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

M, I, HEADS, N_TEXT = 32, 64, 4, 4
CASES = [  # name, use_ffn, ffn_type, ffn_param, single_ffn, mode, B, T, padded (sample, position) pairs
    ("moe_vl", True, "moe", {"num_experts": 8, "topk": 2, "type_route": True}, {"type": "vl"}, "vl", 3, 9, [(1, 8)]),
    ("moe_text", True, "moe", {"num_experts": 8, "topk": 2, "type_route": True}, {"type": "vl"}, "text", 2, 4, [(0, 3)]),
    ("split_vl", False, "moe", None, {"type": "vl"}, "vl", 3, 11, [(2, 10), (2, 9)]),
    ("split_vision", False, "moe", None, {"type": "vl"}, "vision", 2, 7, []),
    ("vlffn_vl", True, "vl", None, {"type": "vl"}, "vl", 2, 9, []),
    ("single_vl", False, "moe", None, {"type": "single"}, "vl", 2, 9, [(1, 8)]),
]


def main():
    sm, layer_mod, experts, mt = mg.import_reference()
    out = {"M": np.int64(M), "I": np.int64(I), "heads": np.int64(HEADS), "n_text": np.int64(N_TEXT),
           "cases": np.array([c[0] for c in CASES])}
    g = torch.Generator().manual_seed(31)
    for name, use_ffn, ffn_type, ffn_param, single_ffn, mode, B, T, pads in CASES:
        n_vis = T - N_TEXT if mode == "vl" else (T if mode == "vision" else 0)
        torch.manual_seed(100 + len(name))
        lay = mt.MultiwayTransformerEnocderLayer(M, "quick_gelu", HEADS, I, (max(n_vis, 1), N_TEXT), 0.0,
                                                layer_scale_init_values=0.1, use_ffn=use_ffn, ffn_type=ffn_type,
                                                ffn_param=ffn_param, single_ffn=single_ffn)
        with torch.no_grad():  # non-trivial values everywhere (the reference initialises biases / scales to constants)
            for p in lay.parameters():
                if p.dim() == 1:
                    p.add_(torch.randn(p.shape, generator=g) * 0.05)
            if hasattr(lay, "mlp_moe"):  # experts are deep copies of one template: make them differ
                for e in lay.mlp_moe.moe.experts.experts:
                    for p in e.parameters():
                        p.add_(torch.randn(p.shape, generator=g) * 0.05)
        h = torch.randn(B, T, M, generator=g).requires_grad_(True)
        dout = torch.randn(B, T, M, generator=g)
        mask = torch.ones(B, T, dtype=torch.long)
        for b, t in pads:
            mask[b, t] = 0
        aux_w = 0.6
        noise = None
        if hasattr(lay, "mlp_moe"):
            E = lay.num_experts
            noise = orc.gumbel_noise((B * T, E), generator=g)
            sm.gumbel_rsample = lambda shape, device, _n=noise: _n.clone()
        lay.train()
        res = lay(h, mask, None, mode=mode)
        loss = (res * dout).sum()
        if lay.aux_loss_vl_i:
            loss = loss + aux_w * lay.aux_loss_vl / lay.aux_loss_vl_i
            out[f"{name}/l_aux"] = (lay.aux_loss_vl / lay.aux_loss_vl_i).detach().numpy()
        loss.backward()
        out.update({f"{name}/h": h.detach().numpy(), f"{name}/dout": dout.numpy(), f"{name}/mask": mask.numpy(),
                    f"{name}/out": res.detach().numpy(), f"{name}/dh": h.grad.numpy(), f"{name}/aux_w": np.float64(aux_w),
                    f"{name}/cfg": np.array([str(use_ffn), ffn_type, repr(ffn_param), repr(single_ffn), mode, str(n_vis)])})
        if noise is not None:
            out[f"{name}/noise"] = noise.numpy()
        for k, v in lay.state_dict().items():
            out[f"{name}/sd/{k}"] = v.numpy()
        for k, p in lay.named_parameters():
            out[f"{name}/grad/{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        print(name, "done", tuple(res.shape), "params", sum(1 for _ in lay.parameters()))
    np.savez_compressed(os.path.join(mg.OUT, "encoder_layer_small.npz"), **out)


if __name__ == "__main__":
    main()

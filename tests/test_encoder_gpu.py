"""Encoder layer on the GPU against the REAL reference layer's fp32 results
(tests/golden/encoder_layer_small.npz): output, input gradient and every parameter gradient."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_fp32_close
from tests.test_encoder_cpu import GOLD, build_layer, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FLOOR = 1e-5


@pytest.mark.parametrize("name", ["moe_vl", "moe_text", "split_vl", "split_vision", "vlffn_vl", "single_vl"])
def test_layer_matches_reference_fp32(name):
    z = np.load(GOLD)
    cfg, mode, n_vis, sd, grads = load_case(z, name)
    lay = build_layer(z, cfg, n_vis)
    lay.load_state_dict(sd, strict=True)
    lay = lay.to(DEV).train()
    h = torch.from_numpy(z[f"{name}/h"]).to(DEV).requires_grad_(True)
    dout = torch.from_numpy(z[f"{name}/dout"]).to(DEV)
    mask = torch.from_numpy(z[f"{name}/mask"]).to(DEV)
    if f"{name}/noise" in z.files:
        lay.mlp_moe.moe.noise_override = torch.from_numpy(z[f"{name}/noise"]).to(DEV)
    out = lay(h, mask, None, mode=mode)
    loss = (out * dout).sum()
    if lay.aux_loss_vl_i:
        l_aux = lay.aux_loss_vl / lay.aux_loss_vl_i
        torch.testing.assert_close(l_aux.detach().cpu(), torch.from_numpy(z[f"{name}/l_aux"]), rtol=1e-5, atol=1e-6)
        loss = loss + float(z[f"{name}/aux_w"]) * l_aux
    loss.backward()
    # the fp32 contract (1e-5 per element, floor 1e-5 of the tensor's largest element) through the whole layer,
    # stock attention included
    assert_fp32_close(out, z[f"{name}/out"], f"{name}: out", floor=FLOOR)
    assert_fp32_close(h.grad, z[f"{name}/dh"], f"{name}: dh", floor=FLOOR)
    for k, p in lay.named_parameters():
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(p).cpu()
        assert_fp32_close(got, grads[k], f"{name}: grad of {k}", floor=FLOOR)


def test_base_layer_bf16_tracks_fp32():
    """EVE-Base dims, one MoE layer and one split layer: the bf16 layer against the same layer run in fp32."""
    from eve_amd.encoder import MultiwayTransformerEnocderLayer
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(6)
    for use_ffn in (True, False):
        lay = MultiwayTransformerEnocderLayer(768, "quick_gelu", 12, 3072, (197, 40), 0.0, use_ffn=use_ffn, ffn_type="moe",
                                              ffn_param={"num_experts": 32, "topk": 2, "type_route": True},
                                              single_ffn={"type": "vl"}).to(DEV)
        with torch.no_grad():
            for p in lay.parameters():
                if p.dim() > 1:
                    torch.nn.init.trunc_normal_(p, std=0.02)
        h = torch.randn(4, 237, 768, generator=g).to(DEV)
        noise = torch.empty(4 * 237, 32, device=DEV).uniform_(1e-6, 1 - 1e-6)
        noise = -torch.log(-torch.log(noise))
        outs = []
        for dt in (torch.float32, torch.bfloat16):
            l2 = lay.to(dt)
            if use_ffn:
                l2.mlp_moe.moe.noise_override = noise
            outs.append(l2(h.to(dt), None, None, mode="vl").float())
            l2.aux_loss_vl, l2.aux_loss_vl_i = 0, 0
        # routing may differ for a handful of near-tie tokens between the two precisions: compare robustly
        err = (outs[0] - outs[1]).detach().abs()
        assert float(err.median()) < 2e-2 and float((err > 0.1).float().mean()) < 0.02

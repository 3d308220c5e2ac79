"""GPU: eve_amd.ModalitySplitFFN (fused split-LayerNorm + two-group grouped GEMM + row permutation) against the
goldens of the reference layer, the CPU oracle, and torch's own LayerNorm."""
import numpy as np
import pytest
import torch

import eve_amd
from eve_amd import ops
from oracle import moe_oracle as orc
from tests.test_split_oracle import CASES, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(p, M, I, n_text, dtype=torch.float32):
    m = eve_amd.ModalitySplitFFN("quick_gelu", M, I, num_text_tokens=n_text)
    with torch.no_grad():
        for i, (norm, mlp) in enumerate(((m.norm2_imag, m.mlp_imag), (m.norm2_text, m.mlp_text))):
            norm.weight.copy_(p["ln_w"][i]); norm.bias.copy_(p["ln_b"][i])
            mlp.fc1.weight.copy_(p["w1"][i]); mlp.fc1.bias.copy_(p["b1"][i])
            mlp.fc2.weight.copy_(p["w2"][i]); mlp.fc2.bias.copy_(p["b2"][i])
    return m.to(dtype).to(DEV)


def grads(m):
    z = lambda q: (q.grad if q.grad is not None else torch.zeros_like(q)).float().cpu()
    st = lambda a, b: torch.stack([z(a), z(b)])
    return {"ln_w": st(m.norm2_imag.weight, m.norm2_text.weight), "ln_b": st(m.norm2_imag.bias, m.norm2_text.bias),
            "w1": st(m.mlp_imag.fc1.weight, m.mlp_text.fc1.weight), "b1": st(m.mlp_imag.fc1.bias, m.mlp_text.fc1.bias),
            "w2": st(m.mlp_imag.fc2.weight, m.mlp_text.fc2.weight), "b2": st(m.mlp_imag.fc2.bias, m.mlp_text.fc2.bias)}


@pytest.mark.parametrize("case", CASES)
def test_split_ffn_matches_reference_golden_fp32(case):
    z, p = load()
    m = build(p, int(z["M"]), int(z["I"]), int(z["n_text"]))
    h = torch.from_numpy(z[f"{case}/h"]).to(DEV).requires_grad_(True)
    out = m(h, mode=str(z[f"{case}/mode"]))
    (out * torch.from_numpy(z[f"{case}/dout"]).to(DEV)).sum().backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.detach().cpu().numpy(), z[f"{case}/out"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(h.grad.cpu().numpy(), z[f"{case}/dh"], rtol=1e-4, atol=5e-6)
    for k, got in grads(m).items():
        np.testing.assert_allclose(got.numpy(), z[f"{case}/d{k}"], rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("mode,B,T", [("vl", 8, 237), ("vision", 3, 197), ("text", 5, 40)])
def test_split_ffn_base_dims_bf16_vs_oracle(mode, B, T):
    """EVE-Base dims (M=768, I=3072, 40 text tokens): the bf16 MFMA path against the fp64 oracle on the same
    bf16-rounded operands; tolerance 2e-2 of each tensor's scale (the brief's bf16 bar)."""
    M, I, n_text = 768, 3072, 40
    p = orc.make_split_params(M, I, seed=5)
    pb = {k: v.to(torch.bfloat16) for k, v in p.items()}
    m = build(pb, M, I, n_text, torch.bfloat16)
    g = torch.Generator().manual_seed(9)
    h = torch.randn(B, T, M, generator=g).to(torch.bfloat16)
    dout = torch.randn(B, T, M, generator=g).to(torch.bfloat16)
    x = h.to(DEV).requires_grad_(True)
    out = m(x, mode=mode)
    (out * dout.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    p64 = {k: v.double().requires_grad_(True) for k, v in pb.items()}
    h64 = h.double().requires_grad_(True)
    ref = orc.split_ffn_forward(h64, p64, n_text, mode)
    (ref * dout.double()).sum().backward()

    def close(got, want, name):
        scale = float(want.abs().max().clamp(min=1e-6))
        err = float((got.double().cpu() - want).abs().max()) / scale
        assert err < 2e-2, (name, err)

    close(out.detach(), ref.detach(), "out")
    close(x.grad, h64.grad, "dh")
    for k, got in grads(m).items():
        want = p64[k].grad if p64[k].grad is not None else torch.zeros_like(p64[k])
        close(got, want, k)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,n_vis,M", [(2, 9, 5, 64), (3, 237, 197, 768), (2, 5, 0, 1024), (1, 6, 6, 256)])
def test_split_layernorm_kernels_vs_torch(dtype, B, T, n_vis, M):
    g = torch.Generator().manual_seed(3)
    h = (torch.randn(B, T, M, generator=g) * 2 + 0.5).to(dtype)
    gam = (1 + 0.3 * torch.randn(2, M, generator=g)).to(dtype)
    bet = (0.2 * torch.randn(2, M, generator=g)).to(dtype)
    R = B * max(n_vis, T - n_vis)
    dxg = torch.randn(2, R, M, generator=g).to(dtype)
    hd, gd, bd = h.to(DEV), gam.to(DEV), bet.to(DEV)
    xg, mean, rstd = ops.split_ln_fwd(hd, gd[0], bd[0], gd[1], bd[1], n_vis, R, 1e-6)
    dh, dp = ops.split_ln_bwd(dxg.to(DEV), hd, mean, rstd, gd[0], gd[1], n_vis)
    back = ops.split_permute(xg, B, T, n_vis, R, to_groups=False)
    again = ops.split_permute(back, B, T, n_vis, R, to_groups=True)
    torch.cuda.synchronize()
    # reference: torch LayerNorm per modality in fp64 on the same (rounded) operands
    h64 = h.double().requires_grad_(True)
    g64, b64 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ln = lambda x, i: torch.nn.functional.layer_norm(x, (M,), g64[i], b64[i], 1e-6)
    want = torch.cat([ln(h64[:, :n_vis], 0), ln(h64[:, n_vis:], 1)], dim=1)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(back.double().cpu(), want.detach(), **tol)
    nv, nt = B * n_vis, B * (T - n_vis)
    assert torch.equal(again[0, :nv], xg[0, :nv]) and torch.equal(again[1, :nt], xg[1, :nt])
    # backward: dxg rows in token order
    dtok = torch.cat([dxg[0, :nv].view(B, n_vis, M), dxg[1, :nt].view(B, T - n_vis, M)], dim=1).double()
    (want * dtok).sum().backward()
    btol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(dh.double().cpu(), h64.grad, **btol)
    scale = lambda t: float(t.abs().max().clamp(min=1.0))
    for i, (name, ref) in enumerate((("dgamma_v", g64.grad[0]), ("dbeta_v", b64.grad[0]), ("dgamma_t", g64.grad[1]),
                                     ("dbeta_t", b64.grad[1]))):
        err = float((dp[i].double().cpu() - ref).abs().max()) / scale(ref)
        assert err < (1e-4 if dtype == torch.float32 else 2e-2), (name, err)


def test_split_ffn_state_dict_names_and_errors():
    m = eve_amd.ModalitySplitFFN("quick_gelu", 64, 128, num_text_tokens=4)
    keys = set(m.state_dict())
    assert {"norm2_imag.weight", "norm2_text.bias", "mlp_imag.fc1.weight", "mlp_text.fc2.bias"} <= keys
    assert len(keys) == 12
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 8, 64))
    m = m.to(DEV)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 8, 64, device=DEV), mode="audio")
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 64, device=DEV))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,M,use_gamma,use_drop", [(3, 7, 64, True, True), (4, 237, 768, True, False),
                                                      (2, 33, 1024, True, True), (5, 9, 72, False, True),
                                                      (2, 5, 4096, True, True)])
def test_scale_residual_matches_the_reference_op_chain(dtype, B, T, M, use_gamma, use_drop):
    """eve_scale_residual_fwd/bwd vs `residual + drop_path(x * gamma)` (multiway_transformer.py:341, :408) in stock
    torch ops: fp32 forward and dx bit for bit (same roundings in the same order), dgamma to summation order;
    bf16 within one rounding."""
    from eve_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(B * T + M)
    x = torch.randn(B, T, M, generator=g).to(dtype).to(dev)
    res = torch.randn(B, T, M, generator=g).to(dtype).to(dev)
    dout = torch.randn(B, T, M, generator=g).to(dtype).to(dev)
    gamma = (0.1 + 0.05 * torch.randn(M, generator=g)).to(dtype).to(dev) if use_gamma else None
    rs = ((torch.rand(B, generator=g) < 0.7).float() / 0.7).to(dtype).to(dev) if use_drop else None
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True) if use_gamma else None
    t = xr * gr if use_gamma else xr
    if use_drop:
        t = t * rs.view(B, 1, 1)
    want = res + t
    want.backward(dout)
    S = B * T
    out = ops.scale_residual_fwd(x.view(S, M), gamma, rs, res.view(S, M), T)
    dx, dgamma = ops.scale_residual_bwd(dout.view(S, M), x.view(S, M), gamma, rs, T)
    torch.cuda.synchronize()
    if dtype == torch.float32:
        assert torch.equal(out.view(B, T, M), want.detach())
        assert torch.equal(dx.view(B, T, M), xr.grad)
    else:
        torch.testing.assert_close(out.view(B, T, M).float(), want.detach().float(), rtol=1e-2, atol=1e-2)
        torch.testing.assert_close(dx.view(B, T, M).float(), xr.grad.float(), rtol=1e-2, atol=1e-2)
    if use_gamma:
        ref64 = ((dout.double() * (rs.double().view(B, 1, 1) if use_drop else 1.0)) * x.double()).sum((0, 1))
        torch.testing.assert_close(dgamma.double(), ref64, rtol=1e-4, atol=1e-4 * float(ref64.abs().max()))
    else:
        assert dgamma is None


def test_encoder_layer_residual_branch_with_drop_path_replays_the_draw():
    """The fused residual branch consumes the generator exactly as DropPath does: same seed -> same masks as the
    stock-op formulation, so outputs agree (bf16 rounding apart) in training mode with drop_path > 0."""
    import eve_amd
    from eve_amd.encoder import MultiwayTransformerEnocderLayer as Layer
    dev = "cuda:0"
    torch.manual_seed(3)
    layer = Layer(64, "quick_gelu", 4, 128, (5, 4), drop_path=0.4, use_ffn=False).to(dev).train()
    x = torch.randn(6, 9, 64, device=dev)
    torch.manual_seed(11)
    got = layer(x, mode="vl")

    def stock_branch(residual, h, gamma):
        return residual + layer.drop_path(h * gamma)

    torch.manual_seed(11)
    layer._branch = stock_branch
    want = layer(x, mode="vl")
    del layer._branch
    torch.cuda.synchronize()
    assert torch.equal(got, want)

"""eve_moe_fwd / eve_moe_bwd (the whole single-rank layer in one C call per direction) against the per-stage entry
points they are composed of: same kernels in the same order, so every output and gradient must be bit-identical.
The golden / oracle parity of the layer itself is in test_layer_gpu.py, which runs through the fused calls (the
default for ep_size == 1)."""
import ctypes

import pytest
import torch

import eve_amd
from eve_amd import _lib, ops
from oracle import moe_oracle as orc
from tests.helpers import build_moe, expert_grads

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(moe, h, dout, noise, mode, fused, train=True, split=False):
    moe.moe.fused_call = fused
    moe.moe.split_call = split
    for p in moe.parameters():
        p.grad = None
    moe.moe.noise_override = noise.clone()
    if not train:
        with torch.no_grad():
            out, l_aux, cnt = moe(h, mode=mode)
        return [out, l_aux, cnt, moe.moe.last_slot.clone()]
    x = h.detach().clone().requires_grad_(True)
    out, l_aux, cnt = moe(x, mode=mode)
    torch.autograd.backward([out, l_aux], [dout, torch.full_like(l_aux, 0.37)])
    torch.cuda.synchronize()
    res = [out.detach(), l_aux.detach(), cnt, moe.moe.last_slot.clone(), x.grad, moe.moe.gate.wg.weight.grad]
    for name in ("vision_embed", "text_embed"):
        g = getattr(moe.moe, name, None)
        res.append(None if g is None or g.grad is None else g.grad.clone())
    res += list(expert_grads(moe, moe.moe.num_local_experts))
    return res


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode,n_vis", [("vl", 5), ("vision", 5), ("text", 5), ("vl", None)])
def test_fused_calls_equal_the_per_stage_calls_bit_for_bit(dtype, mode, n_vis):
    M, I, E, B, T = 64, 128, 8, 6, 13
    p = orc.make_params(M, I, E, seed=3)
    moe = build_moe(p, M, I, E, n_vis, device=DEV, dtype=dtype)
    g = torch.Generator().manual_seed(11)
    h = torch.randn(B, T, M, generator=g).to(dtype).to(DEV)
    dout = torch.randn(B, T, M, generator=g).to(dtype).to(DEV)
    noise = orc.gumbel_noise((B * T, E), generator=g).to(DEV)
    a = _run(moe, h, dout, noise, mode, fused=True)
    b = _run(moe, h, dout, noise, mode, fused=False)
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert (x is None) == (y is None), i
        if x is not None:
            assert x.dtype == y.dtype and torch.equal(x, y), f"result {i} differs"
    # the expert-parallel halves (eve_moe_route_* / eve_moe_finish_* around eve_ffn_*), here with the identity exchange
    c = _run(moe, h, dout, noise, mode, fused=True, split=True)
    for i, (x, y) in enumerate(zip(c, b)):
        assert (x is None) == (y is None), i
        if x is not None:
            assert x.dtype == y.dtype and torch.equal(x, y), f"halves: result {i} differs"
    ai = _run(moe, h, dout, noise, mode, fused=True, train=False)
    bi = _run(moe, h, dout, noise, mode, fused=False, train=False)
    ci = _run(moe, h, dout, noise, mode, fused=True, train=False, split=True)
    for i, (x, y, z) in enumerate(zip(ai, bi, ci)):
        assert torch.equal(x, y) and torch.equal(z, y), f"inference result {i} differs"
    assert torch.equal(ai[0], a[0])  # inference (activations in the workspace, z not stored) == training forward


def test_fused_calls_at_model_dims_bf16_equal_the_per_stage_calls():
    """EVE-Base dims: the MFMA GEMM path, split partial sums and all, through one workspace."""
    M, I, E, B, n_vis, T = 768, 3072, 32, 8, 197, 237
    moe = eve_amd.MoE(M, eve_amd.MultiwayTransformerMLP("quick_gelu", M, I), num_experts=E, k=2, num_tokens=n_vis)
    moe = moe.to(torch.bfloat16).to(DEV)
    g = torch.Generator().manual_seed(5)
    h = torch.nn.functional.layer_norm(torch.randn(B, T, M, generator=g), (M,)).to(torch.bfloat16).to(DEV)
    dout = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(DEV)
    noise = orc.gumbel_noise((B * T, E), generator=g).to(DEV)
    a = _run(moe, h, dout, noise, "vl", fused=True)
    b = _run(moe, h, dout, noise, "vl", fused=False)
    c = _run(moe, h, dout, noise, "vl", fused=True, split=True)
    for i, (x, y, z) in enumerate(zip(a, b, c)):
        assert (x is None) == (y is None) == (z is None), i
        if x is not None:
            assert torch.equal(x, y), f"result {i} differs"
            assert torch.equal(z, y), f"halves: result {i} differs"


def test_fused_entry_points_reject_small_buffers_and_bad_shapes():
    lib = _lib.load()
    S, T, M, I, E = 64, 8, 64, 128, 4
    C = ops.capacity(S, E, 1.0, 4)
    shape = _lib.MoEShape(S, T, 3, _lib.MODE_VL, M, I, E, C, _lib.ACT_QUICK_GELU, _lib.BF16, _lib.BF16)
    ref = ctypes.addressof(shape)
    nsaved = lib.eve_moe_saved_bytes(ref, 1)
    nws = lib.eve_moe_workspace_bytes(ref, _lib.MOE_PASS_FORWARD)
    assert nsaved > 0 and nws > 0
    bf = dict(dtype=torch.bfloat16, device=DEV)
    h, ve, te = torch.zeros(S, M, **bf), torch.zeros(M, **bf), torch.zeros(M, **bf)
    wg = torch.zeros(E, M, device=DEV)
    noise = torch.zeros(S, E, device=DEV)
    w1, b1, w2, b2 = torch.zeros(E, I, M, **bf), torch.zeros(E, I, **bf), torch.zeros(E, M, I, **bf), torch.zeros(E, M, **bf)
    out, l_aux = torch.empty(S, M, **bf), torch.empty(1, device=DEV)
    saved = torch.empty(nsaved, dtype=torch.uint8, device=DEV)
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    p = lambda t: t.data_ptr()

    def fwd(nsv, nw, shp=ref):
        return lib.eve_moe_fwd(shp, p(h), p(ve), p(te), p(wg), p(noise), 0, p(w1), p(b1), p(w2), p(b2), p(out), p(l_aux),
                               None, p(saved), nsv, 1, p(ws), nw, None)

    assert fwd(nsaved, nws) == 0
    torch.cuda.synchronize()
    assert fwd(nsaved - 1, nws) == -2 and b"saved" in lib.eve_moe_last_error()
    assert fwd(nsaved, nws - 1) == -2 and b"workspace" in lib.eve_moe_last_error()
    bad = _lib.MoEShape(S, 7, 3, _lib.MODE_VL, M, I, E, C, _lib.ACT_QUICK_GELU, _lib.BF16, _lib.BF16)  # S % T != 0
    assert fwd(nsaved, nws, ctypes.addressof(bad)) == -1
    big = _lib.MoEShape(S, T, 3, _lib.MODE_VL, M, I, 64, C, _lib.ACT_QUICK_GELU, _lib.BF16, _lib.BF16)
    assert fwd(1 << 30, 1 << 30, ctypes.addressof(big)) == -4
    assert lib.eve_moe_workspace_bytes(ref, 7) == 0


def test_parameter_casts_follow_in_place_updates():
    """fp32 modality embeddings with bf16 activations: the fused path casts them per call; an optimizer-style in-place
    update -- through the parameter (bumps its version counter) or through `.data` (bumps nothing and keeps the storage
    pointer: older transformers.AdamW, fairscale OSS) -- must be seen by the next forward: same results as the
    per-stage path."""
    M, I, E, B, T, n_vis = 64, 128, 8, 4, 11, 6
    p = orc.make_params(M, I, E, seed=5)
    moe = build_moe(p, M, I, E, n_vis, device=DEV)
    moe.moe.experts.to(torch.bfloat16)  # experts bf16, gate + embeddings stay fp32 parameters
    g = torch.Generator().manual_seed(2)
    h = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(DEV)
    dout = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(DEV)
    noise = orc.gumbel_noise((B * T, E), generator=g).to(DEV)
    for step in range(5):
        a = _run(moe, h, dout, noise, "vl", fused=True)
        b = _run(moe, h, dout, noise, "vl", fused=False)
        for i, (x, y) in enumerate(zip(a, b)):
            assert (x is None) == (y is None), (step, i)
            if x is not None:
                assert x.dtype == y.dtype and torch.equal(x, y), f"step {step}: result {i} differs"
        if step % 2 == 0:
            with torch.no_grad():  # in place on the parameter: bumps the version counter
                moe.moe.vision_embed.add_(0.05)
                moe.moe.text_embed.mul_(1.5)
                moe.moe.gate.wg.weight.add_(0.01 * torch.randn_like(moe.moe.gate.wg.weight))
        else:  # through .data: no version bump, same storage
            ver = moe.moe.vision_embed._version
            moe.moe.vision_embed.data.add_(-0.03)
            moe.moe.text_embed.data.mul_(0.7)
            moe.moe.gate.wg.weight.data.add_(0.01)
            assert moe.moe.vision_embed._version == ver


def test_exp_counts_written_to_pinned_host_memory_by_the_gate():
    """eve_moe_fwd's exp_counts_host (ABI 11): with non-blocking exp_counts the layer hands back the pinned tensor the
    gate's closing kernel filled -- no copy on the stream -- equal to the blocking layer's counts."""
    from oracle import moe_oracle as orc
    from tests.helpers import build_moe
    M, I, E, B, n_vis, T = 64, 128, 32, 8, 9, 15
    p = orc.make_params(M, I, E, seed=71)
    h = orc.synth_hidden(B, T, M, seed=72).to(DEV)
    noise = orc.gumbel_noise((B * T, E), generator=torch.Generator().manual_seed(73))
    a, b = build_moe(p, M, I, E, n_vis, device=DEV), build_moe(p, M, I, E, n_vis, device=DEV)
    b.moe.blocking_exp_counts = False
    a.moe.noise_override, b.moe.noise_override = noise.to(DEV), noise.to(DEV)
    _, _, ca = a(h, mode="vl")
    _, _, cb = b(h, mode="vl")
    assert cb.is_pinned() and cb.device.type == "cpu" and cb.dtype == torch.int64
    torch.cuda.synchronize()
    assert torch.equal(ca, cb) and int(cb.sum()) == B * T
    # a second call must not overwrite the tensor handed out by the first (eager: a fresh pinned tensor per call)
    b.moe.noise_override = orc.gumbel_noise((B * T, E), generator=torch.Generator().manual_seed(74)).to(DEV)
    _, _, cb2 = b(h, mode="vl")
    torch.cuda.synchronize()
    assert cb2.data_ptr() != cb.data_ptr() and torch.equal(ca, cb)

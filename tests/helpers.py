"""Shared test helpers (load oracle params into the eve_amd modules, compare tensors)."""
import numpy as np
import torch

import eve_amd


def build_moe(p, M, I, E, n_vis, device=None, dtype=None, **kw):
    moe = eve_amd.MoE(M, eve_amd.MultiwayTransformerMLP("quick_gelu", M, I), num_experts=E, k=2, num_tokens=n_vis, **kw)
    sd = {"moe.gate.wg.weight": p["wg"]}
    if n_vis:
        sd["moe.vision_embed"], sd["moe.text_embed"] = p["vision_embed"], p["text_embed"]
    for e in range(E):
        sd[f"moe.experts.experts.{e}.fc1.weight"] = p["w1"][e]
        sd[f"moe.experts.experts.{e}.fc1.bias"] = p["b1"][e]
        sd[f"moe.experts.experts.{e}.fc2.weight"] = p["w2"][e]
        sd[f"moe.experts.experts.{e}.fc2.bias"] = p["b2"][e]
    missing, unexpected = moe.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    if dtype is not None:
        moe = moe.to(dtype)
    if device is not None:
        moe = moe.to(device)
    return moe


def expert_grads(moe, E):
    ex = moe.moe.experts.experts
    z = lambda p: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().float().cpu()
    return (torch.stack([z(ex[e].fc1.weight) for e in range(E)]), torch.stack([z(ex[e].fc1.bias) for e in range(E)]),
            torch.stack([z(ex[e].fc2.weight) for e in range(E)]), torch.stack([z(ex[e].fc2.bias) for e in range(E)]))


def sparse_pairs(slot, w):
    """slot [S,2], w [S,2] -> golden layout: idx ascending with -1 pad last, weights alongside."""
    slot = np.asarray(slot).astype(np.int64)
    w = np.asarray(w).astype(np.float32)
    S = slot.shape[0]
    idx = np.full((S, 2), -1, dtype=np.int64)
    ww = np.zeros((S, 2), dtype=np.float32)
    for s in range(S):
        items = sorted((int(slot[s, k]), float(w[s, k])) for k in range(2) if slot[s, k] >= 0)
        for k, (j, x) in enumerate(items):
            idx[s, k], ww[s, k] = j, x
    return idx, ww


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def assert_close_scaled(got, ref, rtol, atol_frac, name=""):
    """Per-element |got - ref| <= atol + rtol * |ref| with atol = atol_frac * rms(ref): the tolerance the
    BASELINE states (rtol 2e-2 for bf16, 1e-5 for fp32) on every element, plus an absolute floor tied to the
    tensor's own scale for the elements near zero (a norm-wise bound alone leaves small elements unchecked)."""
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, (name, tuple(got.shape), tuple(ref.shape))
    rms = float(ref.pow(2).mean().sqrt())
    atol = atol_frac * max(rms, 1e-30)
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    if bool(bad.any()):
        i = int(torch.argmax(err - rtol * ref.abs()))
        raise AssertionError(f"{name}: {int(bad.sum())} of {bad.numel()} elements outside rtol {rtol} + atol {atol:.3g} "
                             f"(rms {rms:.3g}); worst: got {float(got.flatten()[i]):.6g} ref {float(ref.flatten()[i]):.6g}")

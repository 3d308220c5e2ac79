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


WORST = {}  # name -> largest err / (atol + rtol * |ref|) seen by the scaled checks of this process (1.0 = at the bar)


def assert_close_scaled(got, ref, rtol, atol_frac, name=""):
    """Per-element |got - ref| <= atol + rtol * |ref| with atol = atol_frac * rms(ref): the tolerance the
    BASELINE states (rtol 2e-2 for bf16, 1e-5 for fp32) on every element, plus an absolute floor tied to the
    tensor's own scale for the elements near zero (a norm-wise bound alone leaves small elements unchecked).
    Returns the measured worst case as a fraction of the bound (also kept in WORST[name] and named in the
    failure message), so that a test can report how much of the floor it uses."""
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, (name, tuple(got.shape), tuple(ref.shape))
    rms = float(ref.pow(2).mean().sqrt())
    atol = atol_frac * max(rms, 1e-30)
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    used = float((err / bound).max()) if err.numel() else 0.0
    WORST[name] = max(WORST.get(name, 0.0), used)
    bad = err > bound
    if bool(bad.any()):
        i = int(torch.argmax(err - rtol * ref.abs()))
        raise AssertionError(f"{name}: {int(bad.sum())} of {bad.numel()} elements outside rtol {rtol} + atol {atol:.3g} "
                             f"(rms {rms:.3g}); worst: got {float(got.flatten()[i]):.6g} ref {float(ref.flatten()[i]):.6g}; "
                             f"measured worst case = {used:.2f} x the bound, largest |err| = {float(err.max()):.3g} "
                             f"= {float(err.max()) / max(rms, 1e-30):.3g} x rms")
    return used


def assert_fp32_close(got, ref, name="", rtol=1e-5, floor=1e-5):
    """The fp32 contract of BASELINE.json (1e-5) on every element: |got - ref| <= rtol * |ref| + floor * max|ref|.
    The floor is tied to the tensor's largest element: a K-term fp32 dot product summed in another order than the
    reference's CPU GEMM differs by about that much wherever its terms cancel."""
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, (name, tuple(got.shape), tuple(ref.shape))
    scale = float(ref.abs().max()) if ref.numel() else 0.0
    err = (got - ref).abs()
    bound = rtol * ref.abs() + floor * max(scale, 1e-30)
    used = float((err / bound).max()) if err.numel() else 0.0
    WORST[name] = max(WORST.get(name, 0.0), used)
    if used > 1.0:
        i = int(torch.argmax(err / bound))
        raise AssertionError(f"{name}: fp32 bar (rtol {rtol} + {floor} x max|ref| = {floor * scale:.3g}) exceeded "
                             f"{used:.2f}-fold; worst: got {float(got.flatten()[i]):.8g} ref {float(ref.flatten()[i]):.8g}")
    return used


# ---------------------------------------------------------------------------------------------------------------
# Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) in numpy: the
# published algorithm the gate's device noise source (include/eve_moe.h, EVE_NOISE_DEVICE_PHILOX) is specified by.
# Pinned by the Random123 known-answer vectors in tests/test_host_cpu.py.
def philox4x32_10(counter, key):
    """counter [..., 4] uint32, key [..., 2] uint32 -> [..., 4] uint32."""
    import numpy as np
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k = [np.asarray(key[..., i], dtype=np.uint64) for i in range(2)]
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k[0]) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c[3] ^ k[1]) & m32, p0 & m32]
        k = [(k[0] + np.uint64(0x9E3779B9)) & m32, (k[1] + np.uint64(0xBB67AE85)) & m32]
    return np.stack(c, axis=-1).astype(np.uint32)


def device_noise_uniforms(seed: int, offset: int, S: int, E: int):
    """The [S,E] standard-uniform draws the gate makes from generator state {seed, offset} (header: element i is word
    i & 3 of the block at counter offset + (i >> 2), mapped to [0,1) as (word >> 8) * 2^-24), fp32."""
    import numpy as np
    n4 = (S * E + 3) // 4
    ctr = (np.arange(n4, dtype=np.uint64) + np.uint64(offset % (1 << 64)))
    counter = np.zeros((n4, 4), np.uint32)
    counter[:, 0] = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    counter[:, 1] = (ctr >> np.uint64(32)).astype(np.uint32)
    seed %= 1 << 64
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32), (n4, 2))
    words = philox4x32_10(counter, key).reshape(-1)[:S * E]
    return ((words >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(S, E)

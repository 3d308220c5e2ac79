"""CPU oracle for EVE's modality-aware MoE layer.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (PyTorch CPU / fp32 or fp64) of the
algorithm the reference executes on the hot path.  It exists to CHECK the HIP
product path and to be timed as the `cpu_baseline` leg of bench.py.  Nothing under
`eve_amd/` may import it; only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s cpu_baseline leg do.

Reference lines restated (paths relative to /root/reference):
  * capacity                         models/sharded_moe.py:150-160, :177
  * Gumbel noise                     models/sharded_moe.py:70-77
  * top-2 gating (dense, one-hot)    models/sharded_moe.py:172-235
  * gate module (fp32 wg, logits)    models/sharded_moe.py:283-297
  * modality routing + dispatch /
    experts / combine                models/sharded_moe.py:345-381
  * expert container                 models/experts.py:18-28
  * expert FFN (fc1, quick_gelu, fc2) models/multiway_transformer.py:44-55
  * MoE wrapper return contract      models/layer.py:77-102

Parity pinning: the reference holds no tests or golden vectors for this path
(SURVEY.md §4, §8c).  The oracle is pinned against the reference ITSELF, imported
in the build container by oracle/make_goldens.py, whose outputs are committed as
tests/golden/*.npz and re-checked by tests/test_oracle_golden.py.

Two formulations are given:
  dense_*   the one-hot [S,E,C] + einsum formulation, i.e. what the reference runs
            (this is what bench.py times as the CPU baseline, kind="port");
  index_*   an index-form (token -> slot) formulation, O(S*M) memory, used to check
            the HIP path at sizes where the dense form needs tens of GB.
Both are differentiable torch code so gradients come from autograd.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch

FP32_EPS = 1.1920928955078125e-07  # torch.finfo(torch.float32).eps
FP32_TINY = 1.1754943508222875e-38  # torch.finfo(torch.float32).tiny


# --------------------------------------------------------------------------- capacity
def capacity(num_tokens: int, num_experts: int, capacity_factor: float, min_capacity: int) -> int:
    """sharded_moe.py:150-160 called as in :177 (factor is already `cf * 2`).

    The reference multiplies the python float S/E by a 0-dim fp32 tensor, so the
    product and the ceil happen in fp32.
    """
    per_expert = torch.tensor(num_tokens / num_experts, dtype=torch.float32)
    factor = torch.tensor(capacity_factor * 2, dtype=torch.float32)
    cap = int(torch.ceil(per_expert * factor).item())
    return max(cap, int(min_capacity))


# ------------------------------------------------------------------------------ noise
def gumbel_noise(shape, generator: Optional[torch.Generator] = None, device="cpu") -> torch.Tensor:
    """sharded_moe.py:70-77: Gumbel(0,1).rsample(shape) in fp32.

    torch.distributions.Gumbel is Uniform(tiny, 1-eps) pushed through
    u -> -log(-log(u)); one torch.rand draw of `shape` per call.
    """
    u = torch.rand(shape, generator=generator, dtype=torch.float32, device=device)
    lo = torch.tensor(FP32_TINY, dtype=torch.float32, device=device)
    hi = torch.tensor(1.0 - FP32_EPS, dtype=torch.float32, device=device)
    u = lo + u * (hi - lo)
    return -torch.log(-torch.log(u))


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """transformers ACT2FN['quick_gelu'] (configs/config_vlmoB_base64k.json:4)."""
    return x * torch.sigmoid(1.702 * x)


# ------------------------------------------------------------------- dense gate (port)
def dense_top2gating(logits: torch.Tensor, noise: torch.Tensor, capacity_factor: float, min_capacity: int):
    """One-hot formulation of top-2 gating, sharded_moe.py:172-235.

    Returns (l_aux, combine_weights [S,E,C] fp32, dispatch_mask [S,E,C] bool,
    exp_counts [E] int64).  `noise` replaces the Gumbel draw of :186.
    """
    S, E = logits.shape
    probs = torch.softmax(logits, dim=1)
    C = capacity(S, E, capacity_factor, min_capacity)

    top1 = probs.argmax(dim=1)
    hot1 = torch.nn.functional.one_hot(top1, E)                      # int64 [S,E]
    noisy = (logits + noise).masked_fill(hot1.bool(), float("-inf"))
    top2 = noisy.argmax(dim=1)
    hot2 = torch.nn.functional.one_hot(top2, E)

    first_total = hot1.sum(dim=0, keepdim=True)
    pos1 = hot1.cumsum(dim=0) - 1
    pos2 = hot2.cumsum(dim=0) - 1 + first_total                      # 2nd choices queue behind all 1st

    exp_counts = first_total.reshape(E).detach().clone()

    frac_prob = probs.mean(dim=0)
    frac_tok = hot1.float().mean(dim=0)
    l_aux = (frac_prob * frac_tok).mean() * E * E

    hot1 = hot1 * (pos1 < C)
    hot2 = hot2 * (pos2 < C)
    slot1 = (pos1 * hot1).sum(dim=1)
    slot2 = (pos2 * hot2).sum(dim=1)

    hot1f, hot2f = hot1.float(), hot2.float()
    g1 = (probs * hot1f).sum(dim=1)
    g2 = (probs * hot2f).sum(dim=1)
    den = (g1 + g2).clamp(min=FP32_EPS)
    g1, g2 = g1 / den, g2 / den

    oh1 = torch.nn.functional.one_hot(slot1, C).float()
    oh2 = torch.nn.functional.one_hot(slot2, C).float()
    combine = (g1[:, None] * hot1f)[:, :, None] * oh1[:, None, :] \
        + (g2[:, None] * hot2f)[:, :, None] * oh2[:, None, :]
    return l_aux, combine, combine.bool(), exp_counts


# ------------------------------------------------------------------- index-form gate
@dataclass
class Routing:
    C: int
    gates: torch.Tensor      # [S,E] fp32 softmax
    i1: torch.Tensor         # [S] int64
    i2: torch.Tensor
    loc1: torch.Tensor       # [S] int64 position in expert queue (before capacity test)
    loc2: torch.Tensor
    keep1: torch.Tensor      # [S] bool  loc < C
    keep2: torch.Tensor
    w1: torch.Tensor         # [S] fp32 renormalised combine weights (0 where dropped)
    w2: torch.Tensor
    slot1: torch.Tensor      # [S] int64 e*C+loc where dispatched (weight != 0), else -1
    slot2: torch.Tensor
    l_aux: torch.Tensor      # scalar fp32
    exp_counts: torch.Tensor  # [E] int64


def index_top2gating(logits: torch.Tensor, noise: torch.Tensor, capacity_factor: float, min_capacity: int) -> Routing:
    """Index form of sharded_moe.py:172-235 (SURVEY.md §8 A5 spec).  Differentiable in
    `logits` through gates -> (w1, w2, l_aux); all indices are constants."""
    S, E = logits.shape
    gates = torch.softmax(logits, dim=1)
    C = capacity(S, E, capacity_factor, min_capacity)
    i1 = gates.argmax(dim=1)
    masked = (logits + noise).detach().clone()
    masked[torch.arange(S), i1] = float("-inf")
    i2 = masked.argmax(dim=1)

    cnt1 = torch.bincount(i1, minlength=E)
    # rank of each token among earlier tokens with the same choice (row-major token order)
    def ranks(idx):
        order = torch.argsort(idx, stable=True)
        sorted_idx = idx[order]
        start = torch.zeros(E + 1, dtype=torch.int64)
        start[1:] = torch.bincount(idx, minlength=E).cumsum(0)
        r_sorted = torch.arange(S) - start[sorted_idx]
        r = torch.empty(S, dtype=torch.int64)
        r[order] = r_sorted
        return r

    loc1 = ranks(i1)
    loc2 = ranks(i2) + cnt1[i2]
    keep1, keep2 = loc1 < C, loc2 < C

    me = gates.mean(dim=0)
    ce = cnt1.float() / S
    l_aux = (me * ce).mean() * E * E

    tok = torch.arange(S)
    g1 = gates[tok, i1] * keep1
    g2 = gates[tok, i2] * keep2
    den = (g1 + g2).clamp(min=FP32_EPS)
    w1, w2 = g1 / den, g2 / den
    slot1 = torch.where(keep1 & (w1 != 0), i1 * C + loc1, torch.full_like(i1, -1))
    slot2 = torch.where(keep2 & (w2 != 0), i2 * C + loc2, torch.full_like(i2, -1))
    return Routing(C, gates, i1, i2, loc1, loc2, keep1, keep2, w1, w2, slot1, slot2, l_aux, cnt1)


def routing_from_dense(combine: torch.Tensor):
    """Decode a dense combine_weights [S,E,C] into per-token (slot, weight) pairs sorted
    by slot id; used to compare dense outputs (ours or the reference's) with index form."""
    S, E, C = combine.shape
    flat = combine.reshape(S, E * C)
    out = []
    for s in range(S):
        nz = torch.nonzero(flat[s]).flatten().tolist()
        out.append([(j, float(flat[s, j])) for j in nz])
    return out


# ------------------------------------------------------------------------ parameters
def make_params(M: int, I: int, E: int, *, seed: int = 0, dtype=torch.float32, modality: bool = True,
                distinct_experts: bool = True) -> Dict[str, torch.Tensor]:
    """Random-init parameters shaped like one reference MoE layer:
    trunc-normal(std .02, +-.02) weights, zero biases unless `distinct_experts`."""
    g = torch.Generator().manual_seed(seed)

    def tn(*shape):
        t = torch.empty(*shape, dtype=torch.float32)
        torch.nn.init.trunc_normal_(t, mean=0.0, std=0.02, a=-0.02, b=0.02, generator=g)
        return t

    p = {"wg": tn(E, M)}
    if distinct_experts:
        p["w1"] = tn(E, I, M)
        p["b1"] = tn(E, I)
        p["w2"] = tn(E, M, I)
        p["b2"] = tn(E, M)
    else:
        p["w1"] = tn(1, I, M).repeat(E, 1, 1)
        p["b1"] = torch.zeros(E, I)
        p["w2"] = tn(1, M, I).repeat(E, 1, 1)
        p["b2"] = torch.zeros(E, M)
    if modality:
        p["vision_embed"] = tn(M)
        p["text_embed"] = tn(M)
    return {k: v.to(dtype) for k, v in p.items()}


def modality_input(h: torch.Tensor, p: Dict[str, torch.Tensor], mode: str, n_vis: Optional[int]) -> torch.Tensor:
    """sharded_moe.py:350-364: what is gated AND dispatched is h plus a per-position
    modality embedding (vision for [:, :n_vis], text after) -- or h itself."""
    M = h.shape[-1]
    if not n_vis or "vision_embed" not in p:
        return h.reshape(-1, M)
    if mode == "vision":
        x = h + p["vision_embed"]
    elif mode == "text":
        x = h + p["text_embed"]
    else:
        x = torch.cat([h[:, :n_vis] + p["vision_embed"], h[:, n_vis:] + p["text_embed"]], dim=1)
    return x.to(h.dtype).reshape(-1, M)


def expert_ffn(x: torch.Tensor, w1, b1, w2, b2) -> torch.Tensor:
    """multiway_transformer.py:51-55 for one expert: fc2(quick_gelu(fc1(x)))."""
    return torch.nn.functional.linear(quick_gelu(torch.nn.functional.linear(x, w1, b1)), w2, b2)


# -------------------------------------------------------------------- dense layer (port)
def dense_moe_forward(h: torch.Tensor, p: Dict[str, torch.Tensor], noise: torch.Tensor, *, mode: str = "vl",
                      n_vis: Optional[int] = None, capacity_factor: float = 1.0, min_capacity: int = 4,
                      jitter: Optional[torch.Tensor] = None):
    """The layer as the reference executes it (one-hot einsums + per-expert loop):
    sharded_moe.py:345-381 + experts.py:18-28.  Returns (out, l_aux, exp_counts).
    `jitter` [S,M]: the factors of multiplicative_jitter (sharded_moe.py:46-67) under noisy_gate_policy='Jitter' in
    training: the gate's Linear sees x.float() * jitter (:290-292); what is dispatched stays x."""
    x = modality_input(h, p, mode, n_vis)
    gate_in = x.float() if jitter is None else x.float() * jitter
    logits = gate_in @ p["wg"].float().t()
    l_aux, combine, mask, exp_counts = dense_top2gating(logits, noise, capacity_factor, min_capacity)
    E = p["wg"].shape[0]
    slots = torch.einsum("sec,sm->ecm", mask.to(h.dtype), x)
    outs = [expert_ffn(slots[e], p["w1"][e], p["b1"][e], p["w2"][e], p["b2"][e]) for e in range(E)]
    y = torch.stack(outs, dim=0)
    out = torch.einsum("sec,ecm->sm", combine.to(h.dtype), y)
    return out.reshape(h.shape), l_aux, exp_counts


# -------------------------------------------------------------------- index-form layer
def index_moe_forward(h: torch.Tensor, p: Dict[str, torch.Tensor], noise: torch.Tensor, *, mode: str = "vl",
                      n_vis: Optional[int] = None, capacity_factor: float = 1.0, min_capacity: int = 4,
                      logits_override: Optional[torch.Tensor] = None):
    """Same function as dense_moe_forward without the [S,E,C] tensors.  Empty capacity
    slots are skipped (their combine weight is 0).  Returns
    (out, l_aux, exp_counts, Routing)."""
    x = modality_input(h, p, mode, n_vis)
    S, M = x.shape
    logits = x.float() @ p["wg"].float().t() if logits_override is None else logits_override
    r = index_top2gating(logits, noise, capacity_factor, min_capacity)
    E = p["wg"].shape[0]
    out = torch.zeros(S, M, dtype=torch.float32)
    for slot, w in ((r.slot1, r.w1), (r.slot2, r.w2)):
        sel = torch.nonzero(slot >= 0).flatten()
        if sel.numel() == 0:
            continue
        e_of = torch.div(slot[sel], r.C, rounding_mode="floor")
        for e in e_of.unique().tolist():
            rows = sel[e_of == e]
            y = expert_ffn(x[rows], p["w1"][e], p["b1"][e], p["w2"][e], p["b2"][e])
            out = out.index_add(0, rows, (w[rows].to(h.dtype)[:, None] * y).float())
    return out.to(h.dtype).reshape(h.shape), r.l_aux, r.exp_counts, r


def synth_hidden(B: int, T: int, M: int, seed: int, dtype=torch.float32) -> torch.Tensor:
    """SURVEY.md §8d synthetic input: LayerNorm(randn) ~ N(0,1) per element."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, M, generator=g)
    return torch.nn.functional.layer_norm(x, (M,)).to(dtype)


# ------------------------------------------------ hard modality split of the non-MoE layers
def make_split_params(M: int, I: int, *, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """norm2_imag / norm2_text affine (perturbed away from the 1/0 init so that they matter) and the two MLPs,
    index 0 = imag, 1 = text."""
    g = torch.Generator().manual_seed(seed)
    p = {
        "ln_w": 1.0 + 0.2 * torch.randn(2, M, generator=g), "ln_b": 0.1 * torch.randn(2, M, generator=g),
        "w1": torch.randn(2, I, M, generator=g) * M ** -0.5, "b1": 0.1 * torch.randn(2, I, generator=g),
        "w2": torch.randn(2, M, I, generator=g) * I ** -0.5, "b2": 0.1 * torch.randn(2, M, generator=g),
    }
    return {k: v.to(dtype) for k, v in p.items()}


def split_ffn_forward(h: torch.Tensor, p: Dict[str, torch.Tensor], num_text_tokens: int, mode: str = "vl",
                      eps: float = 1e-6) -> torch.Tensor:
    """models/multiway_transformer.py:366-370 ('vl'), :386-387 ('vision'), :402-403 ('text')."""
    M = h.shape[-1]

    def branch(x, i):
        x = torch.nn.functional.layer_norm(x, (M,), p["ln_w"][i], p["ln_b"][i], eps)
        return expert_ffn(x, p["w1"][i], p["b1"][i], p["w2"][i], p["b2"][i])

    if mode == "vision":
        return branch(h, 0)
    if mode == "text":
        return branch(h, 1)
    hidden_vision = h[:, :-num_text_tokens]
    hidden_lang = h[:, -num_text_tokens:]
    return torch.cat([branch(hidden_vision, 0), branch(hidden_lang, 1)], dim=1)

"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (the reference is not
present on the GPU box).  The reference is imported unmodified; the only
accommodations are
  * a stub `timm` package (timm is not installed; the reference uses only
    trunc_normal_ / DropPath / registry helpers from it), and
  * replacing `models.sharded_moe.gumbel_rsample` by a function that returns an
    injected tensor, so (logits, noise) -> routing is reproducible.
No reference source is copied: the .npz files hold inputs and outputs only.

    python oracle/make_goldens.py            # rewrites tests/golden/*.npz
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import moe_oracle as orc  # noqa: E402


# ------------------------------------------------------------------ reference import
def _stub_timm():
    import transformers.activations  # noqa: F401  (must be imported before the stub exists)

    def mod(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = m
        return m

    timm = mod("timm")
    models = mod("timm.models")
    layers = mod("timm.models.layers")
    vt = mod("timm.models.vision_transformer")
    reg = mod("timm.models.registry")
    timm.models = models
    models.layers, models.vision_transformer, models.registry = layers, vt, reg

    def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            return x

    layers.trunc_normal_ = trunc_normal_
    layers.DropPath = DropPath
    vt._cfg = lambda **kw: kw
    vt.PatchEmbed = torch.nn.Identity
    reg.register_model = lambda f: f


def import_reference():
    _stub_timm()
    sys.path.insert(0, REF)
    import models.sharded_moe as sm
    import models.layer as layer
    import models.experts as experts
    import models.multiway_transformer as mt
    return sm, layer, experts, mt


# ----------------------------------------------------------------------------- helpers
def sparse_of_combine(cw: torch.Tensor):
    """[S,E,C] -> idx [S,2] int64 (flat e*C+c, ascending, -1 pad), w [S,2] fp32."""
    S, E, C = cw.shape
    flat = cw.reshape(S, E * C)
    idx = np.full((S, 2), -1, dtype=np.int64)
    w = np.zeros((S, 2), dtype=np.float32)
    nz = torch.nonzero(flat)
    fill = np.zeros(S, dtype=np.int64)
    for s, j in nz.tolist():
        k = fill[s]
        assert k < 2
        idx[s, k] = j
        w[s, k] = float(flat[s, j])
        fill[s] += 1
    return idx, w


def gate_case(sm, logits, noise, cf, min_cap):
    sm.gumbel_rsample = lambda shape, device: noise.clone()
    l_aux, cw, mask, cnt = sm.top2gating(logits.clone(), cf, min_cap)
    assert torch.equal(mask, cw.bool())
    idx, w = sparse_of_combine(cw)
    return dict(logits=logits.numpy(), noise=noise.numpy(), cf=np.float64(cf), min_cap=np.int64(min_cap),
                C=np.int64(cw.shape[2]), l_aux=l_aux.numpy(), exp_counts=cnt.numpy(), idx=idx, w=w)


def build_ref_moe(layer, mt, p, M, I, E, n_vis, **kw):
    mlp = mt.MultiwayTransformerMLP("quick_gelu", M, I)
    moe = layer.MoE(hidden_size=M, expert=mlp, num_experts=E, k=2, num_tokens=n_vis, **kw)
    sd = {"moe.gate.wg.weight": p["wg"]}
    if n_vis:
        sd["moe.vision_embed"] = p["vision_embed"]
        sd["moe.text_embed"] = p["text_embed"]
    for e in range(E):
        sd[f"moe.experts.experts.{e}.fc1.weight"] = p["w1"][e]
        sd[f"moe.experts.experts.{e}.fc1.bias"] = p["b1"][e]
        sd[f"moe.experts.experts.{e}.fc2.weight"] = p["w2"][e]
        sd[f"moe.experts.experts.{e}.fc2.bias"] = p["b2"][e]
    missing, unexpected = moe.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return moe


def run_ref_layer(sm, moe, h, noise, dout, aux_w, mode, train=True):
    sm.gumbel_rsample = lambda shape, device: noise.clone()
    moe.train(train)
    moe.zero_grad()
    h = h.clone().requires_grad_(True)
    out, l_aux, cnt = moe(h, mode=mode)
    loss = (out * dout).sum() + aux_w * l_aux
    loss.backward()
    g = {n: (q.grad.clone() if q.grad is not None else torch.zeros_like(q)) for n, q in moe.named_parameters()}
    return out.detach(), l_aux.detach(), cnt, h.grad.clone(), g


def stack_expert_grads(g, E):
    w1 = torch.stack([g[f"moe.experts.experts.{e}.fc1.weight"] for e in range(E)])
    b1 = torch.stack([g[f"moe.experts.experts.{e}.fc1.bias"] for e in range(E)])
    w2 = torch.stack([g[f"moe.experts.experts.{e}.fc2.weight"] for e in range(E)])
    b2 = torch.stack([g[f"moe.experts.experts.{e}.fc2.bias"] for e in range(E)])
    return w1, b1, w2, b2


# ------------------------------------------------------------------------------- main
def main():
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    sm, layer, experts, mt = import_reference()
    real_gumbel = sm.gumbel_rsample

    # --- noise: the oracle's restatement equals the reference draw bit for bit
    torch.manual_seed(77)
    ref_noise = real_gumbel((257, 32), torch.device("cpu"))
    torch.manual_seed(77)
    our_noise = orc.gumbel_noise((257, 32))
    assert torch.equal(ref_noise, our_noise), "gumbel restatement differs from reference"
    np.savez_compressed(os.path.join(OUT, "noise_seed77.npz"), noise=ref_noise.numpy())

    # --- G1/G2: gate-only cases
    cases = {}
    g = torch.Generator().manual_seed(1)

    def rnd(S, E, scale=1.0):
        return torch.randn(S, E, generator=g) * scale, orc.gumbel_noise((S, E), generator=g)

    cases["rand_10x4"] = gate_case(sm, *rnd(10, 4), 1.0, 4)
    cases["rand_64x32"] = gate_case(sm, *rnd(64, 32), 1.0, 4)
    cases["rand_1896x32"] = gate_case(sm, *rnd(1896, 32), 1.0, 4)
    cases["rand_1000x32_sharp"] = gate_case(sm, *rnd(1000, 32, 4.0), 1.0, 4)
    cases["rand_777x8_cf125"] = gate_case(sm, *rnd(777, 8), 1.25, 4)
    cases["rand_300x32_eval2"] = gate_case(sm, *rnd(300, 32), 2.0, 4)
    cases["rand_5x32_mincap"] = gate_case(sm, *rnd(5, 32), 1.0, 4)          # S/E*2 < 4
    cases["rand_3x2"] = gate_case(sm, *rnd(3, 2), 1.0, 4)
    cases["rand_130x3"] = gate_case(sm, *rnd(130, 3), 1.0, 4)               # E not a power of two
    # all-equal rows -> argmax tie goes to index 0; noise decides the 2nd
    lg = torch.zeros(40, 32)
    cases["ties_allequal"] = gate_case(sm, lg, orc.gumbel_noise((40, 32), generator=g), 1.0, 4)
    # all-equal rows AND zero noise: both argmaxes resolved by lowest index
    cases["ties_allequal_nonoise"] = gate_case(sm, torch.zeros(40, 32), torch.zeros(40, 32), 1.0, 4)
    # exact pairwise ties at the top
    lg, nz = rnd(96, 32)
    top = lg.max(dim=1, keepdim=True).values
    lg[:, 7] = top[:, 0]
    lg[:, 19] = top[:, 0]
    cases["ties_pairwise"] = gate_case(sm, lg, nz, 1.0, 4)
    # every token wants expert 0 then 1: capacity overflow (SURVEY §8c example)
    lg = torch.tensor([[5.0, 1.0, 0.0, 0.0]]).repeat(10, 1)
    cases["overflow_10x4_cf05"] = gate_case(sm, lg, torch.zeros(10, 4), 0.5, 4)
    lg = torch.zeros(200, 32)
    lg[:, 3] = 6.0
    lg[:, 9] = 3.0
    cases["overflow_200x32"] = gate_case(sm, lg, orc.gumbel_noise((200, 32), generator=g) * 0.1, 1.0, 4)
    # huge gap: 2nd gate underflows to 0 -> weight 0 -> not dispatched but consumes its location
    lg = torch.randn(64, 8, generator=g)
    lg[::3, 2] += 200.0
    cases["underflow_64x8"] = gate_case(sm, lg, orc.gumbel_noise((64, 8), generator=g), 1.0, 4)
    # large magnitudes / negative logits
    lg, nz = rnd(128, 32, 30.0)
    cases["large_128x32"] = gate_case(sm, lg - 50.0, nz, 1.0, 4)
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[f"{name}/{k}"] = v
    flat["__cases__"] = np.array(sorted(cases.keys()))
    np.savez_compressed(os.path.join(OUT, "gate_cases.npz"), **flat)
    print("gate cases:", len(cases))

    # --- G3: small layer, all modes, forward + every gradient
    M, I, E, B, n_vis, T = 32, 64, 32, 2, 5, 9
    p = orc.make_params(M, I, E, seed=11)
    out = {"M": M, "I": I, "E": E, "B": B, "T": T, "n_vis": n_vis, "aux_w": 0.7}
    for k, v in p.items():
        out[f"p/{k}"] = v.numpy()
    g = torch.Generator().manual_seed(5)
    for mode, nv in (("vl", n_vis), ("vision", n_vis), ("text", n_vis), ("plain", None)):
        # B*T = 18 tokens over 32 experts is too sparse to hit capacity; use a bigger batch for drops
        for tag, Bx in ((("", B), ("_big", 24)) if mode == "vl" else (("_big", 24),)):
            h = torch.randn(Bx, T, M, generator=g)
            noise = orc.gumbel_noise((Bx * T, E), generator=g)
            dout = torch.randn(Bx, T, M, generator=g)
            moe = build_ref_moe(layer, mt, p, M, I, E, nv)
            o, la, cnt, dh, gr = run_ref_layer(sm, moe, h, noise, dout, 0.7, "vl" if mode == "plain" else mode)
            w1, b1, w2, b2 = stack_expert_grads(gr, E)
            key = f"{mode}{tag}"
            rec = dict(h=h, noise=noise, dout=dout, out=o, l_aux=la, exp_counts=cnt, dh=dh, dwg=gr["moe.gate.wg.weight"],
                       dw1=w1, db1=b1, dw2=w2, db2=b2)
            if nv:
                rec["dvision_embed"] = gr["moe.vision_embed"]
                rec["dtext_embed"] = gr["moe.text_embed"]
            for k, v in rec.items():
                out[f"{key}/{k}"] = v.numpy()
    # eval-mode capacity factor switch (train cf 1.0 / eval cf 2.0), min_capacity 4
    h = torch.randn(24, T, M, generator=g)
    noise = orc.gumbel_noise((24 * T, E), generator=g)
    moe = build_ref_moe(layer, mt, p, M, I, E, n_vis, capacity_factor=1.0, eval_capacity_factor=2.0)
    sm.gumbel_rsample = lambda shape, device: noise.clone()
    moe.eval()
    with torch.no_grad():
        o, la, cnt = moe(h, mode="vl")
    out["eval2/h"], out["eval2/noise"], out["eval2/out"] = h.numpy(), noise.numpy(), o.numpy()
    out["eval2/l_aux"], out["eval2/exp_counts"] = la.numpy(), cnt.numpy()
    np.savez_compressed(os.path.join(OUT, "layer_small.npz"), **out)
    print("layer_small done")

    # --- G5: expert-parallel expectation = per-rank independent reference outputs
    ranks = 8
    ep = {"ranks": ranks, "aux_w": 0.3}
    g = torch.Generator().manual_seed(9)
    moe = build_ref_moe(layer, mt, p, M, I, E, n_vis)
    acc = None
    for r in range(ranks):
        h = torch.randn(6, T, M, generator=g)
        noise = orc.gumbel_noise((6 * T, E), generator=g)
        dout = torch.randn(6, T, M, generator=g)
        o, la, cnt, dh, gr = run_ref_layer(sm, moe, h, noise, dout, 0.3, "vl")
        for k, v in dict(h=h, noise=noise, dout=dout, out=o, l_aux=la, exp_counts=cnt, dh=dh).items():
            ep[f"r{r}/{k}"] = v.numpy()
        w1, b1, w2, b2 = stack_expert_grads(gr, E)
        cur = dict(dwg=gr["moe.gate.wg.weight"], dw1=w1, db1=b1, dw2=w2, db2=b2,
                   dvision_embed=gr["moe.vision_embed"], dtext_embed=gr["moe.text_embed"])
        acc = cur if acc is None else {k: acc[k] + cur[k] for k in cur}
    for k, v in acc.items():  # SUM over ranks of per-rank reference grads
        ep[f"sum/{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "ep_small.npz"), **ep)
    print("ep_small done")

    # --- G4: config-1 full shape (EVE-Base, B=8, 197+40 tokens, fp32): slices + checksums
    M, I, E, B, n_vis, T = 768, 3072, 32, 8, 197, 237
    p = orc.make_params(M, I, E, seed=1234)
    h = orc.synth_hidden(B, T, M, seed=4321)
    g = torch.Generator().manual_seed(99)
    noise = orc.gumbel_noise((B * T, E), generator=g)
    dout = torch.randn(B, T, M, generator=g)
    moe = build_ref_moe(layer, mt, p, M, I, E, n_vis)
    o, la, cnt, dh, gr = run_ref_layer(sm, moe, h, noise, dout, 1.0, "vl")
    w1, b1, w2, b2 = stack_expert_grads(gr, E)
    sl = slice(0, None, 29)  # every 29th token of the flattened sequence
    cfg1 = dict(param_seed=1234, hidden_seed=4321, noise_seed=99, aux_w=1.0,
                tok_index=np.arange(B * T)[sl],
                out_slice=o.reshape(B * T, M)[sl].numpy(), out_rowsum=o.reshape(B * T, M).double().sum(1).numpy(),
                dh_slice=dh.reshape(B * T, M)[sl].numpy(), dh_rowsum=dh.reshape(B * T, M).double().sum(1).numpy(),
                l_aux=la.numpy(), exp_counts=cnt.numpy(), dwg=gr["moe.gate.wg.weight"].numpy(),
                dvision_embed=gr["moe.vision_embed"].numpy(), dtext_embed=gr["moe.text_embed"].numpy(),
                db1=b1.numpy(), db2=b2.numpy(),
                dw1_e3_rows=w1[3, ::97].numpy(), dw2_e3_rows=w2[3, ::31].numpy(),
                dw1_abs_sum=w1.double().abs().sum((1, 2)).numpy(), dw2_abs_sum=w2.double().abs().sum((1, 2)).numpy())
    np.savez_compressed(os.path.join(OUT, "layer_cfg1.npz"), **cfg1)
    print("layer_cfg1 done; exp_counts", cnt.tolist())


if __name__ == "__main__":
    main()

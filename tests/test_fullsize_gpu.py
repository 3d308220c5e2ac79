"""Full-size (BASELINE.json configs 2, 4, 5 layer shapes) checks on the GPU.

The dense oracle cannot run at these sizes, so the checks are (a) bit-exact routing against the
index-form oracle on the GPU's own logits (O(S*E) on the CPU), (b) routing invariants, (c) the
layer output and input gradient of a random SAMPLE of tokens against a per-token CPU evaluation
of the reference formulae, (d) exact linearity of the backward in dout, (e) determinism."""
import numpy as np
import pytest
import torch

from oracle import moe_oracle as orc
from tests.helpers import build_moe, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _layer(M, I, E, n_vis, seed, dtype):
    import eve_amd
    torch.manual_seed(seed)
    moe = eve_amd.MoE(M, eve_amd.MultiwayTransformerMLP("quick_gelu", M, I), num_experts=E, k=2, num_tokens=n_vis)
    with torch.no_grad():
        for p in moe.parameters():
            torch.nn.init.trunc_normal_(p, std=0.02, a=-0.04, b=0.04)
    return moe.to(dtype).to(DEV)


def _check_routing(moe, h, noise, S, E, n_vis, T, cf=1.0):
    from eve_amd import ops
    m = moe.moe
    ve, te = m.vision_embed.detach().to(h.dtype), m.text_embed.detach().to(h.dtype)
    logits = ops.gate_logits(h.reshape(S, -1), ve, te, m.gate.wg.weight.detach().float(), T, n_vis, ops.MODES["vl"])
    ref = orc.index_top2gating(logits.cpu(), noise.cpu(), cf, 4)
    slot = m.last_slot.cpu().long()
    assert torch.equal(slot, torch.stack([ref.slot1, ref.slot2], 1)), "routing differs from the index-form oracle"
    assert torch.equal(m.exp_counts, ref.exp_counts)
    used = slot[slot >= 0]
    assert used.numel() == used.unique().numel(), "two tokens share a slot"
    assert int(used.max()) < E * ref.C
    per_expert = torch.bincount(torch.div(used, ref.C, rounding_mode="floor"), minlength=E)
    assert int(per_expert.max()) <= ref.C
    return ref, logits.cpu()


def _sample_check(moe, h, dout, out, dh, ref, aux_w, n_vis, T, E, idx, tol):
    """Per-token evaluation of the reference formulae (SURVEY.md 8 A5 spec) for tokens `idx`."""
    m = moe.moe
    f = lambda t: t.detach().float().cpu()
    wg, ve, te = f(m.gate.wg.weight), f(m.vision_embed), f(m.text_embed)
    ex = m.experts.experts
    S = h.shape[0] * h.shape[1]
    hs, ds = h.reshape(S, -1), dout.reshape(S, -1)
    os_, dhs = out.reshape(S, -1), (dh.reshape(S, -1) if dh is not None else None)
    cnt1 = ref.exp_counts.float()
    worst_o = worst_g = 0.0
    for s in idx.tolist():
        x0 = hs[s].detach().cpu()
        hv = x0.float().requires_grad_(True)
        emb = ve if (s % T) < n_vis else te
        x = (hv.to(h.dtype) + emb.to(h.dtype)).float() if h.dtype != torch.float32 else hv + emb
        if h.dtype != torch.float32:  # straight-through the bf16 rounding of the modality add
            x = (hv + emb) + (x - (hv + emb)).detach()
        gates = torch.softmax(x @ wg.t(), dim=0)
        i1, i2 = int(ref.i1[s]), int(ref.i2[s])
        g1 = gates[i1] * float(ref.keep1[s])
        g2 = gates[i2] * float(ref.keep2[s])
        den = (g1 + g2).clamp(min=orc.FP32_EPS)
        o = torch.zeros_like(hv)
        for e, g, sl in ((i1, g1, int(ref.slot1[s])), (i2, g2, int(ref.slot2[s]))):
            if sl >= 0:
                o = o + (g / den) * orc.expert_ffn(x, f(ex[e].fc1.weight), f(ex[e].fc1.bias), f(ex[e].fc2.weight),
                                                   f(ex[e].fc2.bias))
        worst_o = max(worst_o, float((o.detach() - f(os_[s])).abs().max() / o.detach().abs().max().clamp(min=1e-3)))
        if dhs is not None:
            loss = (o * f(ds[s])).sum() + aux_w * E * (gates * cnt1).sum() / (S * S)
            loss.backward()
            worst_g = max(worst_g, float((hv.grad - f(dhs[s])).abs().max() / hv.grad.abs().max().clamp(min=1e-3)))
    assert worst_o < tol, f"sampled forward error {worst_o}"
    assert worst_g < tol, f"sampled input-gradient error {worst_g}"


def _expert_grad_check(moe, h, dout, ref, n_vis, T, E, experts):
    """dw1 / db1 / dw2 / db2 of a few experts against the CPU oracle's formulae evaluated in fp32 on exactly the
    rows the (oracle-checked) routing sent to them -- at the full model dims (Large: M = 1024, I = 4096)."""
    from tests.helpers import assert_close_scaled
    m = moe.moe
    f = lambda t: t.detach().float().cpu()
    S = h.shape[0] * h.shape[1]
    hs, ds = f(h.reshape(S, -1)), f(dout.reshape(S, -1))
    ve, te = f(m.vision_embed), f(m.text_embed)
    pos = torch.arange(S) % T
    emb = torch.where((pos < n_vis)[:, None], ve[None, :], te[None, :])
    x = (hs.to(h.dtype) + emb.to(h.dtype)).float()  # what is dispatched: h + modality embedding, rounded to bf16
    for e in experts:
        rows, ws = [], []
        for slot, w in ((ref.slot1, ref.w1), (ref.slot2, ref.w2)):
            sel = torch.nonzero((slot >= 0) & (torch.div(slot, ref.C, rounding_mode="floor") == e)).flatten()
            rows.append(sel), ws.append(w[sel])
        rows, ws = torch.cat(rows), torch.cat(ws)
        ex = m.experts.experts[e]
        p = [f(t).requires_grad_(True) for t in (ex.fc1.weight, ex.fc1.bias, ex.fc2.weight, ex.fc2.bias)]
        y = orc.expert_ffn(x[rows], *p)
        (y * (ws[:, None] * ds[rows])).sum().backward()
        for t, q, name in zip((ex.fc1.weight, ex.fc1.bias, ex.fc2.weight, ex.fc2.bias), p, ("dw1", "db1", "dw2", "db2")):
            assert_close_scaled(t.grad, q.grad, 2e-2, 4e-2, f"expert {e} {name}")


@pytest.mark.parametrize("name,M,I,B,n_vis", [("cfg2_base_b64", 768, 3072, 64, 197), ("cfg4_large_vqa480", 1024, 4096, 16, 901),
                                             ("cfg3_large_b32", 1024, 4096, 32, 197)])
def test_fullsize_train_step_bf16(name, M, I, B, n_vis):
    E, T = 32, n_vis + 40
    S = B * T
    moe = _layer(M, I, E, n_vis, seed=1, dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(2)
    h = torch.nn.functional.layer_norm(torch.randn(B, T, M, generator=g), (M,)).to(torch.bfloat16).to(DEV)
    dout = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(DEV)
    noise = orc.gumbel_noise((S, E), generator=g).to(DEV)

    def run(scale):
        for p in moe.parameters():
            p.grad = None
        x = h.detach().requires_grad_(True)
        moe.moe.noise_override = noise
        out, l_aux, cnt = moe(x, mode="vl")
        torch.autograd.backward([out, l_aux], [dout * scale, torch.full_like(l_aux, 0.5 * scale)])
        torch.cuda.synchronize()
        grads = [p.grad.clone() for p in moe.parameters()]
        return out.detach(), l_aux.detach(), x.grad, grads

    out, l_aux, dh, grads = run(1.0)
    ref, _ = _check_routing(moe, h, noise, S, E, n_vis, T)
    assert torch.isfinite(out.float()).all() and torch.isfinite(dh.float()).all()
    np.testing.assert_allclose(l_aux.item(), ref.l_aux.item(), rtol=1e-4)
    idx = torch.randperm(S, generator=g)[:48]
    _sample_check(moe, h, dout, out, dh, ref, 0.5, n_vis, T, E, idx, tol=2e-2)
    _expert_grad_check(moe, h, dout, ref, n_vis, T, E, experts=(0, 17, 31))
    # determinism + exact linearity of the backward in the upstream gradient (x2 is exact in bf16/fp32)
    out_b, _, dh_b, grads_b = run(1.0)
    assert torch.equal(out, out_b) and torch.equal(dh, dh_b)
    assert all(torch.equal(a, b) for a, b in zip(grads, grads_b))
    _, _, dh2, grads2 = run(2.0)
    assert torch.equal(dh2.float(), dh.float() * 2)
    assert all(torch.equal(a.float() * 2, b.float()) for a, b in zip(grads, grads2))


def test_fullsize_rerank_inference_cfg5():
    """config 5: EVE-Large, k_test=128 x (577+40) tokens, eval mode, forward only (S = 78 976)."""
    M, I, E, B, n_vis = 1024, 4096, 32, 128, 577
    T = n_vis + 40
    S = B * T
    moe = _layer(M, I, E, n_vis, seed=3, dtype=torch.bfloat16).eval()
    g = torch.Generator().manual_seed(4)
    h = torch.nn.functional.layer_norm(torch.randn(B, T, M, generator=g), (M,)).to(torch.bfloat16).to(DEV)
    noise = orc.gumbel_noise((S, E), generator=g).to(DEV)
    with torch.no_grad():
        moe.moe.noise_override = noise
        out, l_aux, cnt = moe(h, mode="vl")
    torch.cuda.synchronize()
    ref, _ = _check_routing(moe, h, noise, S, E, n_vis, T)
    assert ref.C == 4936
    idx = torch.randperm(S, generator=g)[:32]
    _sample_check(moe, h, torch.zeros_like(h), out, None, ref, 0.0, n_vis, T, E, idx, tol=2e-2)


def test_expert_parallel_layout_emulated_on_one_gpu():
    """The [ep, E_local, C, M] group layout (reference sharded_moe.py:370) and the weight-gradient
    sum over source ranks, emulated on one GPU: two 'ranks' own 16 experts each and exchange their
    capacity buffers by slicing; results must equal the 32-expert single-rank run bit for bit."""
    from eve_amd import ops
    E, C, M, I, ep = 32, 300, 256, 512, 2
    e_local = E // ep
    g = torch.Generator().manual_seed(5)
    bf = torch.bfloat16
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(bf).to(DEV)
    w1, b1, w2, b2 = mk(E, I, M, sc=0.05), mk(E, I, sc=0.1), mk(E, M, I, sc=0.05), mk(E, M, sc=0.1)
    xd = [mk(E, C, M) for _ in range(ep)]        # each source rank's dispatched buffer
    dy = [mk(E, C, M) for _ in range(ep)]
    act = ops.ACTS["quick_gelu"]
    ref_y, ref_dx, ref_dw = [], [], None
    for r in range(ep):
        z, a, y = ops.ffn_fwd(xd[r], w1, b1, w2, b2, None, act)
        dx, dw1, db1, dw2, db2 = ops.ffn_bwd(dy[r], xd[r], z, a, w1, w2, None, act)
        ref_y.append(y), ref_dx.append(dx)
    # fp32 reference for the summed weight gradients
    for j in range(ep):
        lo, hi = j * e_local, (j + 1) * e_local
        recv = torch.stack([xd[r][lo:hi] for r in range(ep)]).reshape(ep * e_local, C, M).contiguous()
        recv_dy = torch.stack([dy[r][lo:hi] for r in range(ep)]).reshape(ep * e_local, C, M).contiguous()
        z, a, y = ops.ffn_fwd(recv, w1[lo:hi].contiguous(), b1[lo:hi].contiguous(), w2[lo:hi].contiguous(),
                              b2[lo:hi].contiguous(), None, act)
        dx, dw1, db1, dw2, db2 = ops.ffn_bwd(recv_dy, recv, z, a, w1[lo:hi].contiguous(), w2[lo:hi].contiguous(), None, act)
        y, dx = y.view(ep, e_local, C, M), dx.view(ep, e_local, C, M)
        for r in range(ep):
            assert torch.equal(y[r], ref_y[r][lo:hi]) and torch.equal(dx[r], ref_dx[r][lo:hi])
        # summed over source ranks, against fp64 arithmetic for one expert
        e = lo + 3
        A = torch.cat([xd[r][e] for r in range(ep)]).double().cpu()
        Z = A @ w1[e].double().cpu().t() + b1[e].double().cpu()
        H = Z * torch.sigmoid(1.702 * Z)
        DY = torch.cat([dy[r][e] for r in range(ep)]).double().cpu()
        assert rel_err(dw2[3].cpu(), DY.t() @ H) < 2e-2
        assert rel_err(db2[3].cpu(), DY.sum(0)) < 2e-2


@pytest.mark.parametrize("name", ["rand_64x32", "overflow_200x32", "underflow_64x8", "ties_pairwise"])
def test_topkgate_dense_compat_matches_reference(gate_cases, name):
    """`top2gating` keeps the reference's dense return contract for callers that want it."""
    from eve_amd.sharded_moe import top2gating
    c = gate_cases[name]
    logits, noise = torch.from_numpy(c["logits"]).to(DEV), torch.from_numpy(c["noise"]).to(DEV)
    l_aux, cw, mask, cnt = top2gating(logits, float(c["cf"]), int(c["min_cap"]), noise=noise)
    S = logits.shape[0]
    dense = np.zeros((S, cw.shape[1] * cw.shape[2]), np.float32)
    for s in range(S):
        for k in range(2):
            if c["idx"][s, k] >= 0:
                dense[s, c["idx"][s, k]] = c["w"][s, k]
    np.testing.assert_allclose(cw.reshape(S, -1).cpu().numpy(), dense, rtol=2e-6, atol=1e-9)
    assert np.array_equal(mask.reshape(S, -1).cpu().numpy(), dense != 0)
    assert cnt.device.type == "cpu" and np.array_equal(cnt.numpy(), c["exp_counts"])
    assert cw.shape[2] == int(c["C"])


@pytest.mark.parametrize("case", ["first_owns_slot0_second_dropped", "first_dropped_second_owns_slot0"])
def test_dense_compat_slot0_with_a_dropped_partner(case):
    """A token that holds slot 0 (expert 0, location 0) while its other choice is over capacity: the dense
    compatibility tensors must keep its weight at [s, 0, 0] (a dropped choice must not be scattered onto slot 0).
    Expected: the oracle's dense one-hot formulation (reference sharded_moe.py:172-235)."""
    from eve_amd.sharded_moe import top2gating
    S, E = 16, 4  # capacity_factor 0.5 -> C = max(ceil(16 / 4 * 1.0), 4) = 4
    logits = torch.full((S, E), -4.0)
    if case == "first_owns_slot0_second_dropped":
        logits[0, 0], logits[0, 1] = 4.0, 2.0          # token 0: first expert 0 (slot 0), second expert 1 ...
        logits[1:7, 1], logits[1:7, 2] = 4.0, 2.0      # ... which six later tokens pick first: location 6 >= C
        logits[7:, 3], logits[7:, 2] = 4.0, 2.0
        tok = 0
    else:
        logits[0:6, 1], logits[0:6, 2] = 4.0, 2.0      # tokens 0-5 pick expert 1 first: tokens 4, 5 are over capacity
        logits[5, 2], logits[5, 0] = -4.0, 2.0         # token 5: first dropped, second = expert 0, location 0 -> slot 0
        logits[6:, 3], logits[6:, 2] = 4.0, 2.0
        tok = 5
    noise = torch.zeros(S, E)
    ref_l, ref_cw, ref_mask, ref_cnt = orc.dense_top2gating(logits, noise, 0.5, 4)
    assert ref_cw.shape[2] == 4 and float(ref_cw[tok, 0, 0]) > 0 and int((ref_cw[tok] != 0).sum()) == 1
    for _ in range(5):  # a racy scatter would show up on some repetition
        l_aux, cw, mask, cnt = top2gating(logits.to(DEV), 0.5, 4, noise=noise.to(DEV))
        torch.testing.assert_close(cw.cpu(), ref_cw, rtol=2e-6, atol=1e-9)
        assert torch.equal(mask.cpu(), ref_mask) and torch.equal(cnt, ref_cnt)

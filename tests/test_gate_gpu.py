"""HIP gate kernels vs the reference's golden vectors and the CPU oracle (bit-exact routing)."""
import numpy as np
import pytest
import torch

from oracle import moe_oracle as orc
from eve_amd import ops
from tests.helpers import sparse_pairs
from tests.test_oracle_golden import GATE_CASES

pytestmark = pytest.mark.gpu


def _ops():
    from eve_amd import ops
    return ops


@pytest.mark.parametrize("name", GATE_CASES)
def test_gate_matches_reference_golden(gate_cases, name):
    ops = _ops()
    c = gate_cases[name]
    dev = torch.device("cuda:0")
    logits, noise = torch.from_numpy(c["logits"]).to(dev), torch.from_numpy(c["noise"]).to(dev)
    S, E = logits.shape
    C = ops.capacity(S, E, float(c["cf"]), int(c["min_cap"]))
    assert C == int(c["C"])
    r = ops.gate_top2(logits, noise, C)
    torch.cuda.synchronize()
    assert np.array_equal(r.exp_counts.cpu().numpy(), c["exp_counts"])
    idx, w = sparse_pairs(r.slot.cpu().numpy(), r.weight.cpu().numpy())
    assert np.array_equal(idx, c["idx"]), "routing slots differ from the reference (must be bit-exact)"
    np.testing.assert_allclose(w, c["w"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(r.l_aux.cpu().numpy()[0], c["l_aux"], rtol=1e-5)
    # inverse map consistent with the forward map
    slot = r.slot.cpu().numpy()
    st = r.slot_token.cpu().numpy()
    sw = r.slot_weight.cpu().numpy()
    wt = r.weight.cpu().numpy()
    filled = 0
    for s in range(S):
        for k in range(2):
            if slot[s, k] >= 0:
                assert st[slot[s, k]] == s and sw[slot[s, k]] == wt[s, k]
                filled += 1
    assert (st >= 0).sum() == filled
    # rows = min(C, first + second choices)
    ref = orc.index_top2gating(logits.cpu(), noise.cpu(), float(c["cf"]), int(c["min_cap"]))
    cnt2 = torch.bincount(ref.i2, minlength=E)
    assert np.array_equal(r.rows.cpu().numpy(), torch.minimum(ref.exp_counts + cnt2, torch.tensor(C)).numpy())
    assert np.array_equal(r.idx.cpu().numpy(), torch.stack([ref.i1, ref.i2], 1).numpy())
    assert np.array_equal(r.loc.cpu().numpy(), torch.stack([ref.loc1, ref.loc2], 1).numpy())


@pytest.mark.parametrize("S,E", [(1, 32), (63, 32), (64, 32), (65, 32), (255, 32), (256, 32), (257, 32), (4097, 32),
                                 (15168, 32), (78976, 32), (1000, 2), (1000, 5), (3000, 64)])
def test_gate_random_sizes_vs_oracle(S, E):
    ops = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(S * 131 + E)
    logits = torch.randn(S, E, generator=g) * 1.5
    noise = orc.gumbel_noise((S, E), generator=g)
    ref = orc.index_top2gating(logits, noise, 1.0, 4)
    r = ops.gate_top2(logits.to(dev), noise.to(dev), ref.C)
    torch.cuda.synchronize()
    assert torch.equal(r.idx.cpu().long(), torch.stack([ref.i1, ref.i2], 1))
    assert torch.equal(r.loc.cpu().long(), torch.stack([ref.loc1, ref.loc2], 1))
    assert torch.equal(r.slot.cpu().long(), torch.stack([ref.slot1, ref.slot2], 1))
    assert torch.equal(r.exp_counts.cpu(), ref.exp_counts)
    torch.testing.assert_close(r.weight.cpu(), torch.stack([ref.w1, ref.w2], 1), rtol=2e-6, atol=1e-9)
    torch.testing.assert_close(r.gates.cpu(), ref.gates, rtol=2e-6, atol=1e-10)
    torch.testing.assert_close(r.l_aux.cpu()[0], ref.l_aux, rtol=1e-5, atol=0)


def test_gate_is_deterministic():
    ops = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    logits = torch.randn(15168, 32, generator=g).to(dev)
    noise = orc.gumbel_noise((15168, 32), generator=g).to(dev)
    a = ops.gate_top2(logits, noise, 948)
    b = ops.gate_top2(logits, noise, 948)
    for f in ("slot", "weight", "slot_token", "slot_weight", "l_aux", "rows", "exp_counts", "gates"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f


@pytest.mark.parametrize("S,E,cf", [(216, 32, 1.0), (1896, 32, 1.0), (500, 8, 0.5)])
def test_gate_backward_vs_oracle_autograd(S, E, cf):
    ops = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(S + E)
    logits = (torch.randn(S, E, generator=g)).requires_grad_(True)
    noise = orc.gumbel_noise((S, E), generator=g)
    dw = torch.randn(S, 2, generator=g)
    dl = torch.tensor(0.37)
    ref = orc.index_top2gating(logits, noise, cf, 4)
    ((torch.stack([ref.w1, ref.w2], 1) * dw).sum() + dl * ref.l_aux).backward()
    r = ops.gate_top2(logits.detach().to(dev), noise.to(dev), ref.C)
    dlogits = ops.gate_top2_bwd(r, dw.to(dev), dl.reshape(1).to(dev))
    torch.testing.assert_close(dlogits.cpu(), logits.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["vl", "vision", "text", "none"])
@pytest.mark.parametrize("B,T,n_vis,M,E", [(3, 9, 5, 64, 32), (8, 237, 197, 768, 32), (2, 50, 10, 1024, 32), (5, 7, 3, 40, 4)])
def test_gate_logits_and_wgrad(dtype, mode, B, T, n_vis, M, E):
    from eve_amd import _lib
    ops = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * T + M)
    h = torch.randn(B, T, M, generator=g).to(dtype)
    p = {"vision_embed": (torch.randn(M, generator=g) * 0.02).to(dtype), "text_embed": (torch.randn(M, generator=g) * 0.02).to(dtype)}
    wg = torch.randn(E, M, generator=g) * 0.02
    code = ops.MODES[mode]
    x = orc.modality_input(h, p, mode, n_vis if mode != "none" else None).float()
    ref = x @ wg.t()
    ve = p["vision_embed"].to(dev) if mode != "none" else None
    te = p["text_embed"].to(dev) if mode != "none" else None
    h2d = h.reshape(-1, M).to(dev)
    logits = ops.gate_logits(h2d, ve, te, wg.to(dev), T, n_vis, code)
    torch.testing.assert_close(logits.cpu(), ref, rtol=1e-4, atol=2e-5)
    dlog = torch.randn(B * T, E, generator=g)
    dwg = ops.gate_wgrad(h2d, ve, te, dlog.to(dev), T, n_vis, code)
    torch.testing.assert_close(dwg.cpu(), dlog.t() @ x, rtol=1e-4, atol=1e-4)


def test_fused_gumbel_is_torch_gumbel_bit_for_bit():
    """gumbel_rsample (one torch.rand + one HIP kernel) vs torch.distributions.Gumbel(0,1).rsample, the
    call the reference makes (sharded_moe.py:70-77): same values, same generator consumption."""
    from eve_amd import sharded_moe as sm
    dev = torch.device("cuda", 0)
    assert sm._fused_gumbel_matches_torch(dev), "fused kernel differs from torch on this build (fallback would be used)"
    for shape in [(15168, 32), (7, 3), (1, 1)]:
        torch.manual_seed(1234)
        want = sm._torch_gumbel(shape, dev)
        want_next = torch.rand(5, device=dev)
        torch.manual_seed(1234)
        got = sm.gumbel_rsample(shape, dev)
        got_next = torch.rand(5, device=dev)
        assert torch.equal(want, got)
        assert torch.equal(want_next, got_next)  # generator advanced identically
    # extreme draws: r = 0 and the largest fp32 below 1
    r = torch.tensor([0.0, 1.0 - 2.0 ** -24], device=dev)
    tiny, eps = torch.finfo(torch.float32).tiny, torch.finfo(torch.float32).eps
    u = tiny + r * ((1.0 - eps) - tiny)
    assert torch.equal(ops.gumbel_from_uniform(r), -torch.log(-torch.log(u)))


@pytest.mark.parametrize("S,E", [(15168, 32), (1896, 32), (130, 32), (77, 5), (64, 1), (300, 40)])
def test_gate_with_uniform_draws_equals_gate_with_gumbel_noise(S, E):
    """eve_gate_top2_fwd fed the torch.rand draws (transform inside the kernel) vs fed the finished Gumbel
    variates of the same draws: every output identical, bit for bit."""
    dev = "cuda:0"
    g = torch.Generator().manual_seed(S * 31 + E)
    logits = torch.randn(S, E, generator=g).to(dev)
    u = torch.rand(S, E, generator=g).to(dev)
    u[0, 0] = 0.0
    u[-1, -1] = 1.0 - 2.0 ** -24
    C = ops.capacity(S, E, 1.0, 4)
    a = ops.gate_top2(logits, u, C, noise_is_uniform=True)
    b = ops.gate_top2(logits, ops.gumbel_from_uniform(u), C)
    torch.cuda.synchronize()
    for name in ("gates", "idx", "loc", "slot", "weight", "slot_token", "slot_weight", "exp_counts", "rows", "l_aux"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name


def test_layer_draws_noise_like_the_reference_generator():
    """MOELayer's in-kernel Gumbel path consumes the generator exactly like gumbel_rsample and routes like it."""
    from eve_amd import sharded_moe as sm
    dev = torch.device("cuda", 0)
    assert sm._fused_gumbel_usable(dev)
    torch.manual_seed(7)
    u, is_u = sm._draw_noise((500, 32), dev)
    nxt = torch.rand(3, device=dev)
    torch.manual_seed(7)
    n = sm.gumbel_rsample((500, 32), dev)
    nxt2 = torch.rand(3, device=dev)
    assert is_u and torch.equal(ops.gumbel_from_uniform(u), n) and torch.equal(nxt, nxt2)

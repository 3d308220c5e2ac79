"""HIP-graph replay of the whole training step (eve_amd.GraphedTrainStep) vs the eager step."""
import pytest
import torch

import eve_amd
from eve_amd import sharded_moe
from oracle import moe_oracle as orc
from tests.helpers import build_moe

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _layer(M=256, I=512, E=8, n_vis=5, dtype=torch.bfloat16):
    p = orc.make_params(M, I, E, seed=3)
    return build_moe(p, M, I, E, n_vis, device=DEV).to(dtype)


def test_graph_replay_equals_eager_with_fixed_noise(monkeypatch):
    M, E, B, T = 256, 8, 4, 37
    moe = _layer(M=M, E=E)
    g = torch.Generator().manual_seed(5)
    h = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(DEV)
    dout = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(DEV)
    noise = orc.gumbel_noise((B * T, E), generator=g).to(DEV)
    monkeypatch.setattr(sharded_moe, "gumbel_rsample", lambda shape, device: noise)

    x = h.clone().requires_grad_(True)
    out, l_aux, _ = moe(x, mode="vl")
    torch.autograd.backward([out, l_aux], [dout, torch.ones_like(l_aux)])
    want = [out.detach().clone(), l_aux.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in moe.parameters()]

    step = eve_amd.GraphedTrainStep(moe, h, dout, mode="vl")
    for _ in range(2):  # replays recompute everything into the same buffers
        o, l = step(h, dout)
        torch.cuda.synchronize()
        got = [o, l, step.dx] + [p.grad for p in moe.parameters()]
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    # new inputs through the same graph
    h2 = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(DEV)
    x2 = h2.clone().requires_grad_(True)
    out2, l2, _ = moe(x2, mode="vl")
    o, l = step(h2, dout)
    torch.cuda.synchronize()
    assert torch.equal(o, out2.detach()) and torch.equal(l, l2.detach())


def test_graph_replay_draws_fresh_noise():
    M, E, B, T = 256, 8, 4, 64
    moe = _layer(M=M, E=E)
    g = torch.Generator().manual_seed(6)
    h = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(DEV)
    dout = torch.randn(B, T, M, generator=g).to(torch.bfloat16).to(DEV)
    step = eve_amd.GraphedTrainStep(moe, h, dout, mode="vl")
    step()
    torch.cuda.synchronize()
    slot_a = moe.moe.last_slot.clone()
    counts_a = moe.moe.exp_counts.clone()
    step()
    torch.cuda.synchronize()
    assert not torch.equal(slot_a, moe.moe.last_slot)        # second choices follow the new Gumbel draw
    assert int(counts_a.sum()) == B * T == int(moe.moe.exp_counts.sum())  # first-choice counts (reference :199)

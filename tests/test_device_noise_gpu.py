"""The gate's device noise source (include/eve_moe.h, EVE_NOISE_DEVICE_PHILOX): the kernel draws its own uniform
values from a device-resident Philox4x32-10 state and advances it.  No counterpart in the reference (its draw is
torch's generator, sharded_moe.py:70-77, :186); the checker is the published Philox algorithm in numpy
(tests/helpers.py, pinned by Random123's known-answer vectors) fed to the SAME gate through the uniform-draw input."""
import numpy as np
import pytest
import torch

from eve_amd import DeviceNoise, GraphedTrainStep, ops
from oracle import moe_oracle as orc
from tests.helpers import build_moe, device_noise_uniforms

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FIELDS = ("gates", "idx", "loc", "slot", "weight", "slot_token", "slot_weight", "exp_counts", "rows", "l_aux")


def _state(t):
    seed, off = (int(v) for v in t.cpu())
    return seed % (1 << 64), off


@pytest.mark.parametrize("S,E", [(1000, 32), (15168, 32), (777, 8), (515, 27), (300, 48), (20000, 32)])
def test_gate_draws_the_philox_stream_and_advances_it(S, E):
    """E = 32 (two whole blocks per lane), E < 32 (element-wise), E > 32 (generic kernel), S > 16384 (scan + assign
    kernels): every output bit-equal to the gate fed the numpy stream's uniforms; offset += ceil(S*E/4) per call."""
    g = torch.Generator().manual_seed(S + E)
    logits = torch.randn(S, E, generator=g).to(DEV)
    C = ops.capacity(S, E, 1.0, 4)
    src = DeviceNoise(DEV, seed=0xDEADBEEFCAFEF00D)
    src.state[1] = 41
    for call in range(2):
        seed, off = _state(src.state)
        assert seed == 0xDEADBEEFCAFEF00D and off == 41 + call * ((S * E + 3) // 4)
        got = ops.gate_top2(logits, src.state, C, ops.NOISE_DEVICE_PHILOX)
        u = torch.from_numpy(device_noise_uniforms(seed, off, S, E)).to(DEV)
        want = ops.gate_top2(logits, u, C, True)
        for f in FIELDS:
            assert torch.equal(getattr(got, f), getattr(want, f)), (f, call)
    assert _state(src.state)[1] == 41 + 2 * ((S * E + 3) // 4)


def test_bad_generator_state_is_refused():
    logits = torch.randn(64, 32, device=DEV)
    with pytest.raises((TypeError, ValueError, RuntimeError)):
        ops.gate_top2(logits, torch.zeros(2, dtype=torch.float32, device=DEV), 8, ops.NOISE_DEVICE_PHILOX)
    with pytest.raises((TypeError, ValueError, RuntimeError)):
        ops.gate_top2(logits, torch.zeros(3, dtype=torch.int64, device=DEV), 8, ops.NOISE_DEVICE_PHILOX)
    with pytest.raises(ValueError):
        ops.gate_top2(logits, torch.zeros(64, 32, device=DEV), 8, 3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layer_with_device_noise_equals_layer_fed_the_same_noise(dtype):
    """The whole layer (fused two-call path), forward and every gradient: device source == the same layer given the
    stream's Gumbel variates through noise_override, bit for bit; fp32 also against the oracle."""
    M, I, E, B, n_vis, T = 64, 128, 32, 8, 9, 15
    S = B * T
    p = orc.make_params(M, I, E, seed=51)
    h = orc.synth_hidden(B, T, M, seed=52)
    dout = torch.randn(B, T, M, generator=torch.Generator().manual_seed(53))

    def run(device_source):
        moe = build_moe(p, M, I, E, n_vis, device=DEV, dtype=dtype)
        if device_source:
            moe.set_noise_source("device", seed=99)
            moe.moe.device_noise.state[1] = 1000
        else:
            u = torch.from_numpy(device_noise_uniforms(99, 1000, S, E)).to(DEV)
            moe.moe.noise_override = ops.gumbel_from_uniform(u)
        x = h.to(DEV, dtype).requires_grad_(True)
        out, l_aux, cnt = moe(x, mode="vl")
        ((out.float() * dout.to(DEV)).sum() + 0.5 * l_aux).backward()
        torch.cuda.synchronize()
        grads = {n: q.grad.clone() for n, q in moe.named_parameters() if q.grad is not None}
        return moe, out.detach(), l_aux.detach(), cnt.clone(), x.grad.clone(), grads

    moe_a, out_a, aux_a, cnt_a, dx_a, g_a = run(True)
    _, out_b, aux_b, cnt_b, dx_b, g_b = run(False)
    assert _state(moe_a.moe.device_noise.state) == (99, 1000 + S * E // 4)
    assert torch.equal(cnt_a, cnt_b) and torch.equal(out_a, out_b) and torch.equal(aux_a, aux_b) and torch.equal(dx_a, dx_b)
    assert g_a.keys() == g_b.keys()
    for n in g_a:
        assert torch.equal(g_a[n], g_b[n]), n
    if dtype == torch.float32:
        noise = ops.gumbel_from_uniform(torch.from_numpy(device_noise_uniforms(99, 1000, S, E)).to(DEV)).cpu()
        ro, rl, rc = orc.dense_moe_forward(h, p, noise, mode="vl", n_vis=n_vis)
        assert torch.equal(cnt_a, rc)
        np.testing.assert_allclose(out_a.cpu().numpy(), ro.numpy(), rtol=1e-5, atol=1e-6)
    # a second forward draws the next stretch of the stream: other noise, other second choices
    first = moe_a.moe.last_slot.clone()
    with torch.no_grad():
        moe_a(h.to(DEV, dtype), mode="vl")
    assert _state(moe_a.moe.device_noise.state) == (99, 1000 + 2 * (S * E // 4))
    assert not torch.equal(first, moe_a.moe.last_slot)


def test_graph_replays_draw_fresh_noise_without_host_work():
    """A captured step holds no torch generator: every replay advances the device state by one draw, routes
    differently, and equals the eager layer fed that replay's stretch of the stream."""
    M, I, E, B, n_vis, T = 64, 128, 32, 8, 9, 15
    S = B * T
    p = orc.make_params(M, I, E, seed=61)
    h = orc.synth_hidden(B, T, M, seed=62).to(DEV)
    dout = torch.randn(B, T, M, generator=torch.Generator().manual_seed(63)).to(DEV)
    moe = build_moe(p, M, I, E, n_vis, device=DEV)
    moe.set_noise_source("device", seed=5)
    step = GraphedTrainStep(moe, h, dout, mode="vl", aux_weight=0.5)
    per = S * E // 4
    off0 = _state(moe.moe.device_noise.state)[1]
    assert off0 % per == 0 and off0 > 0   # the eager warm-up steps and the capture pass's... draws so far
    slots = []
    for k in range(3):
        out, l_aux = step()
        torch.cuda.synchronize()
        assert _state(moe.moe.device_noise.state)[1] == off0 + (k + 1) * per
        slots.append(moe.moe.last_slot.clone())
        ref = build_moe(p, M, I, E, n_vis, device=DEV)
        u = torch.from_numpy(device_noise_uniforms(5, off0 + k * per, S, E)).to(DEV)
        ref.moe.noise_override = ops.gumbel_from_uniform(u)
        x = h.clone().requires_grad_(True)
        ro, rl, _ = ref(x, mode="vl")
        ((ro * dout).sum() + 0.5 * rl).backward()
        assert torch.equal(out, ro) and torch.equal(l_aux, rl)
        assert torch.equal(moe.moe.gate.wg.weight.grad, ref.moe.gate.wg.weight.grad)
    assert not torch.equal(slots[0], slots[1]) and not torch.equal(slots[1], slots[2])


@pytest.mark.parametrize("offset", [(1 << 32) - 100, (1 << 32) + 7, (1 << 40) + 3, (1 << 63) - 50000])
def test_counter_carries_into_the_high_word(offset):
    """The Philox counter is the 64-bit offset + element / 4 in two 32-bit words: a draw that straddles 2^32 (reached
    after ~35 000 cfg-2 steps) and large offsets must follow the numpy stream, and the advance must carry."""
    S, E = 1000, 32
    logits = torch.randn(S, E, generator=torch.Generator().manual_seed(3)).to(DEV)
    C = ops.capacity(S, E, 1.0, 4)
    src = DeviceNoise(DEV, seed=0x0123456789ABCDEF)
    src.state[1] = offset
    got = ops.gate_top2(logits, src.state, C, ops.NOISE_DEVICE_PHILOX)
    u = torch.from_numpy(device_noise_uniforms(0x0123456789ABCDEF, offset, S, E)).to(DEV)
    want = ops.gate_top2(logits, u, C, True)
    for f in FIELDS:
        assert torch.equal(getattr(got, f), getattr(want, f)), f
    assert _state(src.state)[1] == offset + S * E // 4

"""CPU-side tests: C-ABI surface, host arithmetic, module interface / error behaviour, parameter
packing.  No kernel is launched here (this container has no GPU)."""
import ctypes
import os
import re

import pytest
import torch

import eve_amd
from eve_amd import _lib
from oracle import moe_oracle as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "eve_moe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eve_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared_symbols()
    assert len(names) >= 17
    assert sorted(_lib.SIGNATURES) == names, "eve_amd/_lib.py and include/eve_moe.h disagree"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"libeve_moe.so does not export {n}"
    assert _lib.load().eve_moe_abi_version() == _lib.ABI_VERSION


def test_capacity_matches_reference_arithmetic():
    lib = _lib.load()
    for S in (1, 3, 5, 10, 64, 130, 237, 777, 1896, 15056, 15168, 78976):
        for E in (1, 2, 3, 4, 8, 32, 64):
            for cf in (0.5, 1.0, 1.25, 2.0):
                for mc in (0, 4, 8):
                    assert lib.eve_capacity(S, E, cf, mc) == orc.capacity(S, E, cf, mc), (S, E, cf, mc)
    # config table
    assert lib.eve_capacity(1896, 32, 1.0, 4) == 119 and lib.eve_capacity(15168, 32, 1.0, 4) == 948
    assert lib.eve_capacity(7584, 32, 1.0, 4) == 474 and lib.eve_capacity(78976, 32, 1.0, 4) == 4936


def test_argument_errors_are_reported_without_a_gpu():
    lib = _lib.load()
    rc = lib.eve_ffn_fwd(None, None, None, None, None, None, None, None, None, 1, 1, 1, 8, 8, 0, 0, 0, None)
    assert rc == -1 and b"null" in lib.eve_moe_last_error()
    with pytest.raises(RuntimeError, match="eve_combine_fwd failed"):
        _lib.call("eve_combine_fwd", None, None, None, None, 1, 8, 0, None)


def _moe(M=32, I=64, E=8, **kw):
    return eve_amd.MoE(M, eve_amd.MultiwayTransformerMLP("quick_gelu", M, I), num_experts=E, k=2, **kw)


def test_state_dict_keys_match_the_reference_layout():
    moe = _moe(num_tokens=5)
    keys = list(moe.state_dict().keys())
    assert keys[:3] == ["moe.vision_embed", "moe.text_embed", "moe.gate.wg.weight"]
    for e in range(8):
        for leaf in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
            assert f"moe.experts.experts.{e}.{leaf}" in keys
    assert len(keys) == 3 + 32
    assert moe.moe.gate.wg.weight.dtype == torch.float32 and tuple(moe.moe.gate.wg.weight.shape) == (8, 32)
    assert moe.moe.num_vision_tokens == 5 and moe.num_local_experts == 8
    plain = _moe()
    assert not hasattr(plain.moe, "vision_embed") and plain.moe.num_vision_tokens is None
    # all experts start as identical copies of the template (reference experts.py:15)
    ex = moe.moe.experts.experts
    assert all(torch.equal(ex[0].fc1.weight, ex[e].fc1.weight) for e in range(8))
    # modality embeddings: trunc-normal(std .02) clipped at +-.02
    assert float(moe.moe.vision_embed.abs().max()) <= 0.02


def test_reference_error_behaviour():
    mlp = eve_amd.MultiwayTransformerMLP("quick_gelu", 32, 64)
    with pytest.raises(ValueError, match="top-1 and top-2"):
        eve_amd.MoE(32, mlp, num_experts=8, k=3)
    with pytest.raises(AssertionError, match="divisible"):
        eve_amd.MoE(32, mlp, num_experts=8, ep_size=3, k=2)
    with pytest.raises(AssertionError, match="Unsupported noisy_gate_policy"):
        eve_amd.MoE(32, mlp, num_experts=8, k=2, noisy_gate_policy="bogus")
    with pytest.raises(NotImplementedError):
        eve_amd.MoE(32, torch.nn.Linear(32, 32), num_experts=8, k=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _moe(num_tokens=5)(torch.randn(2, 9, 32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mlp(torch.randn(4, 32))


def test_expert_parameters_alias_packed_buffers_and_repack():
    moe = _moe()
    ex, bank = moe.moe.experts.experts, moe.moe.experts._bank
    assert bank.w1.shape == (8, 64, 32) and bank.w2.shape == (8, 32, 64)
    for e in range(8):
        assert ex[e].fc1.weight.data_ptr() == bank.w1[e].data_ptr()
        assert ex[e].fc2.bias.data_ptr() == bank.b2[e].data_ptr()
    # in-place writes through the per-expert Parameters (optimizer step, load_state_dict, the
    # reference's fix_init_weight reach-in) are visible in the packed buffer
    with torch.no_grad():
        ex[3].fc2.weight.data.div_(2.0)
        ex[5].fc1.bias.add_(1.0)
    assert torch.equal(bank.w2[3], ex[3].fc2.weight) and torch.equal(bank.b1[5], ex[5].fc1.bias)
    sd = {k: torch.randn_like(v) for k, v in moe.state_dict().items()}
    moe.load_state_dict(sd)
    assert torch.equal(bank.w1[7], sd["moe.experts.experts.7.fc1.weight"])
    # .to() replaces every Parameter's storage: the next sync re-packs without changing values
    moe = moe.to(torch.bfloat16)
    old = bank.w1
    moe.moe.experts._bank.sync(moe.moe.experts.experts)
    assert bank.w1 is not old and bank.w1.dtype == torch.bfloat16
    assert torch.equal(bank.w1[7], sd["moe.experts.experts.7.fc1.weight"].to(torch.bfloat16))
    assert ex[7].fc1.weight.data_ptr() == bank.w1[7].data_ptr()
    # a second sync is a no-op
    cur = bank.w1
    moe.moe.experts._bank.sync(moe.moe.experts.experts)
    assert bank.w1 is cur


def test_packed_parameter_gradients_reach_the_leaves():
    moe = _moe(E=4)
    w1, b1, w2, b2 = moe.moe.experts.packed_parameters()
    assert w1.requires_grad and w1.shape == (4, 64, 32)
    (w1.sum() * 2 + (b2 * torch.arange(4.0)[:, None]).sum()).backward()
    ex = moe.moe.experts.experts
    assert all(torch.equal(ex[e].fc1.weight.grad, torch.full((64, 32), 2.0)) for e in range(4))
    assert torch.equal(ex[3].fc2.bias.grad, torch.full((32,), 3.0))
    assert ex[0].fc1.bias.grad is None or float(ex[0].fc1.bias.grad.abs().sum()) == 0.0


def test_activation_probe():
    from eve_amd.experts import activation_code
    from eve_amd import ops
    assert activation_code(eve_amd.QuickGELU()) == ops.ACTS["quick_gelu"]
    assert activation_code(torch.nn.GELU()) == ops.ACTS["gelu"]
    assert activation_code(torch.nn.ReLU()) == ops.ACTS["relu"]
    assert activation_code(lambda x: x * torch.sigmoid(1.702 * x)) == ops.ACTS["quick_gelu"]
    with pytest.raises(NotImplementedError):
        activation_code(torch.tanh)
    try:
        from transformers.activations import ACT2FN
    except Exception:
        return
    assert activation_code(ACT2FN["quick_gelu"]) == ops.ACTS["quick_gelu"]  # what the reference passes


def test_gumbel_rsample_is_the_reference_draw():
    torch.manual_seed(77)
    a = eve_amd.gumbel_rsample((257, 32), torch.device("cpu"))
    import numpy as np
    z = np.load(os.path.join(REPO, "tests", "golden", "noise_seed77.npz"))
    assert np.array_equal(a.numpy(), z["noise"])


def test_fused_layer_buffer_layouts():
    """eve_moe_saved_bytes / eve_moe_saved_offsets / eve_moe_workspace_bytes: host arithmetic, no GPU."""
    lib = _lib.load()
    S, T, M, I, E = 16 * 237, 237, 768, 3072, 32
    C = int(lib.eve_capacity(S, E, 1.0, 4))
    shape = _lib.MoEShape(S, T, 197, _lib.MODE_VL, M, I, E, C, _lib.ACT_QUICK_GELU, _lib.BF16, _lib.BF16)
    ref = ctypes.addressof(shape)
    n = len(_lib.MOE_SAVED_FIELDS)
    sizes = {"gates": S * E * 4, "idx": S * 8, "loc": S * 8, "slot": S * 8, "weight": S * 8, "slot_token": E * C * 4,
             "slot_weight": E * C * 4, "rows": E * 4, "exp_counts": E * 8, "xd": E * C * M * 2, "z": E * C * I * 2,
             "a": E * C * I * 2, "y": E * C * M * 2}
    for nb in (0, 1):
        offs = (ctypes.c_size_t * n)()
        assert lib.eve_moe_saved_offsets(ref, nb, ctypes.addressof(offs), n) == 0
        total = lib.eve_moe_saved_bytes(ref, nb)
        end = 0
        for name, o in zip(_lib.MOE_SAVED_FIELDS, offs):
            if not nb and name in ("xd", "z", "a", "y"):
                assert o == ctypes.c_size_t(-1).value
                continue
            assert o % 256 == 0 and o >= end, name  # aligned, in order, no overlap
            end = o + sizes[name]
        assert end <= total < end + 256
    inf, fwd, bwd = (lib.eve_moe_workspace_bytes(ref, p) for p in
                     (_lib.MOE_PASS_INFERENCE, _lib.MOE_PASS_FORWARD, _lib.MOE_PASS_BACKWARD))
    assert fwd >= S * E * 4 + lib.eve_gate_workspace_bytes(S, E)
    assert inf >= fwd + E * C * (2 * M + I) * 2  # the activation buffers move into the workspace
    assert bwd >= E * C * (2 * M + I) * 2 + S * E * 4 + lib.eve_ffn_bwd_workspace_bytes(E, E, C, M, I, _lib.BF16)
    assert lib.eve_moe_saved_offsets(ref, 1, None, n) == -1 and lib.eve_moe_saved_bytes(None, 1) == 0


def test_philox_reference_known_answers():
    """The numpy Philox4x32-10 of tests/helpers.py (the checker of the gate's device noise source) against the
    known-answer vectors of the Random123 distribution (kat_vectors: philox4x32 10)."""
    import numpy as np
    from tests.helpers import device_noise_uniforms, philox4x32_10

    def run(c, k):
        return [int(x) for x in philox4x32_10(np.array([c], np.uint32), np.array([k], np.uint32))[0]]

    assert run([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    u = device_noise_uniforms(seed=7, offset=5, S=999, E=32)
    assert u.dtype == np.float32 and u.shape == (999, 32) and 0.0 <= u.min() and u.max() < 1.0
    assert abs(float(u.mean()) - 0.5) < 0.01 and abs(float(u.var()) - 1 / 12) < 0.005
    # a later offset continues the same stream: element i at offset o == element i - 4 at offset o + 1
    v = device_noise_uniforms(seed=7, offset=6, S=999, E=32)
    assert np.array_equal(u.reshape(-1)[4:], v.reshape(-1)[:-4])


def test_noise_source_selection_and_device_noise_state():
    """Host side of the gate's device noise source: state layout {seed, offset} as int64, distinct default seeds per
    instance, snapshot / restore, and `set_noise_source` argument checking (no GPU involved)."""
    import torch
    import eve_amd
    from eve_amd import DeviceNoise
    a, b = DeviceNoise("cpu"), DeviceNoise("cpu")
    assert a.state.dtype == torch.int64 and a.state.shape == (2,) and int(a.state[1]) == 0
    assert int(a.state[0]) != int(b.state[0])
    c = DeviceNoise("cpu", seed=(1 << 64) - 1)            # seeds are 64-bit unsigned, stored two's complement
    assert int(c.state[0]) == -1
    snap = c.snapshot()
    c.state[1] = 77
    c.restore(snap)
    assert int(c.state[1]) == 0
    moe = eve_amd.MoE(16, eve_amd.MultiwayTransformerMLP("quick_gelu", 16, 32), num_experts=4, k=2, num_tokens=3)
    assert moe.moe.device_noise is None
    moe.set_noise_source("device", seed=5)
    assert int(moe.moe.device_noise.state[0]) == 5
    moe.set_noise_source("torch")
    assert moe.moe.device_noise is None
    with pytest.raises(ValueError):
        moe.set_noise_source("numpy")

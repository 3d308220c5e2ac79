"""eve_gate_top2_fwd_noalloc (scratch built into the library, no allocation) vs eve_gate_top2_fwd (caller
workspace): every output bit for bit, on calls in a row that share the built-in scratch -- a large input, then a
smaller one (stale scratch of the first call must not leak into the second), then E > 32 (scan + assign path).
Also: a shape beyond the built-in buffer is rejected with a clear error instead of being run."""
import pytest
import torch

from eve_amd import _lib, ops

pytestmark = pytest.mark.gpu

FIELDS = ("gates", "idx", "loc", "slot", "weight", "slot_token", "slot_weight", "exp_counts", "rows", "l_aux")


def _noalloc(logits, noise, C, noise_is_uniform=False):
    S, E = logits.shape
    dev = logits.device
    i32, f32 = dict(dtype=torch.int32, device=dev), dict(dtype=torch.float32, device=dev)
    # outputs are the caller's; poisoned so that an unwritten word shows
    r = ops.Routing(S, E, C, torch.full((S, E), 7.0, **f32), torch.full((S, 2), -7, **i32), torch.full((S, 2), -7, **i32),
                    torch.full((S, 2), -7, **i32), torch.full((S, 2), 7.0, **f32), torch.full((E * C,), -7, **i32),
                    torch.full((E * C,), 7.0, **f32), torch.full((E,), -7, dtype=torch.int64, device=dev),
                    torch.full((E,), -7, **i32), torch.full((1,), 7.0, **f32))
    p = lambda t: t.data_ptr()
    _lib.call("eve_gate_top2_fwd_noalloc", p(logits), p(noise), int(noise_is_uniform), S, E, C, p(r.gates), p(r.idx),
              p(r.loc), p(r.slot), p(r.weight), p(r.slot_token), p(r.slot_weight), p(r.exp_counts), p(r.rows), p(r.l_aux),
              torch.cuda.current_stream().cuda_stream)
    return r


def _inputs(S, E, seed, dev):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(S, E, generator=g) * 1.5).to(dev), torch.rand(S, E, generator=g).to(dev)


def test_noalloc_gate_equals_workspace_gate_on_consecutive_calls():
    dev = torch.device("cuda:0")
    # 78 976 tokens (cfg 5: scan + assign kernels), then 15 168 (cfg 2: fused assign), then smaller still, then E > 32
    for S, E, seed in [(78976, 32, 1), (15168, 32, 2), (1896, 32, 3), (77, 5, 4), (3000, 64, 5), (130, 32, 6)]:
        logits, u = _inputs(S, E, seed, dev)
        C = ops.capacity(S, E, 1.0, 4)
        got = _noalloc(logits, u, C, noise_is_uniform=True)
        want = ops.gate_top2(logits, u, C, noise_is_uniform=True)
        torch.cuda.synchronize()
        for name in FIELDS:
            assert torch.equal(getattr(got, name), getattr(want, name)), (S, E, name)


def test_noalloc_gate_rejects_inputs_beyond_its_buffer():
    dev = torch.device("cuda:0")
    S, E = 200000, 32
    assert _lib.load().eve_gate_workspace_bytes(S, E) > (2 << 20)
    logits, u = _inputs(S, E, 9, dev)
    with pytest.raises(RuntimeError, match="built-in buffer"):
        _noalloc(logits, u, ops.capacity(S, E, 1.0, 4), noise_is_uniform=True)
    # and the entry point still works afterwards
    logits, u = _inputs(300, 32, 10, dev)
    C = ops.capacity(300, 32, 1.0, 4)
    got, want = _noalloc(logits, u, C, True), ops.gate_top2(logits, u, C, True)
    torch.cuda.synchronize()
    for name in FIELDS:
        assert torch.equal(getattr(got, name), getattr(want, name)), name


def test_noalloc_gate_draws_from_a_device_generator_state():
    """noise_is_uniform == 2 (the gate draws its own noise and advances the caller's 16-byte state) through the
    allocation-free entry point: same outputs and same state advance as eve_gate_top2_fwd from the same state."""
    dev = torch.device("cuda:0")
    for S, E in [(15168, 32), (500, 32)]:
        logits, _ = _inputs(S, E, 11, dev)
        C = ops.capacity(S, E, 1.0, 4)
        st_a = torch.tensor([1234, 77], dtype=torch.int64, device=dev)
        st_b = st_a.clone()
        got = _noalloc(logits, st_a, C, noise_is_uniform=ops.NOISE_DEVICE_PHILOX)
        want = ops.gate_top2(logits, st_b, C, ops.NOISE_DEVICE_PHILOX)
        torch.cuda.synchronize()
        for name in FIELDS:
            assert torch.equal(getattr(got, name), getattr(want, name)), (S, name)
        assert torch.equal(st_a, st_b) and int(st_a[1]) == 77 + S * E // 4

"""MI355X-native gate + MoE layer behind the reference's module interface.

Mirrors the public names of the reference's models/sharded_moe.py (`TopKGate`, `MOELayer`,
`top2gating`, `gumbel_rsample`, `_AllToAll`, `_capacity`) so callers and checkpoints drop in,
but none of its computation: the dense one-hot [S,E,C] tensors and einsums
(reference sharded_moe.py:172-235, :367, :377) are replaced by index-form routing produced
by the fused HIP gate kernel and by coalesced row gathers (eve_amd/csrc/*.hip through the
C ABI in include/eve_moe.h).  Everything runs on the GPU; there is no eager fallback.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor, nn

from . import _lib, ops

gumbel_map: Dict[torch.device, Callable] = {}


def trunc_normal_(tensor: Tensor, mean: float = 0.0, std: float = 1.0) -> Tensor:
    """Truncated normal clipped at +-std (reference sharded_moe.py:43-44 via timm)."""
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=-std, b=std)


def _torch_gumbel(shape, device: torch.device) -> Tensor:
    sampler = gumbel_map.get(device)
    if sampler is None:
        sampler = torch.distributions.gumbel.Gumbel(torch.tensor(0.0, device=device),
                                                    torch.tensor(1.0, device=device)).rsample
        gumbel_map[device] = sampler
    return sampler(shape)


_fused_gumbel_ok: Dict[torch.device, bool] = {}


def _fused_gumbel_matches_torch(device: torch.device) -> bool:
    """One-time check per device that the one-pass kernel reproduces torch's sampler bit for bit
    (same generator state in, same fp32 values out); routing parity with the reference rests on it."""
    gen_state = torch.cuda.get_rng_state(device)
    try:
        want = _torch_gumbel((4096,), device)
        torch.cuda.set_rng_state(gen_state, device)
        got = ops.gumbel_from_uniform(torch.rand(4096, dtype=torch.float32, device=device))
        edge = torch.tensor([0.0, 1.0 - 2.0 ** -24, 2.0 ** -24, 0.5], dtype=torch.float32, device=device)
        tiny, eps = torch.finfo(torch.float32).tiny, torch.finfo(torch.float32).eps
        u = tiny + edge * ((1.0 - eps) - tiny)
        want_edge = 0.0 + -1.0 * torch.log(0.0 + -1.0 * torch.log(u))
        return bool(torch.equal(want, got)) and bool(torch.equal(want_edge, ops.gumbel_from_uniform(edge)))
    finally:
        torch.cuda.set_rng_state(gen_state, device)


def _fused_gumbel_usable(device: torch.device) -> bool:
    """True when the in-kernel Gumbel transform may stand in for torch's sampler on `device` (checked once,
    bit for bit, outside graph capture)."""
    ok = _fused_gumbel_ok.get(device)
    if ok is None:
        if torch.cuda.is_current_stream_capturing():
            return False
        ok = _fused_gumbel_ok[device] = _fused_gumbel_matches_torch(device)
        if not ok:
            import warnings
            warnings.warn("eve_amd: fused Gumbel kernel differs from torch.distributions on this build; using torch's")
    return ok


def gumbel_rsample(shape, device: torch.device) -> Tensor:
    """One Gumbel(0,1) draw of `shape` from the GLOBAL torch generator of `device`, value for value the
    draw the reference makes (sharded_moe.py:70-77): the same single `torch.rand` consumption of the
    generator -- so activation-checkpoint RNG replay is unchanged -- followed by the distribution's
    transform chain, which runs here as one HIP kernel instead of ten elementwise ones.  The kernel is
    checked once per device against torch's own sampler; should it ever differ, torch's is used."""
    device = torch.device(device)
    if device.type != "cuda":
        return _torch_gumbel(shape, device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if not _fused_gumbel_usable(device):
        return _torch_gumbel(shape, device)
    return ops.gumbel_from_uniform(torch.rand(tuple(shape), dtype=torch.float32, device=device))


_own_gumbel_rsample = gumbel_rsample


class DeviceNoise:
    """A device-resident Philox4x32-10 generator state {seed, offset} (int64[2]) from which the gate kernel draws the
    uniform values of its Gumbel noise ITSELF (include/eve_moe.h, EVE_NOISE_DEVICE_PHILOX) and which it advances on
    the stream: no noise tensor, no draw kernel, no host work per step -- and, inside a HIP graph, none of the
    generator bookkeeping torch runs before every replay of a graph that holds a `torch.rand` (two fill kernels and
    a launch gap, ~25 us per step at the cfg-2 shape).  Stands for the draw of reference sharded_moe.py:186 where
    torch's generator STREAM is not needed; the values are not torch's, so `torch.manual_seed` does not steer them
    (`snapshot()` / `restore()` do).  Activation-checkpoint recomputes replay the draw: see `for_this_forward`."""

    _instances = 0

    def __init__(self, device, seed: Optional[int] = None) -> None:
        if seed is None:  # a distinct stream per instance, reproducible under torch.manual_seed
            seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (DeviceNoise._instances + 1)) % (1 << 64)
        DeviceNoise._instances += 1
        seed = int(seed) % (1 << 64)
        self.state = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed, 0], dtype=torch.int64, device=device)
        import collections
        self._stash = collections.deque(maxlen=64)  # states that recent forwards started from (see for_this_forward)

    def snapshot(self) -> Tensor:
        return self.state.clone()

    def restore(self, snap: Tensor) -> None:
        self.state.copy_(snap)

    def for_this_forward(self) -> Tensor:
        """The state tensor the gate of the forward being issued should draw from (and advance).  A forward issued
        from INSIDE a backward pass is an activation-checkpoint recompute (reference multiway_transformer.py:574-586,
        fairscale's checkpoint_wrapper, VQA.py:239-244): it must see the noise of the forward it repeats, so every
        ordinary forward leaves a 16-byte copy of the state it started from on a short stack and a recompute draws
        from the most recent one (segments are recomputed in reverse order) without touching the live state.  Under
        HIP-graph capture nothing is stashed (a captured step is never recomputed)."""
        if torch._C._current_graph_task_id() != -1:  # inside autograd's backward: a recompute
            if self._stash:
                return self._stash.pop()
            raise RuntimeError("eve_amd.DeviceNoise: a forward was re-executed inside backward (activation "
                               "checkpointing) but no state of an earlier forward is on record")
        if not torch.cuda.is_current_stream_capturing():
            self._stash.append(self.state.clone())
        return self.state


def _draw_noise(shape, device: torch.device, source: Optional[DeviceNoise] = None) -> Tuple[Tensor, int]:
    """(noise, is_uniform) for the gate kernel: the torch.rand draw itself when the kernel may apply the
    Gumbel transform (the same generator consumption as `gumbel_rsample`), else finished Gumbel variates
    -- also when a caller has replaced this module's `gumbel_rsample`, as the goldens' scripts do.  With a
    `DeviceNoise` source: its state, for the kernel to draw from (nothing is drawn here)."""
    if source is not None:
        if source.state.device != torch.device(device):  # the module was moved after set_noise_source
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("eve_amd: the DeviceNoise state lives on another device than the layer's input")
            source.state = source.state.to(device)
            source._stash.clear()
        return source.for_this_forward(), ops.NOISE_DEVICE_PHILOX
    device = torch.device(device)
    if device.index is None and device.type == "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
    if gumbel_rsample is _own_gumbel_rsample and device.type == "cuda" and _fused_gumbel_usable(device):
        return torch.rand(tuple(shape), dtype=torch.float32, device=device), True
    return gumbel_rsample(shape, device=device), False


def _capacity(num_tokens: int, num_experts: int, capacity_factor: float, min_capacity: int) -> int:
    """Expert queue length for `capacity_factor` ALREADY doubled by the caller, as in the
    reference's `_capacity(gates, tensor(cf * 2), tensor(min_capacity))` (sharded_moe.py:150-160,:177)."""
    return ops.capacity(num_tokens, num_experts, capacity_factor / 2.0, min_capacity)


def _mode_code(mode: Optional[str], has_embed: bool) -> int:
    if not has_embed:
        return _lib.MODE_NONE
    if mode == "vision":
        return _lib.MODE_VISION
    if mode == "text":
        return _lib.MODE_TEXT
    return _lib.MODE_VL  # the reference treats every other value as 'vl' (sharded_moe.py:355)


# ------------------------------------------------------------------ autograd functions
class _AllToAll(torch.autograd.Function):
    """Equal-split all-to-all over the expert-parallel group (RCCL on MI355X); its backward
    is the same exchange applied to the gradient (reference sharded_moe.py:84-100).

    A gloo group has no all-to-all for GPU tensors: there the rows are exchanged through host memory.  That is
    how the expert-parallel path is rehearsed with several ranks on ONE GPU (tests/test_ep_gpu.py,
    `EVE_BENCH_REHEARSAL=1 ... bench.py --gpus 2`); the compute stays on the GPU either way."""

    @staticmethod
    def forward(ctx: Any, group, input: Tensor) -> Tensor:  # type: ignore[override]
        ctx.group = group
        input = input.contiguous()
        if input.is_cuda and dist.get_backend(group) == "gloo":
            host_in = input.cpu()
            host_out = torch.empty_like(host_in)
            dist.all_to_all_single(host_out, host_in, group=group)
            return host_out.to(input.device)
        output = torch.empty_like(input)
        dist.all_to_all_single(output, input, group=group)
        return output

    @staticmethod
    def backward(ctx: Any, grad_output: Tensor):  # type: ignore[override]
        return None, _AllToAll.apply(ctx.group, grad_output)


class _Done:
    """A finished exchange (the host-staged gloo path is synchronous)."""

    def wait(self):
        return True


class _PackedWork:
    """all_to_all_single on packed copies; `wait()` unpacks into the receive views."""

    def __init__(self, work, packed_in, packed_out, recv):
        self.work, self.packed_in, self.packed_out, self.recv = work, packed_in, packed_out, recv

    def wait(self):
        self.work.wait()
        for p, t in enumerate(self.recv):
            t.copy_(self.packed_out[p])
        return True


def _exchange_form(group, on_gpu: bool) -> str:
    """Which collective moves per-peer views, decided from the group's backend ALONE so that every rank of the
    group takes the same branch (a rank that fell back on its own while its peers did not would pair different
    collectives and hang the job):
      'staged'  GPU rows over a gloo group: through host memory (several ranks on ONE GPU, rehearsal only);
      'list'    RCCL (backend 'nccl'): asynchronous list-form all_to_all straight on the views, no packing copy;
      'packed'  anything else (gloo on CPU tensors, ...): all_to_all_single on packed copies."""
    backend = str(dist.get_backend(group)).lower()
    if on_gpu and backend == "gloo":
        return "staged"
    if on_gpu and "nccl" in backend:
        return "list"
    return "packed"


def _exchange_views(group, send, recv):
    """Peer p gets `send[p]` and `recv[p]` is filled from peer p; every tensor is a contiguous view.  Returns a
    work handle: `wait()` makes the current stream wait for the exchange, which runs on RCCL's own stream -- so
    kernels launched on the current stream in the meantime overlap it."""
    form = _exchange_form(group, send[0].is_cuda)
    if form == "staged":
        host_in = torch.stack([t.cpu() for t in send])
        host_out = torch.empty_like(host_in)
        dist.all_to_all_single(host_out, host_in, group=group)
        for p, t in enumerate(recv):
            t.copy_(host_out[p])
        return _Done()
    if form == "list":
        return dist.all_to_all(recv, send, group=group, async_op=True)
    packed_in = torch.stack(send)
    packed_out = torch.empty_like(packed_in)
    return _PackedWork(dist.all_to_all_single(packed_out, packed_in, group=group, async_op=True), packed_in, packed_out, recv)


def ep_chunks(e_local: int) -> int:
    """Pieces the local experts are cut into for the pipelined exchange (EVE_EP_CHUNKS overrides; 1 = one
    exchange each way around one FFN launch sequence, as `_AllToAll` + `Experts.forward` do it).  Two chunks
    while a chunk keeps at least 4 experts.  Smaller GEMM launches cost time: measured at the cfg-2 dims on one
    GPU (`tools/ffn_only.py`), two half launch sequences against one take +0.10 ms per step at 2 x 8 experts
    (ep 2), +0.13 ms at 2 x 4 (ep 4) and +0.14 ms at 2 x 2 (ep 8), and twice as many collectives are launched;
    pipelining hides about half of the exchange time, estimated 0.6 / 0.36 / 0.2-0.3 ms per step at ep 2 / 4 / 8
    -- a clear gain up to ep 4, break-even at ep 8, which therefore keeps the plain sequence."""
    import os
    v = os.environ.get("EVE_EP_CHUNKS")
    n = int(v) if v else (2 if e_local >= 8 else 1)
    n = max(1, min(n, e_local))
    while e_local % n:
        n -= 1
    return n


class _EPPipelineFn(torch.autograd.Function):
    """Expert-parallel exchange -> expert FFNs -> exchange back, pipelined over chunks of the local experts:
    all chunks' dispatch exchanges are posted at once, the FFN of chunk c starts when its rows have arrived
    and runs beside the exchange of chunk c+1, and its result travels back beside the FFN of chunk c+1.  xGMI
    moves the rows (point-to-point links, RCCL's own stream) while the CUs compute; only the first chunk's way
    in and the last chunk's way out are exposed.  Backward is the same pipeline on the gradients.  Same
    arithmetic per row as the unchunked path (the FFN kernels work per expert group), same `[ep, E_local, C, M]`
    layout on both sides (reference sharded_moe.py:368-379)."""

    @staticmethod
    def forward(ctx, group, nch, act, xd4, w1, b1, w2, b2):  # type: ignore[override]
        ep, e_l, C, M = xd4.shape
        xd4 = xd4.contiguous()
        step = e_l // nch
        need_grad = any(ctx.needs_input_grad[3:])
        out4 = torch.empty_like(xd4)
        recv, works = [], []
        for c in range(nch):
            lo = c * step
            buf = torch.empty(ep, step, C, M, dtype=xd4.dtype, device=xd4.device)
            works.append(_exchange_views(group, [xd4[p, lo:lo + step] for p in range(ep)], [buf[p] for p in range(ep)]))
            recv.append(buf)
        saved, back, keep = [], [], []
        for c in range(nch):
            lo = c * step
            works[c].wait()
            x = recv[c].view(ep * step, C, M)
            # RCCL's kernels hold CUs beside these launches: one workgroup per tile (EVE_FFN_SHARED_GPU)
            z, a, y = ops.ffn_fwd(x, w1[lo:lo + step], b1[lo:lo + step], w2[lo:lo + step], b2[lo:lo + step], None,
                                  act, save_z=need_grad, shared_gpu=True)
            y4 = y.view(ep, step, C, M)
            back.append(_exchange_views(group, [y4[p] for p in range(ep)], [out4[p, lo:lo + step] for p in range(ep)]))
            keep.append(y)
            saved += [x, z, a]
        for w in back:
            w.wait()
        if need_grad:
            ctx.save_for_backward(w1, w2, *saved)
            ctx.group, ctx.nch, ctx.act = group, nch, act
        return out4

    @staticmethod
    def backward(ctx, dout4):  # type: ignore[override]
        w1, w2, *saved = ctx.saved_tensors
        group, nch, act = ctx.group, ctx.nch, ctx.act
        dout4 = dout4.contiguous()
        ep, e_l, C, M = dout4.shape
        step = e_l // nch
        I = w1.shape[1]
        dev, dt = dout4.device, dout4.dtype
        recv, works = [], []
        for c in range(nch):
            lo = c * step
            buf = torch.empty(ep, step, C, M, dtype=dt, device=dev)
            works.append(_exchange_views(group, [dout4[p, lo:lo + step] for p in range(ep)], [buf[p] for p in range(ep)]))
            recv.append(buf)
        dxd4 = torch.empty_like(dout4)
        dw1 = torch.empty(e_l, I, M, dtype=dt, device=dev)
        db1 = torch.empty(e_l, I, dtype=dt, device=dev)
        dw2 = torch.empty(e_l, M, I, dtype=dt, device=dev)
        db2 = torch.empty(e_l, M, dtype=dt, device=dev)
        back, keep = [], []
        for c in range(nch):
            lo = c * step
            x, z, a = saved[3 * c:3 * c + 3]
            works[c].wait()
            dx, *_ = ops.ffn_bwd(recv[c].view(ep * step, C, M), x, z, a, w1[lo:lo + step], w2[lo:lo + step], None,
                                 act, grads_out=(dw1[lo:lo + step], db1[lo:lo + step], dw2[lo:lo + step],
                                                 db2[lo:lo + step]), shared_gpu=True)
            dx4 = dx.view(ep, step, C, M)
            back.append(_exchange_views(group, [dx4[p] for p in range(ep)], [dxd4[p, lo:lo + step] for p in range(ep)]))
            keep.append(dx)
        for w in back:
            w.wait()
        return None, None, None, dxd4, dw1, db1, dw2, db2


class _GateFn(torch.autograd.Function):
    """logits, noise -> (weight [S,2], l_aux) + index tensors; HIP eve_gate_top2_fwd/bwd."""

    @staticmethod
    def forward(ctx, logits: Tensor, noise: Tensor, C: int, noise_is_uniform: bool = False):  # type: ignore[override]
        r = ops.gate_top2(logits.contiguous(), noise.contiguous(), C, noise_is_uniform)
        # only what the backward reads, and through save_for_backward: keeping the Routing object (which holds
        # r.weight, an OUTPUT of this Function) on ctx would tie output -> grad_fn -> ctx -> output into a cycle
        # that only Python's cyclic GC frees
        ctx.save_for_backward(r.gates, r.idx, r.loc, r.exp_counts)
        ctx.sec = (r.S, r.E, r.C)
        ctx.set_materialize_grads(False)  # no zero tensors for the index outputs' "gradients"
        outs = (r.weight, r.l_aux.view(()), r.slot, r.slot_token, r.slot_weight, r.rows, r.exp_counts, r.idx, r.loc)
        ctx.mark_non_differentiable(*outs[2:])
        return outs

    @staticmethod
    def backward(ctx, d_weight, d_l_aux, *unused):  # type: ignore[override]
        gates, idx, loc, exp_counts = ctx.saved_tensors
        S, E, C = ctx.sec
        dl = None if d_l_aux is None else d_l_aux.reshape(1).float().contiguous()
        return ops.gate_top2_bwd_raw(gates, idx, loc, exp_counts, S, E, C, d_weight, dl), None, None, None


class _RouteDispatchFn(torch.autograd.Function):
    """h -> (dispatched [E*C,M], weight [S,2], l_aux, routing tensors).

    Fuses reference sharded_moe.py:350-367: modality embedding add, fp32 gate Linear,
    top-2 gating and the dispatch einsum.  The backward produces dh once (dispatch gather
    + gate-Linear input gradient), dwg and the two modality-embedding gradients."""

    @staticmethod
    def forward(ctx, h, wg, vemb, temb, noise, n_vis, mode, cap_factor, min_capacity,  # type: ignore[override]
                noise_is_uniform=False):
        B, T, M = h.shape
        S = B * T
        h2d = h.contiguous().view(S, M)
        wg32 = wg.float().contiguous()
        ve = None if vemb is None else vemb.to(h.dtype).contiguous()
        te = None if temb is None else temb.to(h.dtype).contiguous()
        E = wg32.shape[0]
        logits = ops.gate_logits(h2d, ve, te, wg32, T, n_vis, mode)
        C = ops.capacity(S, E, cap_factor, min_capacity)
        r = ops.gate_top2(logits, noise.contiguous(), C, noise_is_uniform)
        xd = ops.dispatch_fwd(h2d, ve, te, r.slot_token, T, n_vis, mode, slot=r.slot)
        ctx.dims, ctx.sec = (B, T, M, n_vis, mode), (r.S, r.E, r.C)
        ctx.set_materialize_grads(False)  # no zero tensors for the index outputs' "gradients"
        # (not the Routing object: it holds r.weight, an output of this Function -- a reference cycle through ctx)
        ctx.save_for_backward(h2d, wg32, ve, te, r.gates, r.idx, r.loc, r.exp_counts, r.slot)
        ctx.param_dtypes = (wg.dtype, None if vemb is None else vemb.dtype, None if temb is None else temb.dtype)
        outs = (xd, r.weight, r.l_aux.view(()), r.slot, r.slot_token, r.slot_weight, r.rows, r.exp_counts)
        ctx.mark_non_differentiable(*outs[3:])
        return outs

    @staticmethod
    def backward(ctx, dxd, d_weight, d_l_aux, *unused):  # type: ignore[override]
        B, T, M, n_vis, mode = ctx.dims
        h2d, wg32, ve, te, gates, idx, loc, exp_counts, slot = ctx.saved_tensors
        S, E, C = ctx.sec
        dl = None if d_l_aux is None else d_l_aux.reshape(1).float().contiguous()
        dlogits = ops.gate_top2_bwd_raw(gates, idx, loc, exp_counts, S, E, C, d_weight, dl)
        if dxd is None:
            dxd = torch.zeros(E * C, M, dtype=h2d.dtype, device=h2d.device)
        wg_dt, v_dt, t_dt = ctx.param_dtypes
        want_v = mode in (_lib.MODE_VL, _lib.MODE_VISION) and ctx.needs_input_grad[2]
        want_t = mode in (_lib.MODE_VL, _lib.MODE_TEXT) and ctx.needs_input_grad[3]
        e_dt = v_dt if v_dt is not None else t_dt
        fused_dt = e_dt if e_dt in (torch.float32, torch.bfloat16) and v_dt in (None, e_dt) and t_dt in (None, e_dt) \
            else torch.float32
        dh, dv, dt = ops.route_bwd(dxd.contiguous(), slot, dlogits, wg32, S, T, n_vis, mode, want_v, want_t, fused_dt)
        dwg = ops.gate_wgrad(h2d, ve, te, dlogits, T, n_vis, mode).to(wg_dt) if ctx.needs_input_grad[1] else None
        if dv is not None and dv.dtype != v_dt:
            dv = dv.to(v_dt)
        if dt is not None and dt.dtype != t_dt:
            dt = dt.to(t_dt)
        return dh.view(B, T, M), dwg, dv, dt, None, None, None, None, None, None


class _CombineFn(torch.autograd.Function):
    """out[s] = w1*y[slot1] + w2*y[slot2]  (reference einsum 'sec,ecm->sm', sharded_moe.py:377)."""

    @staticmethod
    def forward(ctx, y, weight, slot, slot_token, slot_weight):  # type: ignore[override]
        y = y.contiguous()
        ctx.save_for_backward(y, slot, slot_token, slot_weight)
        return ops.combine_fwd(y, slot, weight.contiguous())

    @staticmethod
    def backward(ctx, dout):  # type: ignore[override]
        y, slot, slot_token, slot_weight = ctx.saved_tensors
        dy, dw = ops.combine_bwd(dout.contiguous(), y, slot, slot_token, slot_weight)
        return dy, dw, None, None, None


class _DispatchFn(torch.autograd.Function):
    """x [S,M] (modality embedding already added) -> dispatched [E*C,M]: the einsum 'sec,sm->ecm' of reference
    sharded_moe.py:367 on its own, for the path whose gate input differs from what is dispatched (Jitter)."""

    @staticmethod
    def forward(ctx, x2d, slot, slot_token):  # type: ignore[override]
        ctx.save_for_backward(slot)
        ctx.S = x2d.shape[0]
        return ops.dispatch_fwd(x2d.contiguous(), None, None, slot_token, 1, 0, _lib.MODE_NONE, slot=slot)

    @staticmethod
    def backward(ctx, dxd):  # type: ignore[override]
        (slot,) = ctx.saved_tensors
        return ops.dispatch_bwd(dxd.contiguous(), slot, None, None, ctx.S), None, None


def _cast(t: Optional[Tensor], dtype: torch.dtype) -> Optional[Tensor]:
    """`t.to(dtype).contiguous()` of a (small) parameter, made anew on every call like the per-stage path makes it.
    Round 2 kept the copy per (tensor, version counter, storage pointer); an update through `.data` -- what older
    `transformers.AdamW` builds and fairscale's OSS do -- changes neither, so the fused path would have gone on using
    stale values.  The cast is a [M]-sized kernel and only exists when parameter and activation dtypes differ."""
    if t is None:
        return None
    if t.dtype == dtype:
        return t if t.is_contiguous() else t.contiguous()
    return t.detach().to(dtype).contiguous()


_fused_plans: Dict[tuple, tuple] = {}


def _fused_plan(key: tuple):
    """(shape struct, {pass: workspace bytes}, {need_backward: (saved bytes, offsets)}) of one layer shape."""
    plan = _fused_plans.get(key)
    if plan is None:
        import ctypes
        lib = _lib.load()
        shape = _lib.MoEShape(*key)
        ref = ctypes.addressof(shape)
        ws = {p: int(lib.eve_moe_workspace_bytes(ref, p)) for p in
              (_lib.MOE_PASS_INFERENCE, _lib.MOE_PASS_FORWARD, _lib.MOE_PASS_BACKWARD, _lib.MOE_PASS_ROUTE_FWD,
               _lib.MOE_PASS_ROUTE_BWD)}
        saved = {}
        n = len(_lib.MOE_SAVED_FIELDS)
        for nb in (0, 1):
            offs = (ctypes.c_size_t * n)()
            _lib.call("eve_moe_saved_offsets", ref, nb, ctypes.addressof(offs), n)
            saved[nb] = (int(lib.eve_moe_saved_bytes(ref, nb)), dict(zip(_lib.MOE_SAVED_FIELDS, list(offs))))
        plan = _fused_plans[key] = (shape, ref, ws, saved)
    return plan


class _FusedMoEFn(torch.autograd.Function):
    """The whole single-rank layer, one C call per direction (eve_moe_fwd / eve_moe_bwd): reference
    sharded_moe.py:345-381 with every expert local.  Same kernels, order and results as the per-stage Functions
    above (`_RouteDispatchFn` -> `_GroupedFFNFn` -> `_CombineFn`); the intermediates live in one `saved` buffer
    and one workspace per call instead of about twenty separately allocated tensors."""

    @staticmethod
    def forward(ctx, h, wg, vemb, temb, noise, w1, b1, w2, b2, n_vis, mode, cap_factor, min_capacity,  # type: ignore[override]
                noise_is_uniform, act, grad_mode, counts_host=None):
        B, T, M = h.shape
        S = B * T
        h2d = h.contiguous().view(S, M)
        wg32 = _cast(wg, torch.float32)
        ve = _cast(vemb, h.dtype)
        te = _cast(temb, h.dtype)
        noise = noise.contiguous()
        E, I = wg32.shape[0], w1.shape[1]
        for t, n in ((w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2"), (ve, "vision_embed"), (te, "text_embed")):
            ops._chk(t, n, h.dtype)
        ops._chk(h2d, "hidden"), ops._chk(wg32, "wg", torch.float32)
        ncode = ops.noise_code(noise, noise_is_uniform, S, E)
        if w1.shape != (E, I, M) or w2.shape != (E, M, I) or b1.shape != (E, I) or b2.shape != (E, M) \
                or wg32.shape[1] != M:
            raise RuntimeError("eve_amd: expert / gate parameter shapes do not match the layer input")
        C = ops.capacity(S, E, cap_factor, min_capacity)
        v_dt = None if vemb is None else vemb.dtype
        t_dt = None if temb is None else temb.dtype
        e_dt = v_dt if v_dt is not None else t_dt
        fused_dt = e_dt if e_dt in (torch.float32, torch.bfloat16) and v_dt in (None, e_dt) and t_dt in (None, e_dt) \
            else torch.float32
        need_bwd = bool(grad_mode and any(ctx.needs_input_grad))
        key = (S, T, n_vis, mode, M, I, E, C, act, ops._dt(h2d), ops._DTYPES[fused_dt])
        _, ref, ws_bytes, saved_plan = _fused_plan(key)
        nbytes, offs = saved_plan[1 if need_bwd else 0]
        dev = h.device
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        nws = ws_bytes[_lib.MOE_PASS_FORWARD if need_bwd else _lib.MOE_PASS_INFERENCE]
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        out = torch.empty(B, T, M, dtype=h.dtype, device=dev)
        l_aux = torch.empty((), dtype=torch.float32, device=dev)
        ops.PROFILE.call("eve_moe_fwd", ref, ops._p(h2d), ops._p(ve), ops._p(te), ops._p(wg32), ops._p(noise),
                         ncode, ops._p(w1), ops._p(b1), ops._p(w2), ops._p(b2), ops._p(out),
                         ops._p(l_aux), ops._p(counts_host), ops._p(blob), nbytes, 1 if need_bwd else 0, ops._p(ws), nws,
                         ops._stream())
        o = offs["slot"]
        slot = blob[o:o + S * 8].view(torch.int32).view(S, 2)
        o = offs["exp_counts"]
        exp_counts = blob[o:o + E * 8].view(torch.int64)
        if need_bwd:
            ctx.save_for_backward(h2d, wg32, ve, te, w1, w2, blob)
            ctx.key, ctx.dims = key, (B, T, M, E, I)
            ctx.param_dtypes = (wg.dtype, v_dt, t_dt, fused_dt)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(slot, exp_counts)
        return out, l_aux, slot, exp_counts

    @staticmethod
    def backward(ctx, dout, d_l_aux, *unused):  # type: ignore[override]
        h2d, wg32, ve, te, w1, w2, blob = ctx.saved_tensors
        B, T, M, E, I = ctx.dims
        S = B * T
        mode = ctx.key[3]
        _, ref, ws_bytes, _ = _fused_plan(ctx.key)
        dev, dt = h2d.device, h2d.dtype
        wg_dt, v_dt, t_dt, fused_dt = ctx.param_dtypes
        dout = torch.zeros(S, M, dtype=dt, device=dev) if dout is None else dout.contiguous()
        ops._chk(dout, "dout", dt)
        dl = None if d_l_aux is None else d_l_aux.reshape(1).float().contiguous()
        need = ctx.needs_input_grad
        want_v = mode in (_lib.MODE_VL, _lib.MODE_VISION) and need[2]
        want_t = mode in (_lib.MODE_VL, _lib.MODE_TEXT) and need[3]
        dh = torch.empty(B, T, M, dtype=dt, device=dev)
        dwg = torch.empty(E, M, dtype=torch.float32, device=dev) if need[1] else None
        dv = torch.empty(M, dtype=fused_dt, device=dev) if want_v else None
        dte = torch.empty(M, dtype=fused_dt, device=dev) if want_t else None
        dw1 = torch.empty(E, I, M, dtype=dt, device=dev)
        db1 = torch.empty(E, I, dtype=dt, device=dev)
        dw2 = torch.empty(E, M, I, dtype=dt, device=dev)
        db2 = torch.empty(E, M, dtype=dt, device=dev)
        nws = ws_bytes[_lib.MOE_PASS_BACKWARD]
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        ops.PROFILE.call("eve_moe_bwd", ref, ops._p(dout), ops._p(dl), ops._p(h2d), ops._p(ve), ops._p(te), ops._p(wg32),
                         ops._p(w1), ops._p(w2), ops._p(blob), blob.numel(), ops._p(dh), ops._p(dwg), ops._p(dv),
                         ops._p(dte), ops._p(dw1), ops._p(db1), ops._p(dw2), ops._p(db2), ops._p(ws), nws, ops._stream())
        if dwg is not None and dwg.dtype != wg_dt:
            dwg = dwg.to(wg_dt)
        if dv is not None and dv.dtype != v_dt:
            dv = dv.to(v_dt)
        if dte is not None and dte.dtype != t_dt:
            dte = dte.to(t_dt)
        return (dh, dwg, dv, dte, None, dw1, db1, dw2, db2) + (None,) * 8



class _EPRouteFn(torch.autograd.Function):
    """Everything of the layer BEFORE the first exchange, one C call per direction (eve_moe_route_fwd / _bwd):
    modality add + gate Linear -> top-2 gating -> dispatch forward; gate backward -> dispatch backward + gate weight
    gradient + modality-embedding gradients backward.  Same kernels and results as `_RouteDispatchFn` (reference
    sharded_moe.py:350-367).  Returns the dispatched rows, the combine weights (a view into the routing blob, the
    differentiable link to `_EPFinishFn`), l_aux and the views the module exposes; the routing blob itself is left in
    `holder["blob"]`."""

    @staticmethod
    def forward(ctx, h, wg, vemb, temb, noise, key, noise_is_uniform, holder):  # type: ignore[override]
        B, T, M = h.shape
        S, E, C = key[0], key[6], key[7]
        h2d = h.contiguous().view(S, M)
        wg32 = _cast(wg, torch.float32)
        ve, te = _cast(vemb, h.dtype), _cast(temb, h.dtype)
        noise = noise.contiguous()
        ops._chk(h2d, "hidden"), ops._chk(wg32, "wg", torch.float32)
        ncode = ops.noise_code(noise, noise_is_uniform, S, E)
        ops._chk(ve, "vision_embed", h.dtype), ops._chk(te, "text_embed", h.dtype)
        if tuple(wg32.shape) != (E, M):
            raise RuntimeError("eve_amd: gate parameter shapes do not match the layer input")
        v_dt = None if vemb is None else vemb.dtype
        t_dt = None if temb is None else temb.dtype
        fused_dt = torch.bfloat16 if key[10] == _lib.BF16 else torch.float32
        _, ref, ws_bytes, saved_plan = _fused_plan(key)
        nbytes, offs = saved_plan[0]
        dev = h.device
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        nws = ws_bytes[_lib.MOE_PASS_ROUTE_FWD]
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        xd = torch.empty(E * C, M, dtype=h.dtype, device=dev)
        l_aux = torch.empty((), dtype=torch.float32, device=dev)
        ops.PROFILE.call("eve_moe_route_fwd", ref, ops._p(h2d), ops._p(ve), ops._p(te), ops._p(wg32), ops._p(noise),
                         ncode, ops._p(xd), ops._p(l_aux), ops._p(blob), nbytes, ops._p(ws), nws,
                         ops._stream())
        view = lambda name, n, dt: blob[offs[name]:offs[name] + n].view(dt)
        weight = view("weight", S * 8, torch.float32).view(S, 2)
        slot = view("slot", S * 8, torch.int32).view(S, 2)
        exp_counts = view("exp_counts", E * 8, torch.int64)
        ctx.save_for_backward(h2d, wg32, ve, te, blob)
        ctx.key, ctx.dims = key, (B, T, M, E)
        ctx.param_dtypes = (wg.dtype, v_dt, t_dt, fused_dt)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(slot, exp_counts)
        holder["blob"] = blob  # (not an output: `weight`, `slot` and `exp_counts` are views into it)
        return xd, weight, l_aux, slot, exp_counts

    @staticmethod
    def backward(ctx, dxd, d_weight, d_l_aux, *unused):  # type: ignore[override]
        h2d, wg32, ve, te, blob = ctx.saved_tensors
        B, T, M, E = ctx.dims
        S = B * T
        mode, C = ctx.key[3], ctx.key[7]
        _, ref, ws_bytes, _ = _fused_plan(ctx.key)
        dev, dt = h2d.device, h2d.dtype
        wg_dt, v_dt, t_dt, fused_dt = ctx.param_dtypes
        dxd = torch.zeros(E * C, M, dtype=dt, device=dev) if dxd is None else dxd.contiguous()
        ops._chk(dxd, "dxd", dt)
        if d_weight is not None:
            d_weight = d_weight.contiguous()
        ops._chk(d_weight, "d_weight", torch.float32)
        dl = None if d_l_aux is None else d_l_aux.reshape(1).float().contiguous()
        need = ctx.needs_input_grad
        want_v = mode in (_lib.MODE_VL, _lib.MODE_VISION) and need[2]
        want_t = mode in (_lib.MODE_VL, _lib.MODE_TEXT) and need[3]
        dh = torch.empty(B, T, M, dtype=dt, device=dev)
        dwg = torch.empty(E, M, dtype=torch.float32, device=dev) if need[1] else None
        dv = torch.empty(M, dtype=fused_dt, device=dev) if want_v else None
        dte = torch.empty(M, dtype=fused_dt, device=dev) if want_t else None
        nws = ws_bytes[_lib.MOE_PASS_ROUTE_BWD]
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        ops.PROFILE.call("eve_moe_route_bwd", ref, ops._p(dxd), ops._p(d_weight), ops._p(dl), ops._p(h2d), ops._p(ve),
                         ops._p(te), ops._p(wg32), ops._p(blob), blob.numel(), ops._p(dh), ops._p(dwg), ops._p(dv),
                         ops._p(dte), ops._p(ws), nws, ops._stream())
        if dwg is not None and dwg.dtype != wg_dt:
            dwg = dwg.to(wg_dt)
        if dv is not None and dv.dtype != v_dt:
            dv = dv.to(v_dt)
        if dte is not None and dte.dtype != t_dt:
            dte = dte.to(t_dt)
        return dh, dwg, dv, dte, None, None, None, None


class _EPFinishFn(torch.autograd.Function):
    """Everything AFTER the second exchange: the combine (reference sharded_moe.py:377) on the routing blob of
    `_EPRouteFn`, one C call per direction (eve_moe_finish_fwd / _bwd).  `weight` is the blob's view: it carries the
    gradient of the combine weights back to the gate."""

    @staticmethod
    def forward(ctx, y, weight, blob, key):  # type: ignore[override]
        S, M = key[0], key[4]
        y = y.contiguous()
        _, ref, _, _ = _fused_plan(key)
        out = torch.empty(S, M, dtype=y.dtype, device=y.device)
        ops._chk(y, "expert_output")
        ops.PROFILE.call("eve_moe_finish_fwd", ref, ops._p(y), ops._p(blob), blob.numel(), ops._p(out), ops._stream())
        ctx.save_for_backward(y, blob)
        ctx.key = key
        return out

    @staticmethod
    def backward(ctx, dout):  # type: ignore[override]
        y, blob = ctx.saved_tensors
        S = ctx.key[0]
        _, ref, _, _ = _fused_plan(ctx.key)
        dout = dout.contiguous().view(S, -1)
        ops._chk(dout, "dout", y.dtype)
        dy = torch.empty_like(y)
        dw = torch.empty(S, 2, dtype=torch.float32, device=y.device)
        ops.PROFILE.call("eve_moe_finish_bwd", ref, ops._p(dout), ops._p(y), ops._p(blob), blob.numel(), ops._p(dy),
                         ops._p(dw), ops._stream())
        return dy, dw, None, None


# -------------------------------------------------------------------------- functional
def top2gating(logits: Tensor, capacity_factor: float, min_capacity: int,
               noise: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Reference-compatible top-2 gating (sharded_moe.py:172-235): returns
    (l_aux, combine_weights [S,E,C], dispatch_mask [S,E,C], exp_counts [E] int64 on CPU).

    Routing itself is computed by the fused HIP gate kernel; the dense tensors are only
    materialised here, by one scatter, for callers that still want them.  `MOELayer` never
    builds them.  `noise` overrides the Gumbel draw of :186 (tests)."""
    S, E = logits.shape
    logits = logits.float()
    is_uniform = False
    if noise is None:
        noise, is_uniform = _draw_noise(logits.shape, logits.device)
    C = ops.capacity(S, E, capacity_factor, min_capacity)
    weight, l_aux, slot, _, _, _, exp_counts, _, _ = _GateFn.apply(logits, noise, C, is_uniform)
    # a token's two slots are distinct (different experts), dropped choices (slot -1) go to a dummy last column:
    # no index appears twice in a row, so the scatter is deterministic and slot 0 can never be overwritten by a
    # dropped choice's zero
    keep = slot >= 0
    col = torch.where(keep, slot, torch.full_like(slot, E * C)).long()
    combine = torch.zeros(S, E * C + 1, dtype=torch.float32, device=logits.device)
    combine = combine.scatter(1, col, weight * keep)[:, :E * C]
    combine = combine.reshape(S, E, C)
    return l_aux, combine, combine.bool(), exp_counts.to("cpu")


class TopKGate(nn.Module):
    """Gate module, constructor and attributes as the reference's (sharded_moe.py:238-297).
    `k` is validated but, as in the reference (:294), gating is always top-2."""

    wg: nn.Linear

    def __init__(self, model_dim: int, num_experts: int, k: int = 1, capacity_factor: float = 1.0,
                 eval_capacity_factor: float = 1.0, min_capacity: int = 8, noisy_gate_policy: Optional[str] = None,
                 drop_tokens: bool = True, use_rts: bool = True) -> None:
        super().__init__()
        if k != 1 and k != 2:
            raise ValueError('Only top-1 and top-2 gatings are supported.')
        self.wg = nn.Linear(model_dim, num_experts, bias=False).float()
        self.k = k
        self.model_dim = model_dim
        self.num_experts = num_experts
        self.capacity_factor = capacity_factor
        self.eval_capacity_factor = eval_capacity_factor
        self.min_capacity = min_capacity
        self.noisy_gate_policy = noisy_gate_policy
        self.wall_clock_breakdown = False
        self.gate_time = 0.0
        self.drop_tokens = drop_tokens
        self.use_rts = use_rts

    def active_capacity_factor(self) -> float:
        return self.capacity_factor if self.training else self.eval_capacity_factor

    def forward(self, input: Tensor, noise: Optional[Tensor] = None):
        """[S,M] -> (l_aux, combine_weights, dispatch_mask, exp_counts), reference :283-297."""
        if self.wg.weight.dtype != torch.float32:
            self.wg = self.wg.float()
        x = input
        if self.noisy_gate_policy == 'Jitter' and self.training:
            x = multiplicative_jitter(x.float(), device=x.device)
        x = x.contiguous()
        logits = _LogitsFn.apply(x, self.wg.weight)
        return top2gating(logits, self.active_capacity_factor(), self.min_capacity, noise=noise)


class _LogitsFn(torch.autograd.Function):
    """x [S,M] -> fp32 logits [S,E] with the HIP gate-logits kernel (no modality add)."""

    @staticmethod
    def forward(ctx, x, wg):  # type: ignore[override]
        wg32 = wg.float().contiguous()
        ctx.save_for_backward(x, wg32)
        ctx.wg_dtype = wg.dtype
        return ops.gate_logits(x, None, None, wg32, 1, 0, _lib.MODE_NONE)

    @staticmethod
    def backward(ctx, dlogits):  # type: ignore[override]
        x, wg32 = ctx.saved_tensors
        dlogits = dlogits.contiguous()
        S, M = x.shape
        no_slots = torch.full((S, 2), -1, dtype=torch.int32, device=x.device)
        dx = ops.dispatch_bwd(torch.zeros(1, M, dtype=x.dtype, device=x.device), no_slots, dlogits, wg32, S)
        dwg = ops.gate_wgrad(x, None, None, dlogits, 1, 0, _lib.MODE_NONE).to(ctx.wg_dtype)
        return dx, dwg


def multiplicative_jitter(x: Tensor, device: torch.device, epsilon: float = 1e-2) -> Tensor:
    """Scales every element of `x` by an independent draw from U[1 - epsilon, 1 + epsilon) (what the reference's
    gate applies to its input under noisy_gate_policy='Jitter', sharded_moe.py:46-67; EVE never sets that policy).
    One `torch.rand` draw of x's shape from the global generator of `device`."""
    if epsilon == 0:
        return x
    u = torch.rand(x.shape, dtype=x.dtype if x.is_floating_point() else torch.float32, device=torch.device(device))
    return x * (u * (2.0 * epsilon) + (1.0 - epsilon))


class MOELayer(nn.Module):
    """Modality-aware MoE layer with the reference's constructor, attributes and forward
    contract (sharded_moe.py:300-381): `forward(h, mode='vl')` returns the combined output
    and leaves `l_aux` / `exp_counts` on the module."""

    def __init__(self, gate: nn.Module, experts: nn.Module, ep_group_name, ep_size, num_local_experts: int,
                 num_tokens=None) -> None:
        super().__init__()
        self.gate = gate
        self.experts = experts
        self.ep_group = None
        self.ep_size = ep_size
        self.ep_group_name = ep_group_name
        self.num_local_experts = num_local_experts
        self.time_falltoall = 0.0
        self.time_salltoall = 0.0
        self.time_moe = 0.0
        self.wall_clock_breakdown = False
        # exp_counts is handed back as a CPU int64 tensor like the reference's (:199), by default with the
        # reference's blocking `.to('cpu')`: valid the moment forward returns.  Opt-in False: the copy goes
        # asynchronously into a pinned buffer (no host sync per layer) and is valid once the stream has been
        # synchronised.  Under HIP-graph capture the copy node targets a pinned buffer owned by THIS layer.
        self.blocking_exp_counts = True
        # ep_size == 1: issue the layer as one C call per direction (eve_moe_fwd / eve_moe_bwd).  False = one call
        # per stage (same kernels and results; what the per-call HIP-event profile of bench.py times).
        self.fused_call = True
        # the two halves around the exchange as one C call each per direction (eve_moe_route_* / eve_moe_finish_*): what
        # an expert-parallel layer runs; True on a single rank selects them too (the exchange is then the identity;
        # tests compare the three call forms bit for bit)
        self.split_call = False
        self._graph_counts_host: Optional[Tensor] = None
        self.l_aux = None
        self.exp_counts = None
        self.noise_override: Optional[Tensor] = None  # tests: replaces the Gumbel draw once
        # `last_slot` ([S,2] token -> slot map of the latest forward, diagnostics) is a COPY on the fused paths (their
        # map lives in the saved-state blob); False skips that copy (one ~5 us copy node of a captured step)
        self.track_last_slot = True
        import collections
        self._counts_in_flight = collections.deque()  # (event, pinned exp_counts tensor) of calls the stream has not passed
        self.device_noise: Optional[DeviceNoise] = None  # set_noise_source("device"): the gate draws its own noise

        self.num_vision_tokens = None
        if num_tokens:
            self.vision_embed = nn.Parameter(torch.zeros(self.gate.model_dim))
            trunc_normal_(self.vision_embed, std=0.02)
            self.text_embed = nn.Parameter(torch.zeros(self.gate.model_dim))
            trunc_normal_(self.text_embed, std=0.02)
            self.num_vision_tokens = num_tokens

    def set_noise_source(self, source: str = "torch", seed: Optional[int] = None, device=None) -> None:
        """"torch" (default): the Gumbel draw consumes torch's device generator exactly as the reference's
        `gumbel_rsample` does (sharded_moe.py:70-77).  "device": the gate kernel draws from a `DeviceNoise` state of
        this layer (see there for what that changes)."""
        if source == "torch":
            self.device_noise = None
        elif source == "device":
            dev = device if device is not None else self.gate.wg.weight.device
            self.device_noise = DeviceNoise(dev, seed)
        else:
            raise ValueError("eve_amd.MOELayer.set_noise_source: 'torch' or 'device'")

    def _pinned_counts(self, E: int) -> Tensor:
        """The pinned host tensor a non-blocking forward returns exp_counts in: this layer's own persistent one under
        HIP-graph capture (refilled by every replay), a fresh one per eager call (as `_host_counts` hands out)."""
        if torch.cuda.is_current_stream_capturing():
            if self._graph_counts_host is None or self._graph_counts_host.numel() != E:
                raise RuntimeError("eve_amd: run the layer once eagerly before capturing it into a graph")
            return self._graph_counts_host
        if self._graph_counts_host is None or self._graph_counts_host.numel() != E:
            self._graph_counts_host = torch.empty(E, dtype=torch.int64, pin_memory=True)
        # The gate kernel writes through the raw pointer, which torch's pinned-memory allocator does not know about
        # (a `copy_(non_blocking=True)` would have recorded the stream on the block): the layer keeps every buffer it
        # hands out alive until the stream has passed the call that fills it, so that a caller who drops the tensor
        # early cannot have its block recycled under a write that is still queued.
        while self._counts_in_flight and self._counts_in_flight[0][0].query():
            self._counts_in_flight.popleft()
        return torch.empty(E, dtype=torch.int64, pin_memory=True)

    def _counts_issued(self, counts_host: Tensor) -> None:
        if not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record()
            self._counts_in_flight.append((ev, counts_host))

    def _host_counts(self, exp_counts: Tensor) -> Tensor:
        if torch.cuda.is_current_stream_capturing():
            # under HIP-graph capture nothing may be allocated or synchronised: the copy node targets this layer's
            # own pinned buffer, refilled by every replay (two layers captured in one graph keep separate counts)
            if self._graph_counts_host is None:
                raise RuntimeError("eve_amd: run the layer once eagerly before capturing it into a graph")
            self._graph_counts_host.copy_(exp_counts, non_blocking=True)
            return self._graph_counts_host
        if self._graph_counts_host is None or self._graph_counts_host.shape != exp_counts.shape:
            self._graph_counts_host = torch.empty(exp_counts.shape, dtype=exp_counts.dtype, pin_memory=True)
        if self.blocking_exp_counts:
            return exp_counts.to("cpu")
        host = torch.empty(exp_counts.shape, dtype=exp_counts.dtype, pin_memory=True)
        host.copy_(exp_counts, non_blocking=True)
        return host

    def _set_ep_group(self, ep_group) -> None:
        self.ep_group = ep_group

    def _resolve_ep_group(self):
        if self.ep_size == 1:
            return None
        if self.ep_group is None:
            if not dist.is_initialized():
                raise RuntimeError("eve_amd.MOELayer: ep_size > 1 needs torch.distributed to be initialised")
            if dist.get_world_size() != self.ep_size:
                raise RuntimeError("eve_amd.MOELayer: ep_size != world size; pass the expert-parallel process group "
                                   "with MOELayer._set_ep_group(group)")
            self.ep_group = dist.group.WORLD
        return self.ep_group

    def _forward_halves(self, h, orig_shape, vemb_in, temb_in, noise, is_uniform, n_vis, code, E):
        """The expert-parallel form of the layer: route half -> exchange -> expert FFNs -> exchange -> finish half
        (reference sharded_moe.py:345-381 with the `_AllToAll`s of :84-100 live).  Six foreign calls per step plus
        the collectives; peers' fill counts are not exchanged, so every capacity row received is processed."""
        from .experts import _GroupedFFNFn
        gate = self.gate
        B, T, M = h.shape
        S = B * T
        C = ops.capacity(S, E, gate.active_capacity_factor(), gate.min_capacity)
        v_dt = None if vemb_in is None else vemb_in.dtype
        t_dt = None if temb_in is None else temb_in.dtype
        e_dt = v_dt if v_dt is not None else t_dt
        fused_dt = e_dt if e_dt in (torch.float32, torch.bfloat16) and v_dt in (None, e_dt) and t_dt in (None, e_dt) \
            else torch.float32
        act = self.experts._act_code
        # (the halves never read I: the plan is keyed with I = M)
        key = (S, T, n_vis, code, M, M, E, C, act, ops._dt(h), ops._DTYPES[fused_dt])
        holder: Dict[str, Tensor] = {}
        xd, weight, l_aux, slot, exp_counts = _EPRouteFn.apply(h, gate.wg.weight, vemb_in, temb_in, noise, key, is_uniform,
                                                               holder)
        blob = holder["blob"]
        self.l_aux = l_aux
        self.exp_counts = self._host_counts(exp_counts)
        self.last_slot = slot.clone() if self.track_last_slot else None
        group = self._resolve_ep_group()
        w1, b1, w2, b2 = self.experts.packed_parameters(xd.dtype)
        ep, e_l = self.ep_size, self.num_local_experts
        xd4 = xd.view(ep, e_l, C, M)
        nch = ep_chunks(e_l) if group is not None else 1
        if group is not None and nch > 1:  # pipelined exchange / FFN / exchange over chunks of the local experts
            y = _EPPipelineFn.apply(group, nch, act, xd4, w1, b1, w2, b2)
        else:
            if group is not None:
                x_in = _AllToAll.apply(group, xd4.view(ep, -1)).view(ep * e_l, C, M)
                rows = None
            else:  # a single rank (split_call): the routing blob's fill counts bound the rows the GEMMs touch
                x_in = xd4.view(ep * e_l, C, M)
                o = _fused_plan(key)[3][0][1]["rows"]
                rows = blob[o:o + E * 4].view(torch.int32)
            y = _GroupedFFNFn.apply(x_in, w1, b1, w2, b2, rows, act)
            if group is not None:
                y = _AllToAll.apply(group, y.reshape(ep, -1))
        out = _EPFinishFn.apply(y.reshape(E * C, M), weight, blob, key)
        return out.view(orig_shape)

    def _forward_jitter(self, h, orig_shape, vemb, temb, noise, is_uniform, n_vis, code, E, jitter):
        """noisy_gate_policy='Jitter' in training (reference sharded_moe.py:290-291 inside :345-381): the logits come
        from `multiplicative_jitter(x.float())`, x = h + modality embedding, while x itself is dispatched."""
        B, T, M = h.shape
        S = B * T
        gate = self.gate
        if code == _lib.MODE_VL:
            x = torch.cat([h[:, :n_vis] + vemb, h[:, n_vis:] + temb], dim=1)
        elif code == _lib.MODE_VISION:
            x = h + vemb
        elif code == _lib.MODE_TEXT:
            x = h + temb
        else:
            x = h
        x2d = x.to(h.dtype).reshape(S, M).contiguous()
        logits = _LogitsFn.apply((x2d.float() * jitter).contiguous(), gate.wg.weight)
        C = ops.capacity(S, E, gate.active_capacity_factor(), gate.min_capacity)
        weight, l_aux, slot, slot_token, slot_weight, rows, exp_counts, _, _ = _GateFn.apply(logits, noise, C, is_uniform)
        self.l_aux = l_aux
        self.exp_counts = self._host_counts(exp_counts)
        self.last_slot = slot
        xd = _DispatchFn.apply(x2d, slot, slot_token)
        dispatched = xd.view(1, self.num_local_experts, C, M)
        y = self.experts(dispatched, rows=rows) if _accepts_rows(self.experts) else self.experts(dispatched)
        out = _CombineFn.apply(y.reshape(E * C, M), weight, slot, slot_token, slot_weight)
        return out.view(orig_shape)

    def forward(self, *input: Tensor, mode='vl', **kwargs: Any) -> Tensor:
        h = input[0]
        if not h.is_cuda:
            raise RuntimeError("eve_amd.MOELayer runs on MI355X only (input is on %s); no CPU fallback" % h.device)
        orig_shape = h.shape
        d_model = h.shape[-1]
        if h.dim() == 2:
            h = h.unsqueeze(0)
        elif h.dim() > 3:
            h = h.reshape(-1, h.shape[-2], d_model)
        B, T, M = h.shape
        S = B * T
        gate = self.gate
        if gate.wg.weight.dtype != torch.float32:
            gate.wg = gate.wg.float()
        has_embed = bool(self.num_vision_tokens)
        code = _mode_code(mode, has_embed)
        vemb = self.vision_embed if has_embed and code in (_lib.MODE_VL, _lib.MODE_VISION) else None
        temb = self.text_embed if has_embed and code in (_lib.MODE_VL, _lib.MODE_TEXT) else None
        if has_embed:  # kernels want both pointers; the unused one is never read
            vemb_in = self.vision_embed if vemb is not None else self.text_embed.detach()
            temb_in = self.text_embed if temb is not None else self.vision_embed.detach()
        else:
            vemb_in = temb_in = None
        E = gate.num_experts if hasattr(gate, "num_experts") else gate.wg.weight.shape[0]
        jitter = None
        if gate.noisy_gate_policy == 'Jitter' and self.training:
            # (drawn before the Gumbel noise, as the reference's gate does, :290-293: same generator consumption)
            jitter = multiplicative_jitter(torch.ones(S, M, dtype=torch.float32, device=h.device), device=h.device)
        noise = self.noise_override
        self.noise_override = None
        is_uniform = False
        if noise is None:
            noise, is_uniform = _draw_noise((S, E), h.device, self.device_noise)
        n_vis = int(self.num_vision_tokens) if has_embed else 0
        if gate.noisy_gate_policy == 'Jitter' and self.training:
            # never set by EVE; the gate then sees a jittered copy of what is dispatched (reference :290-291), so the
            # fused route / dispatch calls (one input for both) do not apply: stage by stage, single rank only
            if self.ep_size > 1:
                raise NotImplementedError("eve_amd.MOELayer: noisy_gate_policy='Jitter' with expert parallelism")
            return self._forward_jitter(h, orig_shape, vemb, temb, noise, is_uniform, n_vis, code, E, jitter)

        if (self.ep_size == 1 and self.fused_call and not self.split_call and not ops.PROFILE.on
                and getattr(self.experts, "packed_parameters", None) is not None and E == self.num_local_experts
                and E <= 32):
            # every expert is local: the whole layer is one C call per direction (eve_moe_fwd / eve_moe_bwd)
            w1, b1, w2, b2 = self.experts.packed_parameters(h.dtype)
            # (non-blocking exp_counts: the gate's closing kernel writes them into the pinned host buffer itself)
            counts_host = None if self.blocking_exp_counts else self._pinned_counts(E)
            out, l_aux, slot, exp_counts = _FusedMoEFn.apply(
                h, gate.wg.weight, vemb_in, temb_in, noise, w1, b1, w2, b2, n_vis, code, gate.active_capacity_factor(),
                gate.min_capacity, is_uniform, self.experts._act_code, torch.is_grad_enabled(), counts_host)
            self.l_aux = l_aux
            if counts_host is not None:
                self._counts_issued(counts_host)
            self.exp_counts = counts_host if counts_host is not None else self._host_counts(exp_counts)
            # a copy (S * 8 bytes): a view would keep the whole `saved` blob (xd, z, a, y: ~0.5 GB at the cfg-2 shape)
            # alive on the module after backward, until this layer's next forward
            self.last_slot = slot.clone() if self.track_last_slot else None
            return out.view(orig_shape)

        packed = getattr(self.experts, "packed_parameters", None) is not None
        if ((self.ep_size > 1 or self.split_call) and self.fused_call and not ops.PROFILE.on and packed and E <= 32
                and E == self.num_local_experts * self.ep_size):
            return self._forward_halves(h, orig_shape, vemb_in, temb_in, noise, is_uniform, n_vis, code, E)

        xd, weight, l_aux, slot, slot_token, slot_weight, rows, exp_counts = _RouteDispatchFn.apply(
            h, gate.wg.weight, vemb_in, temb_in, noise, n_vis, code, gate.active_capacity_factor(), gate.min_capacity,
            is_uniform)
        self.l_aux = l_aux
        self.exp_counts = self._host_counts(exp_counts)
        self.last_slot = slot  # [S,2] int32 token -> slot map of the latest forward (diagnostics)
        C = xd.shape[0] // E

        group = self._resolve_ep_group()
        if group is not None and getattr(self.experts, "packed_parameters", None) is not None:
            nch = ep_chunks(self.num_local_experts)
            if nch > 1:  # pipelined exchange / FFN / exchange over chunks of the local experts
                w1, b1, w2, b2 = self.experts.packed_parameters(xd.dtype)
                expert_output = _EPPipelineFn.apply(group, nch, self.experts._act_code,
                                                    xd.view(self.ep_size, self.num_local_experts, C, M), w1, b1, w2, b2)
                out = _CombineFn.apply(expert_output.view(-1, M), weight, slot, slot_token, slot_weight)
                return out.view(orig_shape)
        if group is not None:
            xd = _AllToAll.apply(group, xd.view(self.ep_size, -1)).view(-1, M)
            rows = None  # peers' fill counts are not exchanged: every capacity row is processed
        # [ep, E_local, C, M] as in the reference (:370)
        dispatched = xd.view(self.ep_size, self.num_local_experts, C, M)
        expert_output = self.experts(dispatched, rows=rows) if _accepts_rows(self.experts) else self.experts(dispatched)
        expert_output = expert_output.reshape(self.ep_size * self.num_local_experts * C, M)
        if group is not None:
            expert_output = _AllToAll.apply(group, expert_output.view(self.ep_size, -1)).view(-1, M)
        out = _CombineFn.apply(expert_output, weight, slot, slot_token, slot_weight)
        return out.view(orig_shape)


def _accepts_rows(experts: nn.Module) -> bool:
    return getattr(experts, "accepts_rows", False)

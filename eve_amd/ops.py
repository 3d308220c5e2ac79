"""Tensor-level wrappers over the C ABI (include/eve_moe.h).

Each function checks device / dtype / contiguity, allocates outputs with torch (device
memory and streams are the only things PyTorch is used for here) and enqueues the HIP
kernels on the current stream.  No function here has a CPU implementation.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib

_DTYPES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
MODES = {None: _lib.MODE_NONE, "none": _lib.MODE_NONE, "vl": _lib.MODE_VL, "vision": _lib.MODE_VISION,
         "text": _lib.MODE_TEXT}
ACTS = {"quick_gelu": _lib.ACT_QUICK_GELU, "gelu": _lib.ACT_GELU, "relu": _lib.ACT_RELU,
        "identity": _lib.ACT_IDENTITY}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError(f"eve_amd: unsupported dtype {t.dtype} (float32 or bfloat16)") from None


def _chk(t: Optional[torch.Tensor], name: str, dtype=None):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"eve_amd: `{name}` must live on a GPU (got {t.device}); there is no CPU path")
    if not t.is_contiguous():
        raise RuntimeError(f"eve_amd: `{name}` must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"eve_amd: `{name}` must be {dtype}, got {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Profile:
    """Optional HIP-event timing of every C-ABI call, recorded on the stream the kernels are
    launched on (bench.py uses it for the per-kernel roofline numbers).  Off by default."""

    def __init__(self):
        self.on = False
        self._ev = []
        self.tagged = {}

    def enable(self):
        self.on, self._ev = True, []

    def disable(self):
        """-> {entry point: (total ms, calls)}; `self.tagged` then holds the same split by the calls' tags
        ({"eve_ffn_fwd[G=2]": ...}: the grouped GEMMs of a 2-group modality split beside those of a 32-expert layer)."""
        self.on = False
        torch.cuda.synchronize()
        out, tagged = {}, {}
        for name, tag, a, b in self._ev:
            dt = a.elapsed_time(b)
            ms, n = out.get(name, (0.0, 0))
            out[name] = (ms + dt, n + 1)
            if tag is not None:
                key = f"{name}[{tag}]"
                ms, n = tagged.get(key, (0.0, 0))
                tagged[key] = (ms + dt, n + 1)
        self._ev = []
        self.tagged = tagged
        return out

    def call(self, name, *args, tag=None):
        if not self.on:
            return _lib.call(name, *args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.call(name, *args)
        b.record()
        self._ev.append((name, tag, a, b))


PROFILE = _Profile()


def capacity(S: int, E: int, capacity_factor: float, min_capacity: int) -> int:
    """models/sharded_moe.py:150-160,:177 (host arithmetic inside the library)."""
    return int(_lib.load().eve_capacity(int(S), int(E), float(capacity_factor), int(min_capacity)))


@dataclass
class Routing:
    """Index-form result of top-2 gating (replaces combine_weights / dispatch_mask [S,E,C])."""
    S: int
    E: int
    C: int
    gates: torch.Tensor        # [S,E] f32
    idx: torch.Tensor          # [S,2] i32
    loc: torch.Tensor          # [S,2] i32
    slot: torch.Tensor         # [S,2] i32 (-1 = not dispatched)
    weight: torch.Tensor       # [S,2] f32
    slot_token: torch.Tensor   # [E*C] i32 (-1 = empty)
    slot_weight: torch.Tensor  # [E*C] f32
    exp_counts: torch.Tensor   # [E] i64 (device)
    rows: torch.Tensor         # [E] i32
    l_aux: torch.Tensor        # [1] f32


def gate_logits(h2d: torch.Tensor, vemb, temb, wg: torch.Tensor, T: int, n_vis: int, mode: int) -> torch.Tensor:
    _chk(h2d, "hidden"), _chk(wg, "wg", torch.float32)
    _chk(vemb, "vision_embed", h2d.dtype), _chk(temb, "text_embed", h2d.dtype)
    S, M = h2d.shape
    E = wg.shape[0]
    assert wg.shape[1] == M
    logits = torch.empty(S, E, dtype=torch.float32, device=h2d.device)
    PROFILE.call("eve_gate_logits_fwd", _p(h2d), _p(vemb), _p(temb), _p(wg), _p(logits), S, T, n_vis, mode, M, E,
              _dt(h2d), _stream())
    return logits


def gumbel_from_uniform(u: torch.Tensor) -> torch.Tensor:
    """Gumbel(0,1) variates of standard-uniform draws, torch.distributions.Gumbel's arithmetic in one kernel."""
    _chk(u, "uniform", torch.float32)
    out = torch.empty_like(u)
    PROFILE.call("eve_gumbel_from_uniform", _p(u), _p(out), u.numel(), _stream())
    return out


NOISE_GUMBEL, NOISE_UNIFORM, NOISE_DEVICE_PHILOX = 0, 1, 2  # include/eve_moe.h: values of noise_is_uniform


def noise_code(noise: torch.Tensor, noise_is_uniform, S: int, E: int) -> int:
    """The `noise_is_uniform` argument of the C calls for (noise, flag) -- False / True / NOISE_DEVICE_PHILOX -- after
    checking that `noise` is what that value says: an fp32 [S,E] tensor, or the int64 {seed, offset} generator state
    the gate kernel draws from and advances (`eve_amd.DeviceNoise.state`)."""
    code = int(noise_is_uniform)
    if code == NOISE_DEVICE_PHILOX:
        _chk(noise, "generator state", torch.int64)
        if noise.numel() != 2:
            raise ValueError("eve_amd: the device generator state is an int64 tensor {seed, offset}")
        return code
    if code not in (NOISE_GUMBEL, NOISE_UNIFORM):
        raise ValueError(f"eve_amd: noise_is_uniform={noise_is_uniform!r}")
    _chk(noise, "noise", torch.float32)
    if tuple(noise.shape) != (S, E):
        raise RuntimeError("eve_amd: gate parameter / noise shapes do not match the layer input")
    return code


def gate_top2(logits: torch.Tensor, noise: torch.Tensor, C: int, noise_is_uniform=False) -> Routing:
    """`noise`: Gumbel variates [S,E], or (noise_is_uniform) the torch.rand draws they are made of, or
    (noise_is_uniform == NOISE_DEVICE_PHILOX) the device generator state the kernel draws from itself."""
    _chk(logits, "logits", torch.float32)
    S, E = logits.shape
    code = noise_code(noise, noise_is_uniform, S, E)
    dev = logits.device
    i32 = dict(dtype=torch.int32, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    r = Routing(S, E, C, torch.empty(S, E, **f32), torch.empty(S, 2, **i32), torch.empty(S, 2, **i32),
                torch.empty(S, 2, **i32), torch.empty(S, 2, **f32), torch.empty(E * C, **i32),
                torch.empty(E * C, **f32), torch.empty(E, dtype=torch.int64, device=dev), torch.empty(E, **i32),
                torch.empty(1, **f32))
    nbytes = _lib.load().eve_gate_workspace_bytes(S, E)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    PROFILE.call("eve_gate_top2_fwd", _p(logits), _p(noise), code, S, E, C, _p(r.gates), _p(r.idx), _p(r.loc), _p(r.slot),
              _p(r.weight), _p(r.slot_token), _p(r.slot_weight), _p(r.exp_counts), _p(r.rows), _p(r.l_aux), _p(ws),
              nbytes, _stream())
    return r


def gate_top2_bwd_raw(gates, idx, loc, exp_counts, S: int, E: int, C: int, d_weight: Optional[torch.Tensor],
                      d_l_aux: Optional[torch.Tensor]) -> torch.Tensor:
    """dlogits [S,E] from d_weight [S,2] (None = no gradient reached the combine weights) and d_l_aux [1]."""
    if d_weight is not None:
        d_weight = d_weight.contiguous()
    _chk(d_weight, "d_weight", torch.float32), _chk(d_l_aux, "d_l_aux", torch.float32)
    dlogits = torch.empty(S, E, dtype=torch.float32, device=gates.device)
    PROFILE.call("eve_gate_top2_bwd", _p(gates), _p(idx), _p(loc), _p(d_weight), _p(d_l_aux), _p(exp_counts),
              S, E, C, _p(dlogits), _stream())
    return dlogits


def gate_top2_bwd(r: Routing, d_weight: Optional[torch.Tensor], d_l_aux: Optional[torch.Tensor]) -> torch.Tensor:
    return gate_top2_bwd_raw(r.gates, r.idx, r.loc, r.exp_counts, r.S, r.E, r.C, d_weight, d_l_aux)


def gate_wgrad(h2d, vemb, temb, dlogits, T: int, n_vis: int, mode: int) -> torch.Tensor:
    _chk(h2d, "hidden"), _chk(dlogits, "dlogits", torch.float32)
    S, M = h2d.shape
    E = dlogits.shape[1]
    dwg = torch.empty(E, M, dtype=torch.float32, device=h2d.device)
    nbytes = _lib.load().eve_gate_wgrad_workspace_bytes(S, M, E)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=h2d.device)
    PROFILE.call("eve_gate_wgrad", _p(h2d), _p(vemb), _p(temb), _p(dlogits), _p(dwg), S, T, n_vis, mode, M, E, _dt(h2d),
              _p(ws), nbytes, _stream())
    return dwg


def dispatch_fwd(h2d, vemb, temb, slot_token: torch.Tensor, T: int, n_vis: int, mode: int,
                 slot: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`slot` [S,2] (the gate's forward map): each token row is then read once for both of its slots."""
    _chk(h2d, "hidden"), _chk(slot_token, "slot_token", torch.int32), _chk(slot, "slot", torch.int32)
    S, M = h2d.shape
    n = slot_token.numel()
    assert slot is None or tuple(slot.shape) == (S, 2)
    xd = torch.empty(n, M, dtype=h2d.dtype, device=h2d.device)
    PROFILE.call("eve_dispatch_fwd", _p(h2d), _p(vemb), _p(temb), _p(slot), _p(slot_token), _p(xd), S, T, n_vis, mode, M, n,
              _dt(h2d), _stream())
    return xd


def dispatch_bwd(dxd: torch.Tensor, slot: torch.Tensor, dlogits, wg, S: int) -> torch.Tensor:
    _chk(dxd, "dxd"), _chk(slot, "slot", torch.int32), _chk(dlogits, "dlogits", torch.float32)
    _chk(wg, "wg", torch.float32)
    M = dxd.shape[-1]
    E = wg.shape[0] if wg is not None else 1
    dh = torch.empty(S, M, dtype=dxd.dtype, device=dxd.device)
    PROFILE.call("eve_dispatch_bwd", _p(dxd), _p(slot), _p(dlogits), _p(wg), _p(dh), S, M, E, _dt(dxd), _stream())
    return dh


def route_bwd(dxd, slot, dlogits, wg, S: int, T: int, n_vis: int, mode: int, want_v: bool, want_t: bool,
              embed_dtype: torch.dtype = torch.float32):
    """dispatch_bwd + embed_grad in one pass: -> (dh [S,M], dvision|None, dtext|None); the embedding
    gradients are stored as `embed_dtype` (fp32, or bf16 for bf16 parameters)."""
    _chk(dxd, "dxd"), _chk(slot, "slot", torch.int32), _chk(dlogits, "dlogits", torch.float32)
    _chk(wg, "wg", torch.float32)
    M = dxd.shape[-1]
    E = wg.shape[0]
    dev = dxd.device
    dh = torch.empty(S, M, dtype=dxd.dtype, device=dev)
    want_v = want_v and mode != _lib.MODE_NONE
    want_t = want_t and mode != _lib.MODE_NONE
    dv = torch.empty(M, dtype=embed_dtype, device=dev) if want_v else None
    dt = torch.empty(M, dtype=embed_dtype, device=dev) if want_t else None
    nbytes = _lib.load().eve_embed_grad_workspace_bytes(S, M) if (want_v or want_t) else 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
    PROFILE.call("eve_route_bwd", _p(dxd), _p(slot), _p(dlogits), _p(wg), _p(dh), _p(dv), _p(dt), S, T, n_vis, mode, M, E,
                 _dt(dxd), _DTYPES[embed_dtype], _p(ws), nbytes, _stream())
    return dh, dv, dt


def embed_grad(dh: torch.Tensor, T: int, n_vis: int, mode: int, want_v=True, want_t=True):
    _chk(dh, "dh")
    S, M = dh.shape
    dv = torch.empty(M, dtype=torch.float32, device=dh.device) if want_v else None
    dt = torch.empty(M, dtype=torch.float32, device=dh.device) if want_t else None
    nbytes = _lib.load().eve_embed_grad_workspace_bytes(S, M)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dh.device)
    PROFILE.call("eve_embed_grad", _p(dh), _p(dv), _p(dt), S, T, n_vis, mode, M, _dt(dh), _p(ws), nbytes, _stream())
    return dv, dt


def combine_fwd(y: torch.Tensor, slot: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    _chk(y, "expert_output"), _chk(slot, "slot", torch.int32), _chk(weight, "weight", torch.float32)
    S = slot.shape[0]
    M = y.shape[-1]
    out = torch.empty(S, M, dtype=y.dtype, device=y.device)
    PROFILE.call("eve_combine_fwd", _p(y), _p(slot), _p(weight), _p(out), S, M, _dt(y), _stream())
    return out


def combine_bwd(dout: torch.Tensor, y: torch.Tensor, slot, slot_token, slot_weight) -> Tuple[torch.Tensor, torch.Tensor]:
    _chk(dout, "dout", y.dtype), _chk(y, "expert_output")
    S, M = dout.shape
    n = slot_token.numel()
    dy = torch.empty(n, M, dtype=y.dtype, device=y.device)
    dw = torch.empty(S, 2, dtype=torch.float32, device=y.device)
    PROFILE.call("eve_combine_bwd", _p(dout), _p(y), _p(slot), _p(slot_token), _p(slot_weight), _p(dy), _p(dw), S, M, n,
              _dt(y), _stream())
    return dy, dw


def ffn_fwd(x, w1, b1, w2, b2, rows, act: int, save_z: bool = True, shared_gpu: bool = False):
    """x [G,R,M]; w1 [E_l,I,M]; b1 [E_l,I]; w2 [E_l,M,I]; b2 [E_l,M] -> (z|None, a, y).
    `shared_gpu`: other kernels (RCCL exchanges) run beside these launches (EVE_FFN_SHARED_GPU)."""
    for t, n in ((x, "x"), (w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2")):
        _chk(t, n, x.dtype)
    _chk(rows, "rows", torch.int32)
    G, R, M = x.shape
    E_l, I, M2 = w1.shape
    assert M2 == M and w2.shape == (E_l, M, I) and b1.shape == (E_l, I) and b2.shape == (E_l, M) and G % E_l == 0
    z = torch.empty(G, R, I, dtype=x.dtype, device=x.device) if save_z else None
    a = torch.empty(G, R, I, dtype=x.dtype, device=x.device)
    y = torch.empty(G, R, M, dtype=x.dtype, device=x.device)
    PROFILE.call("eve_ffn_fwd", _p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(z), _p(a), _p(y), _p(rows), G, E_l, R, M, I,
              act, _dt(x), _lib.FFN_SHARED_GPU if shared_gpu else 0, _stream(), tag=f"G={G}")
    return z, a, y


def ffn_bwd(dy, x, z, a, w1, w2, rows, act: int, grads_out=None, shared_gpu: bool = False):
    """-> (dx [G,R,M], dw1, db1, dw2, db2) with the shapes / dtype of the forward operands.
    `grads_out` = (dw1, db1, dw2, db2) to write the parameter gradients into caller-provided tensors."""
    for t, n in ((dy, "dy"), (x, "x"), (z, "z"), (a, "a"), (w1, "w1"), (w2, "w2")):
        _chk(t, n, x.dtype)
    G, R, M = x.shape
    E_l, I, _ = w1.shape
    dev = x.device
    dz = torch.empty(G, R, I, dtype=x.dtype, device=dev)
    dx = torch.empty(G, R, M, dtype=x.dtype, device=dev)  # rows >= rows[g] stay unwritten and are never read
    if grads_out is not None:
        dw1, db1, dw2, db2 = grads_out
        for t, n, shp in ((dw1, "dw1", (E_l, I, M)), (db1, "db1", (E_l, I)), (dw2, "dw2", (E_l, M, I)), (db2, "db2", (E_l, M))):
            _chk(t, n, x.dtype)
            assert tuple(t.shape) == shp, (n, tuple(t.shape), shp)
    else:
        dw1 = torch.empty(E_l, I, M, dtype=x.dtype, device=dev)
        db1 = torch.empty(E_l, I, dtype=x.dtype, device=dev)
        dw2 = torch.empty(E_l, M, I, dtype=x.dtype, device=dev)
        db2 = torch.empty(E_l, M, dtype=x.dtype, device=dev)
    nbytes = _lib.load().eve_ffn_bwd_workspace_bytes(G, E_l, R, M, I, _dt(x))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
    PROFILE.call("eve_ffn_bwd", _p(dy), _p(x), _p(z), _p(a), _p(w1), _p(w2), _p(dz), _p(dx), _p(dw1), _p(db1), _p(dw2),
              _p(db2), _p(rows), G, E_l, R, M, I, act, _dt(x), _lib.FFN_SHARED_GPU if shared_gpu else 0, _p(ws), nbytes, _stream(),
              tag=f"G={G}")
    return dx, dw1, db1, dw2, db2


# ------------------------------------------------ hard modality split of the non-MoE layers
def split_ln_fwd(h, gamma_v, beta_v, gamma_t, beta_t, n_vis: int, R: int, eps: float):
    """h [B,T,M] -> (xg [2,R,M], mean [S], rstd [S]); see include/eve_moe.h."""
    _chk(h, "hidden")
    for t, n in ((gamma_v, "gamma_v"), (beta_v, "beta_v"), (gamma_t, "gamma_t"), (beta_t, "beta_t")):
        _chk(t, n, h.dtype)
    B, T, M = h.shape
    xg = torch.empty(2, R, M, dtype=h.dtype, device=h.device)
    mean = torch.empty(B * T, dtype=torch.float32, device=h.device)
    rstd = torch.empty(B * T, dtype=torch.float32, device=h.device)
    PROFILE.call("eve_split_ln_fwd", _p(h), _p(gamma_v), _p(beta_v), _p(gamma_t), _p(beta_t), _p(xg), _p(mean), _p(rstd),
                 B, T, n_vis, R, M, float(eps), _dt(h), _stream())
    return xg, mean, rstd


def split_ln_bwd(dxg, h, mean, rstd, gamma_v, gamma_t, n_vis: int):
    """-> (dh [B,T,M], dparams [4,M] fp32 = dgamma_v, dbeta_v, dgamma_t, dbeta_t)."""
    _chk(dxg, "dxg", h.dtype), _chk(h, "hidden")
    B, T, M = h.shape
    R = dxg.shape[1]
    dh = torch.empty_like(h)
    dparams = torch.empty(4, M, dtype=torch.float32, device=h.device)
    nbytes = _lib.load().eve_split_ln_workspace_bytes(B * T, M)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=h.device)
    PROFILE.call("eve_split_ln_bwd", _p(dxg), _p(h), _p(mean), _p(rstd), _p(gamma_v), _p(gamma_t), _p(dh), _p(dparams),
                 B, T, n_vis, R, M, _dt(h), _p(ws), nbytes, _stream())
    return dh, dparams


def layer_norm_fwd(x2d, gamma, beta, eps: float):
    """Plain LayerNorm over the last dimension of x [S,M] on the split-LayerNorm kernel with a single group (every
    token "vision": row s of the input is row s of the output; the second group does not exist and is never touched).
    -> (y [S,M], mean [S], rstd [S])"""
    _chk(x2d, "x"), _chk(gamma, "weight", x2d.dtype), _chk(beta, "bias", x2d.dtype)
    S, M = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(S, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(S, dtype=torch.float32, device=x2d.device)
    PROFILE.call("eve_split_ln_fwd", _p(x2d), _p(gamma), _p(beta), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd),
                 1, S, S, S, M, float(eps), _dt(x2d), _stream(), tag="plain")
    return y, mean, rstd


def layer_norm_bwd(dy2d, x2d, mean, rstd, gamma):
    """-> (dx [S,M], dgamma [M] fp32, dbeta [M] fp32)."""
    _chk(dy2d, "dy", x2d.dtype), _chk(x2d, "x"), _chk(gamma, "weight", x2d.dtype)
    S, M = x2d.shape
    dx = torch.empty_like(x2d)
    dparams = torch.empty(4, M, dtype=torch.float32, device=x2d.device)
    nbytes = _lib.load().eve_split_ln_workspace_bytes(S, M)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x2d.device)
    PROFILE.call("eve_split_ln_bwd", _p(dy2d), _p(x2d), _p(mean), _p(rstd), _p(gamma), _p(gamma), _p(dx), _p(dparams),
                 1, S, S, S, M, _dt(x2d), _p(ws), nbytes, _stream(), tag="plain")
    return dx, dparams[0], dparams[1]


def split_permute(src, B: int, T: int, n_vis: int, R: int, to_groups: bool):
    _chk(src, "src")
    M = src.shape[-1]
    dst = torch.empty((2, R, M) if to_groups else (B, T, M), dtype=src.dtype, device=src.device)
    PROFILE.call("eve_split_permute", _p(src), _p(dst), B, T, n_vis, R, M, 1 if to_groups else 0, _dt(src), _stream())
    return dst


# ----------------------------------------------------- layer-scale residual branch of the encoder layer
def scale_residual_fwd(x, gamma, row_scale, residual, T: int):
    """residual + (x * gamma) * row_scale[s // T]; x, residual [S,M]; gamma [M] | None; row_scale [S // T] | None."""
    _chk(x, "x"), _chk(residual, "residual", x.dtype), _chk(gamma, "gamma", x.dtype), _chk(row_scale, "row_scale", x.dtype)
    S, M = x.shape
    out = torch.empty_like(x)
    PROFILE.call("eve_scale_residual_fwd", _p(x), _p(gamma), _p(row_scale), _p(residual), _p(out), S, T, M, _dt(x), _stream())
    return out


def scale_residual_bwd(dout, x, gamma, row_scale, T: int):
    """-> (dx [S,M], dgamma [M] fp32 | None)."""
    _chk(dout, "dout"), _chk(x, "x", dout.dtype), _chk(gamma, "gamma", dout.dtype), _chk(row_scale, "row_scale", dout.dtype)
    S, M = dout.shape
    dx = torch.empty_like(dout)
    dgamma = torch.empty(M, dtype=torch.float32, device=dout.device) if gamma is not None else None
    nbytes = _lib.load().eve_scale_residual_workspace_bytes(S, M) if gamma is not None else 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dout.device) if nbytes else None
    PROFILE.call("eve_scale_residual_bwd", _p(dout), _p(x), _p(gamma), _p(row_scale), _p(dx), _p(dgamma), S, T, M, _dt(dout),
                 _p(ws), nbytes, _stream())
    return dx, dgamma

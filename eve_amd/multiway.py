"""Hard modality split of EVE's non-MoE layers, MI355X-native.

The reference (models/multiway_transformer.py:366-370) slices the sequence into its vision positions and
its last `num_text_tokens` positions, runs `mlp_imag(norm2_imag(.))` and `mlp_text(norm2_text(.))` on the
slices and concatenates.  Here the LayerNorm writes each token straight into its modality's rows of one
`[2, R, M]` buffer (one pass over the hidden state, no slice copies), the two MLPs run as ONE grouped GEMM
launch sequence (`eve_ffn_fwd/bwd` with two groups and host-known row counts) and one row permutation puts
the result back into token order.  `mode='vision'` / `'text'` (:386-387, :402-403) are the same path with
one group empty.  No CPU fallback.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
from torch import Tensor, nn

from . import ops
from .experts import MultiwayTransformerMLP, _Bank, _GroupedFFNFn, _PackedParamsFn, activation_code


class _SplitLNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, gv, bv, gt, bt, n_vis, R, eps):  # type: ignore[override]
        h = h.contiguous()
        gv, bv, gt, bt = (t.to(h.dtype).contiguous() for t in (gv, bv, gt, bt))
        xg, mean, rstd = ops.split_ln_fwd(h, gv, bv, gt, bt, n_vis, R, eps)
        ctx.save_for_backward(h, mean, rstd, gv, gt)
        ctx.n_vis = n_vis
        return xg

    @staticmethod
    def backward(ctx, dxg):  # type: ignore[override]
        h, mean, rstd, gv, gt = ctx.saved_tensors
        dh, dp = ops.split_ln_bwd(dxg.contiguous(), h, mean, rstd, gv, gt, ctx.n_vis)
        dp = dp.to(h.dtype)
        return dh, dp[0], dp[1], dp[2], dp[3], None, None, None


class _SplitPermuteFn(torch.autograd.Function):
    """Group rows -> token order; the backward is the inverse permutation."""

    @staticmethod
    def forward(ctx, yg, B, T, n_vis):  # type: ignore[override]
        ctx.dims = (B, T, n_vis, yg.shape[1])
        return ops.split_permute(yg.contiguous(), B, T, n_vis, yg.shape[1], to_groups=False)

    @staticmethod
    def backward(ctx, dout):  # type: ignore[override]
        B, T, n_vis, R = ctx.dims
        return ops.split_permute(dout.contiguous(), B, T, n_vis, R, to_groups=True), None, None, None


class ModalitySplitFFN(nn.Module):
    """`norm2_imag`/`mlp_imag` for the vision positions, `norm2_text`/`mlp_text` for the last
    `num_text_tokens` positions.  Attribute names are the reference layer's, so its state-dict entries
    (`...norm2_imag.weight`, `...mlp_text.fc1.weight`, ...) load into this module unchanged."""

    def __init__(self, hidden_act, hidden_size: int, intermediate_size: int, num_text_tokens: int = 40,
                 eps: float = 1e-6) -> None:
        super().__init__()
        self.norm2_imag = nn.LayerNorm(hidden_size, eps=eps)
        self.norm2_text = nn.LayerNorm(hidden_size, eps=eps)
        self.mlp_imag = MultiwayTransformerMLP(hidden_act, hidden_size, intermediate_size)
        self.mlp_text = MultiwayTransformerMLP(hidden_act, hidden_size, intermediate_size)
        self.num_text_tokens = num_text_tokens
        self._bank = _Bank()
        self._act_code = None
        self._rows: Dict[Tuple[int, int, torch.device], Tensor] = {}

    def _mlps(self):
        return [self.mlp_imag, self.mlp_text]

    def forward(self, hidden_states: Tensor, mode: str = "vl") -> Tensor:
        if not hidden_states.is_cuda:
            raise RuntimeError("eve_amd.ModalitySplitFFN runs on MI355X only (input on %s); no CPU fallback"
                               % hidden_states.device)
        if mode not in ("vl", "vision", "text"):
            raise NotImplementedError(f"unsupport mode {mode}")  # reference multiway_transformer.py:404-406
        if self._act_code is None:
            self._act_code = activation_code(self.mlp_imag.activation_fn)
        B, T, M = hidden_states.shape
        n_vis = {"vl": T - self.num_text_tokens, "vision": T, "text": 0}[mode]
        if n_vis < 0:
            raise ValueError(f"sequence of {T} tokens is shorter than num_text_tokens={self.num_text_tokens}")
        nv, nt = B * n_vis, B * (T - n_vis)
        R = max(nv, nt)
        key = (nv, nt, hidden_states.device)
        rows = self._rows.get(key)
        if rows is None:
            rows = self._rows[key] = torch.tensor([nv, nt], dtype=torch.int32, device=hidden_states.device)
        xg = _SplitLNFn.apply(hidden_states, self.norm2_imag.weight, self.norm2_imag.bias, self.norm2_text.weight,
                              self.norm2_text.bias, n_vis, R, self.norm2_imag.eps)
        mlps = self._mlps()
        leaves = self._bank.sync(mlps)  # the two MLPs' parameters alias one packed [2, ...] buffer per kind
        w1, b1, w2, b2 = _PackedParamsFn.apply(self._bank, *leaves)
        dt = hidden_states.dtype
        if w1.dtype != dt:
            w1, b1, w2, b2 = w1.to(dt), b1.to(dt), w2.to(dt), b2.to(dt)
        yg = _GroupedFFNFn.apply(xg, w1, b1, w2, b2, rows, self._act_code)
        return _SplitPermuteFn.apply(yg, B, T, n_vis)

"""EVE's multiway encoder layer around the MI355X-native FFN paths.

Mirrors the reference's `Attention`, `MultiwayTransformerEnocderLayer` and `MultiwayTransformerEncoder`
(models/multiway_transformer.py:57-132, :246-410, :412-595): same constructor arguments, attribute names --
hence state-dict keys -- and forward contract, so a reference checkpoint loads unchanged.  What differs is
what runs:

* the MoE branch (`mlp_moe`, :348-354/:372-377/:388-393) is `eve_amd.MoE` (HIP gate / permute / grouped GEMM);
* the hard modality split of the non-MoE layers (:365-370, :385-387, :401-403) is `eve_amd.ModalitySplitFFN`
  (fused split LayerNorm + two-group grouped GEMM);
* single MLPs (`mlp_vl`, `single_ffn['type'] == 'single'`) are `eve_amd.MultiwayTransformerMLP`;
* attention is PyTorch-ROCm's fused scaled-dot-product attention (fp32 softmax and accumulation inside the
  kernel, as the reference's explicit `q.float() @ k.float().T` asks for) -- a stock op: attention is not on
  the hot path this package rebuilds.

`MultiwayTransformerEncoder.gate_loss()` is the aggregation that the reference's task models do by hand
(model_vqa_beit.py:75-83).  The relative-position-bias TABLE (:462-531, off in EVE's configs) is not built
here; a `relative_position_bias` handed to a layer is honoured.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .experts import MultiwayTransformerMLP
from .layer import MoE
from .multiway import ModalitySplitFFN


class DropPath(nn.Module):
    """Stochastic depth per sample (timm's DropPath, which the reference imports at :15)."""

    def __init__(self, drop_prob: float = 0.0) -> None:
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x: Tensor) -> Tensor:
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        if keep > 0.0:
            mask.div_(keep)
        return x * mask


class _ScaleResidualFn(torch.autograd.Function):
    """residual + drop_path(x * gamma) in one pass (eve_scale_residual_fwd/bwd; reference
    multiway_transformer.py:341, :408).  `row_scale` [B] is DropPath's per-sample mask / keep_prob, or None."""

    @staticmethod
    def forward(ctx, x, gamma, row_scale, residual):  # type: ignore[override]
        from . import ops
        B, T, M = x.shape
        x2 = x.contiguous().view(B * T, M)
        g = None if gamma is None else gamma.to(x.dtype).contiguous()
        rs = None if row_scale is None else row_scale.to(x.dtype).contiguous()
        out = ops.scale_residual_fwd(x2, g, rs, residual.contiguous().view(B * T, M), T)
        ctx.save_for_backward(x2 if g is not None else None, g, rs)
        ctx.dims = (B, T, M)
        ctx.gamma_dtype = None if gamma is None else gamma.dtype
        return out.view(B, T, M)

    @staticmethod
    def backward(ctx, dout):  # type: ignore[override]
        from . import ops
        x2, g, rs = ctx.saved_tensors
        B, T, M = ctx.dims
        dout = dout.contiguous()
        dx, dgamma = ops.scale_residual_bwd(dout.view(B * T, M), x2, g, rs, T)
        if dgamma is not None and dgamma.dtype != ctx.gamma_dtype:
            dgamma = dgamma.to(ctx.gamma_dtype)
        return dx.view(B, T, M), dgamma, None, dout


class _LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension on the library's LayerNorm kernels (the single-group case of
    eve_split_ln_fwd/bwd): one pass forward, one pass + a fixed-order column sum backward, instead of the stock
    forward kernel, two backward kernels and the reduce kernels of the affine gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):  # type: ignore[override]
        from . import ops
        shape = x.shape
        x2 = x.contiguous().view(-1, shape[-1])
        g = weight.to(x.dtype).contiguous()
        y, mean, rstd = ops.layer_norm_fwd(x2, g, bias.to(x.dtype).contiguous(), eps)
        ctx.save_for_backward(x2, mean, rstd, g)
        ctx.param_dtypes = (weight.dtype, bias.dtype)
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):  # type: ignore[override]
        from . import ops
        x2, mean, rstd, g = ctx.saved_tensors
        dx, dgamma, dbeta = ops.layer_norm_bwd(dy.contiguous().view(x2.shape), x2, mean, rstd, g)
        wdt, bdt = ctx.param_dtypes
        return dx.view(dy.shape), dgamma.to(wdt), dbeta.to(bdt), None


def layer_norm(mod: nn.LayerNorm, x: Tensor) -> Tensor:
    """`mod(x)` (reference norm1 / norm2_vl / norm2_imag, multiway_transformer.py:338, :350): on the GPU through the
    library's kernels where their shape limits hold (rows of up to 1024 elements in one wave's registers), else stock."""
    M = x.shape[-1]
    if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and mod.elementwise_affine and mod.bias is not None
            and tuple(mod.normalized_shape) == (M,) and M % 8 == 0 and M <= 1024 and x.numel() > 0):
        return _LayerNormFn.apply(x, mod.weight, mod.bias, mod.eps)
    return mod(x)


class Attention(nn.Module):
    """Reference :57-132.  `qkv` has no bias of its own; `q_bias` / `v_bias` are separate parameters (the key
    bias is fixed at zero), as in BEiT."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(dim))
            self.v_bias = nn.Parameter(torch.zeros(dim))
        else:
            self.q_bias = None
            self.v_bias = None
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x: Tensor, mask: Optional[Tensor] = None, relative_position_bias: Optional[Tensor] = None) -> Tensor:
        B, N, C = x.shape
        qkv_bias = None
        if self.q_bias is not None:
            qkv_bias = torch.cat((self.q_bias, torch.zeros_like(self.v_bias, requires_grad=False), self.v_bias))
        qkv = F.linear(x, self.qkv.weight, qkv_bias)
        q, k, v = qkv.reshape(B, N, 3, self.num_heads, -1).permute(2, 0, 3, 1, 4).unbind(0)
        attn_mask = None
        if relative_position_bias is not None:
            attn_mask = relative_position_bias.unsqueeze(0).to(q.dtype)
        if mask is not None:
            keep = mask.bool()[:, None, None, :]  # [B,1,1,N]: True = attend (reference :120-122)
            if attn_mask is None:
                attn_mask = keep
            else:
                attn_mask = attn_mask.masked_fill(~keep, float("-inf"))
        drop = self.attn_drop.p if self.training else 0.0
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=drop, scale=self.scale)
        out = out.transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(out))


class MultiwayTransformerEnocderLayer(nn.Module):
    """Reference :246-410 (the class name, misspelling included, is the reference's)."""

    def __init__(self, hidden_size, hidden_act, num_attention_heads, intermediate_size, num_tokens, drop_path,
                 layer_scale_init_values=0.1, use_ffn=False, ffn_type='moe', ffn_param=None, single_ffn=False):
        super().__init__()
        self.attn = Attention(hidden_size, num_attention_heads, qkv_bias=True)
        self.norm1 = nn.LayerNorm(hidden_size, eps=1e-6)
        self.num_vision_tokens, self.num_text_tokens = num_tokens
        self.ffn_param = ffn_param
        self.ffn_type = ffn_type
        self.use_ffn = use_ffn
        self.single_ffn = single_ffn if single_ffn else {"type": "vl"}
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        object.__setattr__(self, "_split", None)

        def make_moe(param, default_experts, default_route, default_topk):
            mlp_tmp = MultiwayTransformerMLP(hidden_act, hidden_size, intermediate_size)
            self.num_experts = param.get('num_experts', default_experts)
            self.topk = param.get('topk', default_topk)
            type_route = param.get('type_route', default_route)
            # 'ep_size' is not a key of the reference's configs (it always builds ep_size = 1, layer.py:54-58 unused):
            # expert parallelism is opted into through it, experts sharded over torch.distributed's WORLD group
            return MoE(hidden_size=hidden_size, num_experts=self.num_experts, k=self.topk, expert=mlp_tmp,
                       ep_size=param.get('ep_size', 1), num_tokens=self.num_vision_tokens if type_route else None)

        if self.ffn_type == 'moe' and self.use_ffn:
            self.norm2_vl = nn.LayerNorm(hidden_size, eps=1e-6)
            self.mlp_moe = make_moe(self.ffn_param, 8, False, 2)
        else:
            kind = self.single_ffn['type']
            if kind == 'single':
                self.mlp_imag = MultiwayTransformerMLP(hidden_act, hidden_size, intermediate_size)
                self.norm2_imag = nn.LayerNorm(hidden_size, eps=1e-6)
            elif kind == 'moe':
                self.norm2_imag = nn.LayerNorm(hidden_size, eps=1e-6)
                self.mlp_moe = make_moe(self.single_ffn.get('moe_param', {"num_experts": 2, "topk": 1, "type_route": True}),
                                        2, True, 1)
            else:
                # The four modules are registered on the layer under the reference's names; `_split` is the
                # helper that runs them as one fused path and is deliberately NOT a registered child (its
                # children would otherwise appear twice in the state dict).
                split = ModalitySplitFFN(hidden_act, hidden_size, intermediate_size, self.num_text_tokens)
                self.mlp_imag, self.norm2_imag = split.mlp_imag, split.norm2_imag
                self.mlp_text, self.norm2_text = split.mlp_text, split.norm2_text
                object.__setattr__(self, "_split", split)
            if use_ffn:
                self.norm2_vl = nn.LayerNorm(hidden_size, eps=1e-6)
                self.mlp_vl = MultiwayTransformerMLP(hidden_act, hidden_size, intermediate_size)
        if layer_scale_init_values is not None:
            self.gamma_1 = nn.Parameter(layer_scale_init_values * torch.ones(hidden_size), requires_grad=True)
            self.gamma_2 = nn.Parameter(layer_scale_init_values * torch.ones(hidden_size), requires_grad=True)
        else:
            self.gamma_1 = self.gamma_2 = 1.0
        self.aux_loss_vl = 0
        self.aux_loss_vl_i = 0

    def _moe(self, h: Tensor, **kw) -> Tensor:
        h, aux_loss, _ = self.mlp_moe(h, **kw)
        self.aux_loss_vl = self.aux_loss_vl + aux_loss
        self.aux_loss_vl_i += 1
        return h

    def _ffn(self, h: Tensor, mode: str) -> Tensor:
        if mode not in ('vl', 'vision', 'text'):
            print(f'unsupport mode {mode}')
            raise NotImplementedError
        moe_layer = self.use_ffn and self.ffn_type == 'moe'
        if moe_layer:
            return self._moe(layer_norm(self.norm2_vl, h), mode=mode)
        if mode == 'vl' and self.use_ffn:
            return self.mlp_vl(layer_norm(self.norm2_vl, h))
        kind = self.single_ffn['type']
        if kind == 'single':
            return self.mlp_imag(layer_norm(self.norm2_imag, h))
        if kind == 'moe':
            return self._moe(layer_norm(self.norm2_imag, h))  # the reference passes no mode here (:362)
        return self._split(h, mode=mode)

    def _branch(self, residual: Tensor, x: Tensor, gamma) -> Tensor:
        """`residual + self.drop_path(x * gamma)` (reference :341, :408): on the GPU one fused pass."""
        # the fused pass keeps x's dtype: a gamma of another dtype (fp32 layer scale on bf16 activations promotes the
        # reference's product to fp32) or shape, or a row too wide for the backward's LDS, takes the stock ops
        gamma_ok = not isinstance(gamma, Tensor) or (gamma.dtype == x.dtype and tuple(gamma.shape) == (x.shape[-1],))
        if not (x.is_cuda and x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16) and x.dtype == residual.dtype
                and x.shape == residual.shape and x.shape[-1] % 8 == 0 and x.shape[-1] <= 16384 and gamma_ok):
            return residual + self.drop_path(x * gamma)
        row_scale = None
        p = getattr(self.drop_path, "drop_prob", 0.0)
        if p > 0.0 and self.training:  # DropPath's own draw (same generator consumption), applied inside the kernel
            keep = 1.0 - p
            row_scale = x.new_empty(x.shape[0]).bernoulli_(keep)
            if keep > 0.0:
                row_scale.div_(keep)
        return _ScaleResidualFn.apply(x, gamma if isinstance(gamma, Tensor) else None, row_scale, residual)

    def forward(self, hidden_states: Tensor, attention_mask=None, relative_position_bias=None, mode='vl') -> Tensor:
        residual = hidden_states
        hidden_states = self.attn(layer_norm(self.norm1, hidden_states), mask=attention_mask,
                                  relative_position_bias=relative_position_bias)
        hidden_states = self._branch(residual, hidden_states, self.gamma_1)
        residual = hidden_states
        hidden_states = self._ffn(hidden_states, mode)
        return self._branch(residual, hidden_states, self.gamma_2)


MultiwayTransformerEncoderLayer = MultiwayTransformerEnocderLayer


class MultiwayTransformerEncoder(nn.Module):
    """Reference :412-595 without the relative-position-bias table."""

    def __init__(self, hidden_size, hidden_act, num_attention_heads, intermediate_size, num_hidden_layers, num_tokens,
                 drop_path_rate, ffn_type='moe', ffn_layers=(7, 9, 11), ffn_param=None, single_ffn=False):
        super().__init__()
        self.depth = num_hidden_layers
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, self.depth)]
        self.layers = nn.ModuleList([
            MultiwayTransformerEnocderLayer(hidden_size=hidden_size, hidden_act=hidden_act,
                                            num_attention_heads=num_attention_heads, intermediate_size=intermediate_size,
                                            num_tokens=num_tokens, drop_path=dpr[i], use_ffn=i in ffn_layers,
                                            ffn_type=ffn_type, ffn_param=ffn_param, single_ffn=single_ffn)
            for i in range(num_hidden_layers)])
        self.relative_position_embed = False
        self.num_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.gradient_checkpointing = False
        self.init_std = 0.02
        self.apply(self._init_weights)

    def _build_relative_position_embed(self, config, image_size, patch_size):
        raise NotImplementedError("eve_amd: the relative-position-bias table (reference :462-531) is not built here; "
                                  "EVE's configs run with relative_position_embed = false")

    def fix_init_weight(self) -> None:
        """Reference :533-547."""
        def rescale(param, layer_id):
            param.div_(math.sqrt(2.0 * layer_id))

        for layer_id, layer in enumerate(self.layers):
            rescale(layer.attn.proj.weight.data, layer_id + 1)
            for name in ('mlp_imag', 'mlp_text', 'mlp_vl'):
                if hasattr(layer, name):
                    rescale(getattr(layer, name).fc2.weight.data, layer_id + 1)
            if hasattr(layer, 'mlp_moe'):
                for expert in layer.mlp_moe.moe.experts.experts:
                    rescale(expert.fc2.weight.data, layer_id + 1)

    def _init_weights(self, m: nn.Module) -> None:
        """Reference :549-560 (timm's trunc_normal_: std 0.02, absolute bounds +-2)."""
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            nn.init.trunc_normal_(m.weight, std=self.init_std, a=-2.0, b=2.0)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, inputs_embeds: Tensor, attention_mask=None, mode='vl') -> Tensor:
        hidden_states = inputs_embeds
        for layer in self.layers:
            if self.gradient_checkpointing and self.training:
                hidden_states = torch.utils.checkpoint.checkpoint(
                    lambda h, m, _l=layer: _l(h, m, None, mode), hidden_states, attention_mask, use_reentrant=False)
            else:
                hidden_states = layer(hidden_states=hidden_states, attention_mask=attention_mask,
                                      relative_position_bias=None, mode=mode)
        return hidden_states

    def gate_loss(self, moe_lossweight: float = 1.0):
        """moe_lossweight * mean over MoE layers of (sum of the layer's l_aux / its number of calls), then the
        accumulators are reset: what the reference's task models do after every forward
        (model_vqa_beit.py:75-83).  Returns 0.0 when no MoE layer ran."""
        loss, n = 0.0, 0
        for layer in self.layers:
            if layer.aux_loss_vl_i:
                loss = loss + layer.aux_loss_vl / layer.aux_loss_vl_i
                n += 1
                layer.aux_loss_vl = 0
                layer.aux_loss_vl_i = 0
        return moe_lossweight * loss / n if n else 0.0


def eve_base_encoder(num_vision_tokens: int = 197, num_text_tokens: int = 40,
                     drop_path_rate: float = 0.1) -> MultiwayTransformerEncoder:
    """The encoder of configs/config_vlmoB_base64k.json: 12 layers of width 768, 12 heads, FFN 3072 with
    quick_gelu; layers 10 and 11 are 32-expert top-2 modality-routed MoE layers, the others split hard by modality."""
    return MultiwayTransformerEncoder(hidden_size=768, hidden_act="quick_gelu", num_attention_heads=12,
                                      intermediate_size=3072, num_hidden_layers=12,
                                      num_tokens=(num_vision_tokens, num_text_tokens), drop_path_rate=drop_path_rate,
                                      ffn_type='moe', ffn_layers=[10, 11],
                                      ffn_param={"num_experts": 32, "topk": 2, "type_route": True},
                                      single_ffn={"type": "vl"})

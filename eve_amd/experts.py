"""Expert container and expert FFN on the grouped HIP kernels.

`Experts` keeps the reference's structure -- a ModuleList of deep copies of the template
expert, state-dict keys `experts.{i}.fc1.weight` ... (reference models/experts.py:12-16) --
so checkpoints, `Encoder.apply(_init_weights)` and `fix_init_weight`'s reach-in
(`...moe.experts.experts[i].fc2.weight.data`, reference multiway_transformer.py:545-547)
keep working.  Underneath, the parameters of all local experts alias ONE packed buffer per
tensor kind ([E_local, I, M], ...), which is what the grouped GEMM kernels read; the Python
loop over experts and the chunk/cat copies of the reference's forward (:18-28) are gone.
"""
from __future__ import annotations

import copy
from typing import List, Optional

import torch
from torch import Tensor, nn

from . import ops


# --------------------------------------------------------------------------- activation
class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x) -- `hidden_act: quick_gelu` of both EVE configs
    (reference configs/config_vlmoB_base64k.json:4)."""

    def forward(self, x: Tensor) -> Tensor:
        return x * torch.sigmoid(1.702 * x)


ACT2FN = {"quick_gelu": QuickGELU, "gelu": nn.GELU, "relu": nn.ReLU, "identity": nn.Identity}


def activation_code(fn) -> int:
    """Identify an activation callable by probing it on a few points (works for the HF
    ACT2FN objects the reference uses as well as for ours)."""
    x = torch.linspace(-4.0, 4.0, 33, dtype=torch.float64)
    try:
        y = fn(x.clone())
    except Exception as exc:  # pragma: no cover - exotic callables
        raise NotImplementedError(f"eve_amd: cannot probe expert activation {fn!r}") from exc
    cands = {
        "quick_gelu": x * torch.sigmoid(1.702 * x),
        "gelu": torch.nn.functional.gelu(x),
        "relu": torch.relu(x),
        "identity": x,
    }
    for name, ref in cands.items():
        if torch.allclose(y.double(), ref, atol=1e-6, rtol=0):
            return ops.ACTS[name]
    raise NotImplementedError(f"eve_amd: expert activation {fn!r} has no fused HIP epilogue "
                              "(quick_gelu, gelu, relu, identity are available)")


# ------------------------------------------------------------------- autograd functions
class _GroupedFFNFn(torch.autograd.Function):
    """y[g] = act(x[g] @ w1[e]^T + b1[e]) @ w2[e]^T + b2[e], e = g % E_local (eve_ffn_fwd/bwd)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, rows, act):  # type: ignore[override]
        x = x.contiguous()
        need_grad = any(ctx.needs_input_grad[:5])
        z, a, y = ops.ffn_fwd(x, w1, b1, w2, b2, rows, act, save_z=need_grad)
        if need_grad:
            ctx.save_for_backward(x, z, a, w1, w2)
            ctx.rows, ctx.act = rows, act
        return y

    @staticmethod
    def backward(ctx, dy):  # type: ignore[override]
        x, z, a, w1, w2 = ctx.saved_tensors
        dx, dw1, db1, dw2, db2 = ops.ffn_bwd(dy.contiguous(), x, z, a, w1, w2, ctx.rows, ctx.act)
        return dx, dw1, db1, dw2, db2, None, None


class _PackedParamsFn(torch.autograd.Function):
    """Identity on storage: hands the packed [E_local, ...] buffers that the per-expert
    Parameters alias to autograd, and routes the packed gradients back to those leaves."""

    @staticmethod
    def forward(ctx, bank, *leaves):  # type: ignore[override]
        ctx.n = len(leaves) // 4
        return tuple(t.view_as(t) for t in (bank.w1, bank.b1, bank.w2, bank.b2))

    @staticmethod
    def backward(ctx, dw1, db1, dw2, db2):  # type: ignore[override]
        # one unbind per packed gradient (4 calls) instead of 4 * E_local Python-side selects
        none = (None,) * ctx.n
        cols = [none if g is None else g.unbind(0) for g in (dw1, db1, dw2, db2)]
        grads: List[Optional[Tensor]] = [None]
        for quad in zip(*cols):
            grads.extend(quad)
        return tuple(grads)


class _Bank:
    """Packed storage of the local experts' parameters."""

    def __init__(self):
        self.w1 = self.b1 = self.w2 = self.b2 = None

    @staticmethod
    def _aliased(packed: Optional[Tensor], leaves: List[Tensor]) -> bool:
        if packed is None or packed.shape[0] != len(leaves):
            return False
        step = packed.stride(0) * packed.element_size() if packed.dim() > 0 else 0
        base = packed.data_ptr()
        for e, p in enumerate(leaves):
            if (p.data_ptr() != base + e * step or p.dtype != packed.dtype or p.device != packed.device
                    or p.shape != packed.shape[1:] or not p.is_contiguous()):
                return False
        return True

    @staticmethod
    @torch.no_grad()
    def _pack(leaves: List[nn.Parameter]) -> Tensor:
        packed = torch.stack([p.detach() for p in leaves]).contiguous()
        for e, p in enumerate(leaves):
            p.data = packed[e]  # same values, storage now inside the packed buffer
        return packed

    _KINDS = (("w1", "fc1", "weight"), ("b1", "fc1", "bias"), ("w2", "fc2", "weight"), ("b2", "fc2", "bias"))

    def leaves(self, experts) -> List[nn.Parameter]:
        """The experts' Parameters in (expert, w1, b1, w2, b2) order.  The list is cached and re-validated per
        call by identity through the modules' parameter dicts (a fraction of the cost of 4 * E attribute
        look-ups through nn.Module.__getattr__, which dominated the host time of a step)."""
        cached = getattr(self, "_leaves", None)
        mods = experts._modules.values() if isinstance(experts, nn.Module) else experts
        if cached is not None and len(cached) == 4 * len(mods):
            ok, i = True, 0
            for m in mods:
                sub = m._modules
                p1, p2 = sub["fc1"]._parameters, sub["fc2"]._parameters
                if (p1["weight"] is not cached[i] or p1["bias"] is not cached[i + 1] or p2["weight"] is not cached[i + 2]
                        or p2["bias"] is not cached[i + 3]):
                    ok = False
                    break
                i += 4
            if ok:
                return cached
        out: List[nn.Parameter] = []
        for m in experts:
            out += [m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]
        self._leaves = out
        return out

    def sync(self, experts) -> List[nn.Parameter]:
        """(Re)pack when `.to()/.cuda()/.bfloat16()` or a foreign optimizer replaced storages; returns leaves()."""
        leaves = self.leaves(experts)
        # fast path: every leaf still starts where it did when it was last found aliased.  That memory belongs to the
        # packed buffers this bank holds, so a leaf that points there has not been replaced (`.to()`, `.data = ...`
        # and optimizers that swap storages all move it elsewhere).
        ptrs = [p.data_ptr() for p in leaves]
        if ptrs == getattr(self, "_ptrs", None):
            return leaves
        for k, (kind, _, _) in enumerate(self._KINDS):
            mine = leaves[k::4]
            if not self._aliased(getattr(self, kind), mine):
                setattr(self, kind, self._pack(mine))
        self._ptrs = [p.data_ptr() for p in leaves]
        return leaves


def _is_mlp(expert: nn.Module) -> bool:
    return (isinstance(getattr(expert, "fc1", None), nn.Linear) and isinstance(getattr(expert, "fc2", None), nn.Linear)
            and expert.fc1.bias is not None and expert.fc2.bias is not None and hasattr(expert, "activation_fn")
            and expert.fc1.in_features == expert.fc2.out_features and expert.fc1.out_features == expert.fc2.in_features)


class MultiwayTransformerMLP(nn.Module):
    """fc1 -> activation -> fc2 (reference multiway_transformer.py:44-55), forward on the
    grouped-FFN HIP kernel with a single group."""

    def __init__(self, hidden_act, hidden_size, intermediate_size):
        super().__init__()
        self.activation_fn = ACT2FN[hidden_act]() if isinstance(hidden_act, str) else hidden_act
        self.fc1 = nn.Linear(hidden_size, intermediate_size)
        self.fc2 = nn.Linear(intermediate_size, hidden_size)
        self._act_code = None

    def forward(self, hidden_states: Tensor) -> Tensor:
        if self._act_code is None:
            self._act_code = activation_code(self.activation_fn)
        if not hidden_states.is_cuda:
            raise RuntimeError("eve_amd.MultiwayTransformerMLP runs on MI355X only; no CPU fallback")
        shape = hidden_states.shape
        x = hidden_states.reshape(1, -1, shape[-1])
        dt = x.dtype
        y = _GroupedFFNFn.apply(x, self.fc1.weight.to(dt).unsqueeze(0), self.fc1.bias.to(dt).unsqueeze(0),
                                self.fc2.weight.to(dt).unsqueeze(0), self.fc2.bias.to(dt).unsqueeze(0), None,
                                self._act_code)
        return y.reshape(shape)


class Experts(nn.Module):
    """`Experts(expert, num_local_experts, expert_group_name)`; `forward([ep, E_local, C, M])`
    returns the same shape (reference models/experts.py:10-28)."""

    accepts_rows = True

    def __init__(self, expert, num_local_experts=1, expert_group_name=None):
        super().__init__()
        if not _is_mlp(expert):
            raise NotImplementedError(
                "eve_amd.Experts needs an fc1 -> activation_fn -> fc2 expert with biases "
                "(the reference's MultiwayTransformerMLP); other expert bodies have no HIP kernel here")
        self.experts = nn.ModuleList([copy.deepcopy(expert) for _ in range(num_local_experts)])
        self.num_local_experts = num_local_experts
        self.expert_group_name = expert_group_name
        self._bank = _Bank()
        self._act_code = activation_code(expert.activation_fn)
        self._bank.sync(self.experts)

    def packed_parameters(self, dtype: Optional[torch.dtype] = None):
        """(w1 [E_l,I,M], b1 [E_l,I], w2 [E_l,M,I], b2 [E_l,M]) wired into autograd."""
        leaves = self._bank.sync(self.experts)
        w1, b1, w2, b2 = _PackedParamsFn.apply(self._bank, *leaves)
        if dtype is not None and w1.dtype != dtype:  # fp32 master weights, bf16 compute (autocast-style)
            w1, b1, w2, b2 = w1.to(dtype), b1.to(dtype), w2.to(dtype), b2.to(dtype)
        return w1, b1, w2, b2

    def forward(self, inputs: Tensor, rows: Optional[Tensor] = None) -> Tensor:
        if not inputs.is_cuda:
            raise RuntimeError("eve_amd.Experts runs on MI355X only (input on %s); no CPU fallback" % inputs.device)
        ep, e_local, C, M = inputs.shape
        assert e_local == self.num_local_experts
        w1, b1, w2, b2 = self.packed_parameters(inputs.dtype)
        x = inputs.reshape(ep * e_local, C, M)
        y = _GroupedFFNFn.apply(x, w1, b1, w2, b2, rows, self._act_code)
        return y.view(ep, e_local, C, M)
